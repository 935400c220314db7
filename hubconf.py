"""torch.hub entry point (`torch.hub.load(<this repo>, "UniDepth", version="v2", backbone="vitl14")`), same call as the
reference's hubconf.py:25-41; the implementation lives in unidepth_amd/hub.py."""
dependencies = ["torch", "huggingface_hub"]

from unidepth_amd.hub import UniDepth  # noqa: E402,F401
