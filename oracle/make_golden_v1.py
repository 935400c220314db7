"""TEST INFRASTRUCTURE ONLY -- golden vectors for the V1 components from the REAL reference (/root/reference, CPU fp32, restated timm
layers of oracle/stubs/timm).  Authoring container:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_v1"""
from __future__ import annotations

import contextlib
import io
import json
import os
import warnings

import numpy as np
import torch

from . import ref_loader, synth_v1

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    assert ref_loader.available()
    warnings.simplefilter("ignore")
    ref_loader._prepare()
    with contextlib.redirect_stdout(io.StringIO()):
        from unidepth.models import UniDepthV1  # type: ignore
        from unidepth.utils.misc import max_stack  # type: ignore
        cfg_ref = json.load(open(os.path.join(ref_loader.REF_ROOT, "configs", "config_v1_cnvnxtl.json")))
        model = UniDepthV1(cfg_ref).eval()
    cfg = synth_v1.load_config_v1()
    sd = synth_v1.make_synthetic_checkpoint_v1(cfg, 211)
    model.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 3, 128, 160, generator=g)
    with torch.no_grad():
        outs, cls = model.pixel_encoder(x)
    d = {}
    for j, (a, b) in enumerate(model.pixel_decoder.slices_encoder_range):
        d[f"stage{j}"] = max_stack(outs[a:b])[:, ::3, ::3, ::7].contiguous().numpy()
    d["cls_last4"] = torch.cat([cls[-i - 1] for i in range(4)], dim=-1).numpy()
    np.savez_compressed(os.path.join(OUT, "v1_convnext_128x160.npz"), **d)
    print({k: v.shape for k, v in d.items()})
    # whole infer(): the reference's own decoder / pre- / post-processing on the statement-by-statement restatement of xformers NystromAttention (oracle/stubs/xformers)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(OUT)))
    from test_oracle_v1_pins import V1_CASES, v1_case_inputs, v1_digest
    for name in V1_CASES:
        rgb, K, skip = v1_case_inputs(name)
        with torch.no_grad():
            out = model.infer(rgb, K, skip_camera=skip)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **v1_digest({k: v.detach() for k, v in out.items()}))
        print(name, float(out["depth"].mean()))
    # UniDepthV1 on DINOv2 ViT-L/14 (configs/config_v1_vitl14.json)
    from test_oracle_v1_pins import VITL_CASES, vitl_case_inputs
    with contextlib.redirect_stdout(io.StringIO()):
        model = UniDepthV1(json.load(open(os.path.join(ref_loader.REF_ROOT, "configs", "config_v1_vitl14.json")))).eval()
    cfg = synth_v1.load_config_v1("vitl14")
    model.load_state_dict(synth_v1.make_synthetic_checkpoint_v1(cfg, 212), strict=True)
    for name in VITL_CASES:
        rgb, K, skip = vitl_case_inputs(name)
        with torch.no_grad():
            out = model.infer(rgb, K, skip_camera=skip)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **v1_digest({k: v.detach() for k, v in out.items()}))
        print(name, float(out["depth"].mean()))


if __name__ == "__main__":
    main()
