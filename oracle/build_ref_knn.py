"""TEST INFRASTRUCTURE ONLY -- compiles the REFERENCE's own CPU K-NN (unidepth/ops/knn/src/knn_cpu.cpp + its pybind file knn_ext.cpp,
from where they lie under /root/reference; nothing is copied into this repo) into oracle/_ref/knn/KNN.so, CPU-only (WITH_CUDA
undefined, so knn.h:68-76 dispatches every call to KNearestNeighborIdxCpu).  The module is the `KNN` extension that the reference's
functions/knn.py:13 imports; it pins oracle/restate_eval.py and serves as the "reference" CPU baseline of tools/bench_eval_ops.py.

    python -m oracle.build_ref_knn        (authoring container; ~1 min; needs g++ and ninja, both in the image)

oracle/_ref/ is git-ignored (build output) but travels to the GPU box with the snapshot."""
from __future__ import annotations

import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/unidepth/ops/knn/src"
OUT_DIR = os.path.join(_HERE, "_ref", "knn")
SO_PATH = os.path.join(OUT_DIR, "KNN.so")


def build(verbose: bool = False) -> str:
    """Build if the reference sources are present and the library is missing; return its path ("" if it cannot be built)."""
    if os.path.exists(SO_PATH):
        return SO_PATH
    if not os.path.isdir(REF_SRC):
        return ""
    from torch.utils.cpp_extension import load
    os.makedirs(OUT_DIR, exist_ok=True)
    load(name="KNN", sources=[os.path.join(REF_SRC, "knn_ext.cpp"), os.path.join(REF_SRC, "knn_cpu.cpp")],
         extra_include_paths=[REF_SRC], extra_cflags=["-O3"], build_directory=OUT_DIR, verbose=verbose, is_python_module=True)
    return SO_PATH if os.path.exists(SO_PATH) else ""


def load_ref():
    """The compiled reference module (attributes knn_points_idx, knn_points_backward), or None when it has not been built."""
    if not os.path.exists(SO_PATH):
        return None
    if "KNN" in sys.modules:
        return sys.modules["KNN"]
    import torch  # noqa: F401  (libtorch must be loaded first)
    spec = importlib.util.spec_from_file_location("KNN", SO_PATH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.modules["KNN"] = mod
    return mod


if __name__ == "__main__":
    print(build(verbose=True) or "reference sources not present: nothing built")
