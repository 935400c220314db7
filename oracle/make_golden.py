"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/*.npz by running the REAL reference
(/root/reference, CPU fp32) on the seeded cases of oracle/cases.py.  Run in the authoring container:

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden

The fixtures pin oracle/restate.py on boxes where the reference tree is absent (the GPU box)."""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np
import torch

from . import cases, ref_loader, synth

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main(names=None):
    assert ref_loader.available(), "needs /root/reference"
    warnings.simplefilter("ignore")
    os.makedirs(OUT, exist_ok=True)
    for name, case in cases.CASES.items():
        if names and name not in names:
            continue
        cfg = synth.load_config(case["arch"])
        sd = synth.make_synthetic_checkpoint(cfg, case["ckpt_seed"])
        ref = ref_loader.build_reference(case["arch"], sd)
        rgb, cam = cases.case_inputs(case)
        if isinstance(cam, tuple):                       # the reference's own camera class, built from the case's parameters
            import unidepth.utils.camera as refcam  # type: ignore
            cam_ref = getattr(refcam, cam[0])(params=cam[1].clone())
        else:
            cam_ref = cam.clone() if cam is not None else None
        if case.get("resolution_level") is not None:
            ref.resolution_level = case["resolution_level"]
        with torch.no_grad():
            out = ref.infer(rgb, cam_ref)
        d = cases.digest({k: v.detach() for k, v in out.items()})
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
        print(name, {k: v.shape for k, v in d.items()}, "depth_mean", float(d["depth_mean"][0]))
        del ref, sd


if __name__ == "__main__":
    main(sys.argv[1:])
