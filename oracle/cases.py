"""TEST INFRASTRUCTURE ONLY -- the parity cases shared by make_golden.py, tests/ and smoke().

Each case is fully determined by seeds, so inputs/checkpoints are regenerated wherever a test runs;
only the (sub-sampled) *reference outputs* are stored, in tests/golden/<case>.npz."""
from __future__ import annotations

import torch

DEMO_K = [[518.86, 0.0, 325.58], [0.0, 519.47, 253.74], [0.0, 0.0, 1.0]]   # assets/demo/intrinsics.npy (rounded)

CASES = {
    # BASELINE.json configs[0]: V2 ViT-S/14, single 462x616 RGB (scripts/demo.py shape after resize), bs=1
    "vits_462x616_b1": dict(arch="vits14", H=462, W=616, B=1, camera=False, ckpt_seed=123, img_seed=1),
    # demo-shaped input with the demo's GT pinhole intrinsics (pad-free, resized 480x640 -> 490x644)
    "vits_480x640_b2_cam": dict(arch="vits14", H=480, W=640, B=2, camera=True, ckpt_seed=123, img_seed=2),
    # KITTI-like wide image: aspect padding (60,61) + down-resize (SURVEY.md 8d shape policy table)
    "vits_375x1242_b1": dict(arch="vits14", H=375, W=1242, B=1, camera=False, ckpt_seed=123, img_seed=3),
    "vitb_518x518_b1": dict(arch="vitb14", H=518, W=518, B=1, camera=False, ckpt_seed=124, img_seed=4),
    # BASELINE.json configs[1] at bs=1 (same network shape as the headline bs=8 workload)
    "vitl_518x518_b1": dict(arch="vitl14", H=518, W=518, B=1, camera=False, ckpt_seed=125, img_seed=5),
    # ViT-L with the resolution_level knob set (unidepthv2.py:252-260): level 3 -> pixel bounds [320k, 360k] -> 644x966 runs at 490x728
    "vitl_644x966_b1_lvl3": dict(arch="vitl14", H=644, W=966, B=1, camera=False, ckpt_seed=125, img_seed=12, resolution_level=3),
    # non-pinhole GT cameras: EUCM fisheye-like, and an equirectangular strip wide enough to be aspect-padded
    "vits_300x400_eucm": dict(arch="vits14", H=300, W=400, B=1, camera=("EUCM", [190.0, 192.0, 203.0, 148.0, 0.62, 1.08]),
                              ckpt_seed=123, img_seed=6),
    "vits_200x560_spherical": dict(arch="vits14", H=200, W=560, B=2, camera=("Spherical", [0.0, 0.0, 0.0, 0.0, 560.0, 200.0, 1.4, 0.5]),
                                   ckpt_seed=123, img_seed=7),
    # GT cameras with iterative unprojection (utils/camera.py OPENCV / Fisheye624 / MEI): radial + tangential + thin-prism terms all
    # active; the MEI strip is wide enough to be aspect-padded (pads 28/28) and resized
    "vits_300x400_opencv": dict(arch="vits14", H=300, W=400, B=1, ckpt_seed=123, img_seed=8,
                                camera=("OPENCV", [300.0, 302.0, 203.0, 148.0, -0.25, 0.08, -0.01, 0.0, 0.0, 0.0, 1e-3, -2e-3, 1e-3, 5e-4, -1e-3, 2e-4])),
    "vits_300x400_fisheye624": dict(arch="vits14", H=300, W=400, B=1, ckpt_seed=123, img_seed=9,
                                    camera=("Fisheye624", [190.0, 192.0, 203.0, 148.0, 0.05, -0.01, 0.002, -5e-4, 1e-4, -1e-5, 1e-3, -1e-3, 5e-4, 1e-4, -5e-4, 1e-4])),
    "vits_200x640_mei": dict(arch="vits14", H=200, W=640, B=2, ckpt_seed=123, img_seed=10,
                             camera=("MEI", [250.0, 252.0, 318.0, 101.0, -0.1, 0.02, 1e-3, -1e-3, 0.9])),
}


def case_inputs(case: dict):
    g = torch.Generator().manual_seed(case["img_seed"])
    rgb = torch.randint(0, 256, (case["B"], 3, case["H"], case["W"]), dtype=torch.uint8, generator=g)
    if isinstance(case["camera"], tuple):
        cam = (case["camera"][0], torch.tensor(case["camera"][1], dtype=torch.float32))
    else:
        cam = torch.tensor(DEMO_K, dtype=torch.float32) if case["camera"] else None
    return rgb, cam


def digest(out: dict) -> dict:
    """Compact, deterministic sub-sampling of an infer() result (keeps fixtures small)."""
    d = {"intrinsics": out["intrinsics"].float().numpy()}
    for k in ("depth", "confidence", "radius"):
        d[k] = out[k][:, :, 3::7, 5::7].float().contiguous().numpy()
    for k in ("points", "rays"):
        d[k] = out[k][:, :, 3::11, 5::11].float().contiguous().numpy()
    d["depth_features"] = out["depth_features"][:, 1::8, ::2, ::2].float().contiguous().numpy()
    d["depth_mean"] = out["depth"].double().mean().reshape(1).numpy()
    return d
