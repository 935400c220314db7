"""Dev check (authoring container only): oracle/restate.py's iterative camera models against the reference's own classes
(utils/camera.py OPENCV / Fisheye624 / MEI get_rays) over a few parameter sets, including pad/resize bookkeeping."""
import warnings

import torch

from oracle import ref_loader, restate

warnings.simplefilter("ignore")
ref_loader._prepare()
import unidepth.utils.camera as rc  # noqa: E402

SETS = [
    ("OPENCV", [180., 182., 98., 70., -0.25, 0.08, -0.01, 0, 0, 0, 1e-3, -2e-3, 0, 0, 0, 0]),
    ("OPENCV", [180., 182., 98., 70., -0.25, 0.08, -0.01, 0, 0, 0, 1e-3, -2e-3, 1e-3, 5e-4, -1e-3, 2e-4]),
    ("OPENCV", [180., 182., 98., 70., 0, 0, 0, 0, 0, 0, 0, 0, 1e-3, 5e-4, -1e-3, 2e-4]),        # thin prism only: the ones-Jacobian quirk
    ("OPENCV", [180., 182., 98., 70., -0.3, 0.1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]),                # radial only
    ("OPENCV", [180., 182., 98., 70., 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]),                     # no distortion
    ("Fisheye624", [120., 121., 98., 70., -0.02, 0.01, -0.003, 0.001, 0, 0, 1e-3, -1e-3, 5e-4, 1e-4, -5e-4, 1e-4]),
    ("Fisheye624", [90., 90., 98., 70., 0.05, -0.01, 0.002, -0.0005, 1e-4, -1e-5, 0, 0, 0, 0, 0, 0]),
    ("MEI", [150., 151., 98., 70., -0.1, 0.02, 1e-3, -1e-3, 0.9]),
    ("MEI", [150., 151., 98., 70., -0.1, 0.02, 0, 0, 1.0]),
    ("MEI", [150., 151., 98., 70., 0, 0, 0, 0, 0.5]),
]
H, W = 140, 196
worst = 0.0
for name, p in SETS:
    for pads, rf in [((0, 0, 0, 0), 1.0), ((0, 0, 7, 8), 1.37)]:
        cam = getattr(rc, name)(params=torch.tensor(p))
        pl, pr, pt, pb = pads
        cam = cam.crop(-pl, -pt, -pr, -pb).resize(rf)
        Hn, Wn = int((H + pt + pb) * rf), int((W + pl + pr) * rf)
        ref = cam.get_rays((1, Hn, Wn))
        mine = restate.OracleV2._rays_from_camera_model(name, torch.tensor(p), pads, rf, Hn, Wn)
        err = (ref - mine).abs().max().item()
        worst = max(worst, err)
        print(f"{name:11s} pads={pads} rf={rf}: max abs diff {err:.2e}  finite={bool(ref.isfinite().all())}")
print("worst", worst)
assert worst < 2e-6
