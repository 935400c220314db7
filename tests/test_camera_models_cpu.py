"""The per-pixel arithmetic of the iterative GT camera kernels (unidepth_amd/csrc/camera_models.h: OPENCV, Fisheye624, MEI),
compiled for the HOST and compared with the oracle's restatement of the reference solvers (oracle/restate.py, itself checked
against the reference classes by tools/check_camera_restatement.py and pinned by the tests/golden/*_{opencv,fisheye624,mei}
fixtures).  The GPU runs the same header inside pointwise.hip (tests/test_infer_gpu.py covers that end to end)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SETS = [
    ("OPENCV", 4, [180., 182., 98., 70., -0.25, 0.08, -0.01, 0, 0, 0, 1e-3, -2e-3, 0, 0, 0, 0]),
    ("OPENCV", 4, [180., 182., 98., 70., -0.25, 0.08, -0.01, 0, 0, 0, 1e-3, -2e-3, 1e-3, 5e-4, -1e-3, 2e-4]),
    ("OPENCV", 4, [180., 182., 98., 70., -0.3, 0.1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]),
    ("OPENCV", 4, [180., 182., 98., 70., 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]),
    ("Fisheye624", 5, [120., 121., 98., 70., -0.02, 0.01, -0.003, 0.001, 0, 0, 1e-3, -1e-3, 5e-4, 1e-4, -5e-4, 1e-4]),
    ("Fisheye624", 5, [90., 90., 98., 70., 0.05, -0.01, 0.002, -0.0005, 1e-4, -1e-5, 0, 0, 0, 0, 0, 0]),
    ("MEI", 6, [150., 151., 98., 70., -0.1, 0.02, 1e-3, -1e-3, 0.9]),
    ("MEI", 6, [150., 151., 98., 70., -0.1, 0.02, 0, 0, 1.0]),
    ("MEI", 6, [150., 151., 98., 70., 0, 0, 0, 0, 0.5]),
]


@pytest.fixture(scope="module")
def host_lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cam_host") / "libcam_host.so")
    # -ffp-contract=off: no fused multiply-adds, i.e. the rounding of the torch restatement (the GPU build contracts; its tolerance is in the GPU test)
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", out, os.path.join(HERE, "cam_host", "cam_host.cpp")])
    lib = ctypes.CDLL(out)
    lib.cam_host_rays.restype = ctypes.c_int
    lib.cam_host_rays.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return lib


@pytest.mark.parametrize("name,model,params", SETS)
def test_camera_model_arithmetic_matches_oracle(host_lib, name, model, params):
    from oracle import restate
    Hn, Wn = 140, 196
    p = np.zeros(16, dtype=np.float32)
    p[: len(params)] = params
    rays = np.zeros((3, Hn, Wn), dtype=np.float32)
    steps = host_lib.cam_host_rays(p.ctypes.data, model, Hn, Wn, rays.ctypes.data)
    ref = restate.OracleV2._rays_from_camera_model(name, torch.tensor(params), (0, 0, 0, 0), 1.0, Hn, Wn)[0].numpy()
    assert np.isfinite(rays).all() and np.isfinite(ref).all()
    assert np.abs(rays - ref).max() < 2e-6, np.abs(rays - ref).max()
    assert np.allclose(np.linalg.norm(rays, axis=0), 1.0, atol=1e-6)
    if model != 6 and any(abs(v) > 0 for v in params[4:10]):
        assert 1 <= steps < 10                                   # the image-wide exit fired before the iteration cap


def test_batch_camera_wrapper_host_logic():
    """unidepth_amd.cameras.BatchCamera (the reference's multi-camera wrapper, utils/camera.py:1145-1308): members are matched by class name,
    nested lists flatten, one closed-form class collapses to the batched object the ray kernel takes, mixed / iterative models stay a list."""
    from unidepth_amd import cameras as C

    class Pinhole:                                             # stand-ins for the reference's classes: matched by class name + .params
        def __init__(self, p):
            self.params = torch.tensor([p])

    class EUCM(Pinhole):
        pass

    class OPENCV(Pinhole):
        pass

    class BatchCamera:
        def __init__(self, cams):
            self.cameras = cams
    pin = [Pinhole([100.0 + i, 101.0, 50.0, 40.0]) for i in range(3)]
    b = C.as_camera(BatchCamera(pin))
    assert isinstance(b, C.BatchCamera) and b.gt_modes == (C.GT_PINHOLE,) * 3
    u = b.uniform()
    assert isinstance(u, C.Pinhole) and u.params.shape == (3, 4) and float(u.K[2, 0, 0]) == 102.0
    mixed = C.as_camera(BatchCamera([[pin[0]], [EUCM([90.0, 91.0, 50.0, 40.0, 0.6, 1.1]), OPENCV([80.0, 81.0, 50.0, 40.0] + [0.0] * 12)]]))
    assert mixed.gt_modes == (C.GT_PINHOLE, C.GT_EUCM, C.GT_OPENCV) and mixed.uniform() is None and mixed.params.shape == (3, 16)
    assert isinstance(C.as_camera(BatchCamera([pin[1]])), C.Pinhole)              # a wrapper around one camera is that camera
    with pytest.raises(AssertionError):
        C.BatchCamera([])
