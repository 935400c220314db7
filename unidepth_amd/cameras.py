"""Ground-truth camera models accepted by UniDepthV2.infer(rgb, camera=...) besides a [...,3,3] pinhole K tensor
(SURVEY.md 8f next-3).  Parameter conventions follow the reference classes of the same names (unidepth/utils/camera.py):

    Pinhole(K)                                   # utils/camera.py:229-266
    EUCM(params=[fx, fy, cx, cy, alpha, beta])   # enhanced unified camera model, utils/camera.py:276-328
    Spherical(params=[fx, fy, cx, cy, W, H, hfov/2, vfov/2])   # equirectangular panorama (angles in rad), :331-410
    OPENCV(params=[fx, fy, cx, cy, k1..k6, p1, p2, s1..s4])    # radial (k1..k3; k4..k6 must be 0) + tangential + thin prism, :412-694
    Fisheye624(params=[fx, fy, cx, cy, k1..k6, p1, p2, s1..s4])  # 6-coefficient fisheye + tangential + thin prism, :697-974
    MEI(params=[fx, fy, cx, cy, k1, k2, p1, p2, xi])           # unified omnidirectional model, :977-1082

Only what the infer() path needs is here: the bookkeeping that maps a camera of the ORIGINAL image to the network input
(aspect padding = a crop by negative offsets, then the resize factor; utils/camera.py:78-81,115-120 and the Spherical overrides
:336-357) and a tag for the ray kernel (ud_rays_from_kinv gt_mode).  The unprojection itself runs on the GPU
(csrc/pointwise.hip rays_kernel; the iterative models -- OPENCV, Fisheye624, MEI -- through ud_rays_from_camera, which
reproduces the reference's Newton / trust-region solvers including the image-wide early exit of the radial loop).  Unlike the reference, infer() does not modify the camera object it is given.
Objects of the reference's own classes are accepted as well (matched by class name and `.params`)."""
from __future__ import annotations

import torch

GT_PINHOLE, GT_EUCM, GT_SPHERICAL, GT_OPENCV, GT_FISHEYE624, GT_MEI = 1, 2, 3, 4, 5, 6


class Camera:
    gt_mode = 0
    n_params = 4

    def __init__(self, params: torch.Tensor):
        params = torch.as_tensor(params, dtype=torch.float32)
        if params.ndim == 1:
            params = params.unsqueeze(0)
        assert params.shape[-1] == self.n_params, f"{type(self).__name__} takes {self.n_params} parameters"
        self.params = params.clone()

    @property
    def K(self) -> torch.Tensor:
        K = torch.eye(3).repeat(self.params.shape[0], 1, 1)
        K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2] = self.params[:, 0], self.params[:, 1], self.params[:, 2], self.params[:, 3]
        return K

    def network_params(self, paddings, resize_factor: float) -> torch.Tensor:
        """Parameters after `crop(-pad)` and `resize(rf)` (reference unidepthv2.py:299-303), [n, n_params] fp32."""
        pl, pr, pt, pb = paddings
        p = self.params.clone()
        p[:, 2] += pl
        p[:, 3] += pt
        p[:, :4] *= resize_factor
        return p


class Pinhole(Camera):
    gt_mode = GT_PINHOLE

    def __init__(self, K: torch.Tensor = None, params: torch.Tensor = None):
        if params is None:
            K = torch.as_tensor(K, dtype=torch.float32).reshape(-1, 3, 3)
            params = torch.stack([K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2]], dim=1)
        super().__init__(params)


class EUCM(Camera):
    gt_mode = GT_EUCM
    n_params = 6


class Spherical(Camera):
    gt_mode = GT_SPHERICAL
    n_params = 8

    def network_params(self, paddings, resize_factor: float) -> torch.Tensor:
        pl, pr, pt, pb = paddings
        p = self.params.clone()
        W, H = p[:, 4].clone(), p[:, 5].clone()
        p[:, 2] += pl
        p[:, 3] += pt
        p[:, 4] = W + pl + pr                    # a padded panorama spans proportionally more angle
        p[:, 5] = H + pt + pb
        p[:, 6] *= (W + pl + pr) / W
        p[:, 7] *= (H + pt + pb) / H
        p[:, :6] *= resize_factor
        return p


class _SingleCamera(Camera):
    """The iterative models unproject one camera at a time in the reference (`B` is taken from the pixel grid, which has batch
    1: utils/camera.py:498-509), so one parameter row is what infer() can be given."""

    def __init__(self, params: torch.Tensor):
        super().__init__(params)
        assert self.params.shape[0] == 1, f"{type(self).__name__}: one camera per infer() call (as in the reference)"


class OPENCV(_SingleCamera):
    gt_mode = GT_OPENCV
    n_params = 16

    def __init__(self, params: torch.Tensor):
        super().__init__(params)
        assert float(self.params[..., 7:10].abs().sum()) == 0.0, "Do not support poly division model"   # utils/camera.py:416-418


class Fisheye624(_SingleCamera):
    gt_mode = GT_FISHEYE624
    n_params = 16


class MEI(_SingleCamera):
    gt_mode = GT_MEI
    n_params = 9


class BatchCamera(Camera):
    """Several cameras, one per image of the batch, possibly of DIFFERENT models (the reference's wrapper utils/camera.py:1145-1308: its
    unproject() concatenates every member's own unproject, :1166-1171; infer() crops / resizes each member, :1173-1186).  Members are
    single-camera objects of the classes above.  uniform(): the same cameras as ONE batched object when every member is of one closed-form
    class (Pinhole / EUCM / Spherical), which the batched ray kernel takes; None otherwise (the plan then records one ray launch per image)."""
    gt_mode = -1

    def __init__(self, cameras):
        self.cameras = list(cameras)
        assert self.cameras and all(isinstance(c, Camera) and not isinstance(c, BatchCamera) and c.params.shape[0] == 1 for c in self.cameras), \
            "BatchCamera: a non-empty list of single cameras"
        self.params = torch.zeros(len(self.cameras), 16)
        for i, c in enumerate(self.cameras):
            self.params[i, : c.params.shape[1]] = c.params[0]

    @property
    def gt_modes(self):
        return tuple(c.gt_mode for c in self.cameras)

    def uniform(self):
        cls = type(self.cameras[0])
        if cls in (Pinhole, EUCM, Spherical) and all(type(c) is cls for c in self.cameras):
            p = torch.cat([c.params for c in self.cameras], dim=0)
            return Pinhole(params=p) if cls is Pinhole else cls(p)
        return None


def _flatten(cams):
    out = []
    for c in cams:
        out.extend(_flatten(c) if isinstance(c, (list, tuple)) else [c])
    return out


_BY_NAME = {"Pinhole": Pinhole, "EUCM": EUCM, "Spherical": Spherical, "OPENCV": OPENCV, "Fisheye624": Fisheye624, "MEI": MEI}


def as_camera(obj) -> Camera:
    """Own classes pass through; reference-class instances (or anything with a matching class name and `.params`) are wrapped."""
    if isinstance(obj, Camera):
        return obj
    name = type(obj).__name__
    if name == "BatchCamera" and getattr(obj, "cameras", None):
        cams = [as_camera(c) for c in _flatten(obj.cameras)]
        return cams[0] if len(cams) == 1 else BatchCamera(cams)
    cls = _BY_NAME.get(name)
    if cls is None or not hasattr(obj, "params"):
        raise NotImplementedError(f"camera model '{name}' is not implemented ({', '.join(_BY_NAME)} are)")
    if cls is Pinhole:
        return Pinhole(params=torch.as_tensor(obj.params, dtype=torch.float32)[..., :4])
    return cls(torch.as_tensor(obj.params, dtype=torch.float32))
