"""UniDepthV1 (ConvNeXt-L backbone) on MI355X -- SURVEY.md 8f next-1, BASELINE.json configs[3] (640x480, bs=16, conv-heavy path).
Mirrors the reference class (unidepth/models/unidepthv1/unidepthv1.py:101-450): from_pretrained / to / eval / infer(rgbs, intrinsics,
skip_camera), attribute `image_shape`; device arithmetic in libunidepth_hip.so.

Status (round 2): the ENCODER half runs on the engine -- `pixel_encoder(image)` = ConvNeXt-L (backbones/convnext.py:301-471) as a
launch program of hand-written HIP kernels (depth-wise 7x7 conv, LayerNorm, MFMA GEMMs for stem / down-sampling / MLP, max_stack,
class-token means), parity-tested against the oracle restatement, which is pinned to the reference's own ConvNeXt code.  The
decoder half (unidepthv1/decoder.py: camera head, spherical-harmonics ray embedding, Nystrom attention blocks whose arithmetic
lives in the un-vendored xformers package) is not built yet: infer() raises NotImplementedError after naming what is missing --
it never falls back to another implementation."""
from __future__ import annotations

import json
import math
import os
from typing import List, Optional

import torch

from . import ops
from .ops import UD_ACT_GELU, UD_EPI_F16, UD_EPI_F32

CONVNEXT = {"convnext_large": dict(depths=(3, 3, 27, 3), dims=(192, 384, 768, 1536))}     # models/encoder.py:127-136


def _rup(x, m):
    return (x + m - 1) // m * m


def _padk16(w: torch.Tensor) -> torch.Tensor:
    n, k = w.shape
    kp = _rup(k, 64)
    out = w.new_zeros(n, kp)
    out[:, :k] = w
    return out.to(torch.float16).contiguous()


def pack_convnext(config: dict, sd: dict, device) -> dict:
    """Load-time repack of the reference's `pixel_encoder.*` tensors (exact fp32 algebra, then one fp16 rounding of the GEMM operands):
    LayerNorm affines folded into the consuming Linear / down-sampling Conv2d (zero padding 0: exact), block layer-scale `gamma`
    folded into fc2, depth-wise filters tap-major [49][C], stem / down-sampling filters as GEMM rows in im2col column order."""
    a = CONVNEXT[config["model"]["pixel_encoder"]["name"]]
    f = {k: v.detach().to(torch.float32).cpu() for k, v in sd.items() if k.startswith("pixel_encoder.")}
    pe = "pixel_encoder."
    w = {}

    def p16(name, t):
        w[name] = _padk16(t).to(device)

    def p32(name, t):
        w[name] = t.to(torch.float32).contiguous().to(device)

    p16("stem.w", f[pe + "stem.0.weight"].reshape(a["dims"][0], 48)); p32("stem.b", f[pe + "stem.0.bias"])
    p32("stem.g", f[pe + "stem.1.weight"]); p32("stem.beta", f[pe + "stem.1.bias"])
    for s, (dep, d) in enumerate(zip(a["depths"], a["dims"])):
        if s > 0:
            g, b = f[f"{pe}stages.{s}.downsample.0.weight"], f[f"{pe}stages.{s}.downsample.0.bias"]
            wc, bc = f[f"{pe}stages.{s}.downsample.1.weight"], f[f"{pe}stages.{s}.downsample.1.bias"]      # [Cout, Cin, 2, 2]
            p16(f"ds.{s}.w", (wc * g[None, :, None, None]).permute(0, 2, 3, 1).reshape(d, -1))              # columns (ky, kx, cin)
            p32(f"ds.{s}.b", bc + torch.einsum("ocyx,c->o", wc, b))
        for i in range(dep):
            r = f"{pe}stages.{s}.blocks.{i}."
            p32(f"blk.{s}.{i}.dw.w", f[r + "conv_dw.weight"].reshape(d, 49).t()); p32(f"blk.{s}.{i}.dw.b", f[r + "conv_dw.bias"])
            g, b = f[r + "norm.weight"], f[r + "norm.bias"]
            w1, b1 = f[r + "mlp.fc1.weight"], f[r + "mlp.fc1.bias"]
            p16(f"blk.{s}.{i}.fc1.w", w1 * g[None, :]); p32(f"blk.{s}.{i}.fc1.b", b1 + w1 @ b)
            ls = f[r + "gamma"]
            p16(f"blk.{s}.{i}.fc2.w", f[r + "mlp.fc2.weight"] * ls[:, None]); p32(f"blk.{s}.{i}.fc2.b", f[r + "mlp.fc2.bias"] * ls)
    return w


class _EncPlan:
    """Device buffers + launch program of the ConvNeXt encoder for one (batch, network image shape)."""

    def __init__(self, model: "UniDepthV1", B: int, Hn: int, Wn: int):
        w, dev = model._w, model.device
        a = model._arch
        f16, f32 = torch.float16, torch.float32

        def z(*shape, dtype=f16):
            return torch.zeros(*shape, dtype=dtype, device=dev)

        P = ops.Program()
        self.prog, self.B = P, B
        self.tap_points = []

        def tap(name, fn):
            self.tap_points.append((name, len(P), fn))
        self.img = z(B, 3, Hn, Wn, dtype=f32)
        H, W = Hn // 4, Wn // 4
        dims, depths = a["dims"], a["depths"]
        rows = B * H * W
        patches = z(rows, 64)
        P.patchify4(self.img, patches, B, Hn, Wn, 64)
        x0 = z(rows, dims[0], dtype=f32)
        P.gemm(A=patches, W=w["stem.w"], bias=w["stem.b"], out=x0, M=rows, N=dims[0], K=64, lda=64, ldw=64, ldc=dims[0], epi=UD_EPI_F32, tag="stem")
        x = z(rows, dims[0], dtype=f32)
        P.layernorm(x=x0, y=x, rows=rows, D=dims[0], ldx=dims[0], ldy=dims[0], eps=1e-6, rows_per_img=rows, in_rows_per_img=rows,
                    out_rows_per_img=rows, out_f32=1, gamma=w["stem.g"], beta=w["stem.beta"])
        tap("stem", lambda x=x, H=H, W=W, C=dims[0]: x.view(B, H, W, C).permute(0, 3, 1, 2).clone())
        self.shapes, self.stage_max, self.cls = [], [], []
        nblk = sum(depths)
        blk = 0
        for s, (dep, C) in enumerate(zip(depths, dims)):
            if s > 0:
                Cp = dims[s - 1]
                Ho, Wo = H // 2, W // 2
                rows = B * Ho * Wo
                col = z(rows, 4 * Cp)
                P.layernorm_patchify2(x, col, B, H, W, Cp, 4 * Cp, 1e-6)
                xn = z(rows, C, dtype=f32)
                P.gemm(A=col, W=w[f"ds.{s}.w"], bias=w[f"ds.{s}.b"], out=xn, M=rows, N=C, K=4 * Cp, lda=4 * Cp, ldw=4 * Cp, ldc=C, epi=UD_EPI_F32,
                       tag=f"downsample.{s}")
                x, H, W = xn, Ho, Wo
            y = z(rows, C, dtype=f32)
            xh = z(rows, C)
            hid = z(rows, 4 * C)
            smax = z(rows, C, dtype=f32)
            for i in range(dep):
                P.dwconv7(x=x, w=w[f"blk.{s}.{i}.dw.w"], bias=w[f"blk.{s}.{i}.dw.b"], y=y, B=B, H=H, W=W, C=C, ldx=C, ldy=C, tag=f"dwconv.s{s}")
                P.layernorm(x=y, y=xh, rows=rows, D=C, ldx=C, ldy=C, eps=1e-6, rows_per_img=rows, in_rows_per_img=rows, out_rows_per_img=rows)
                P.gemm(A=xh, W=w[f"blk.{s}.{i}.fc1.w"], bias=w[f"blk.{s}.{i}.fc1.b"], out=hid, M=rows, N=4 * C, K=C, lda=C, ldw=C, ldc=4 * C,
                       epi=UD_EPI_F16, act=UD_ACT_GELU, tag=f"enc.fc1.s{s}")
                P.gemm(A=hid, W=w[f"blk.{s}.{i}.fc2.w"], bias=w[f"blk.{s}.{i}.fc2.b"], out=x, M=rows, N=C, K=4 * C, lda=4 * C, ldw=4 * C, ldc=C,
                       epi=UD_EPI_F32, accumulate=1, tag=f"enc.fc2.s{s}")
                P.max_(smax, x, rows * C, i == 0)
                if blk >= nblk - 4:                        # the decoder reads the class tokens of the LAST four blocks (decoder.py:375-377)
                    cbuf = z(B, C, dtype=f32)
                    P.spatial_mean(x, cbuf, B, H * W, C, C)
                    self.cls.append(cbuf)
                tap(f"block{blk}", lambda x=x, H=H, W=W, C=C: x.view(B, H, W, C).clone())
                blk += 1
            self.shapes.append((H, W, C))
            self.stage_max.append(smax)


class UniDepthV1:
    """Engine counterpart of the reference's UniDepthV1 (unidepthv1.py:101).  See the module docstring for what runs today."""

    def __init__(self, config: dict, eps: float = 1e-6, **kwargs):
        self.config = config
        name = config["model"]["pixel_encoder"]["name"]
        if name not in CONVNEXT:
            raise NotImplementedError(f"UniDepthV1 pixel_encoder {name!r}: only the ConvNeXt-L backbone (config_v1_cnvnxtl) is implemented on this engine")
        self._arch = dict(CONVNEXT[name])
        self._arch["output_idx"] = list(config["model"]["pixel_encoder"].get("output_idx", [3, 6, 33, 36]))
        self.image_shape = list(config["data"]["image_shape"])                         # unidepthv1.py:444
        self._sd = None
        self._w = None
        self._device = torch.device("cpu")
        self._plans: dict = {}

    # ---- checkpoint I/O (same HF layout as V2) ----
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, **kwargs):
        path = str(pretrained_model_name_or_path)
        if not os.path.isdir(path):
            from huggingface_hub import snapshot_download
            path = snapshot_download(path, allow_patterns=["config.json", "model.safetensors", "pytorch_model.bin"])
        with open(os.path.join(path, "config.json")) as f:
            config = json.load(f)
        model = cls(config)
        st = os.path.join(path, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
        return model.load_state_dict(sd)

    def save_pretrained(self, path: str):
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.config, f)
        save_file({k: v.contiguous() for k, v in self._sd.items()}, os.path.join(path, "model.safetensors"))

    def load_state_dict(self, state_dict: dict, strict: bool = False):
        if "model" in state_dict and not torch.is_tensor(state_dict["model"]):
            state_dict = state_dict["model"]                                            # unidepthv1.py:381-385
        self._sd = {k.replace("module.", ""): v.detach().float().cpu() for k, v in state_dict.items()}
        self._w = None
        self._plans.clear()
        return self

    def state_dict(self):
        return dict(self._sd)

    @property
    def device(self):
        return self._device

    def to(self, device):
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if device != self._device:
            self._device = device
            self._w = None
            self._plans.clear()
        return self

    def cuda(self):
        return self.to("cuda")

    def eval(self):
        return self

    def _ensure_packed(self):
        if self._device.type != "cuda":
            raise RuntimeError("UniDepthV1 (MI355X engine) runs on a ROCm GPU only: call .to('cuda') first; there is no CPU path")
        if self._sd is None:
            raise RuntimeError("no weights loaded (use from_pretrained or load_state_dict)")
        if self._w is None:
            with torch.cuda.device(self._device):
                self._w = pack_convnext(self.config, self._sd, self._device)

    def _enc_plan(self, B, Hn, Wn) -> _EncPlan:
        key = ("enc", B, Hn, Wn)
        if key not in self._plans:
            while len(self._plans) >= 4:
                self._plans.pop(next(iter(self._plans)))
            with torch.cuda.device(self._device):
                self._plans[key] = _EncPlan(self, B, Hn, Wn)
        return self._plans[key]

    # ---- encoder seam (backbones/convnext.py:447-458) ----
    embed_dim = 192
    patch_size = 16                                                                   # unidepthv1.py:427-429 (non-DINO encoders)

    @property
    def embed_dims(self):
        return [d for dep, d in zip(self._arch["depths"], self._arch["dims"]) for _ in range(dep)]

    @property
    def depths(self):
        return list(self._arch["output_idx"])

    @torch.no_grad()
    def pixel_encoder(self, image: torch.Tensor, keep_all: bool = True):
        """image [B,3,H,W] (normalised network image) -> (outs, cls_tokens) like ConvNeXt.forward: 36 block outputs [B,h,w,C] and their
        spatial means [B,1,C].  keep_all=False skips copying the 32 intermediate block outputs out of the reused stream buffers and
        returns None for them (the decoder only reads stage-wise maxima and the last four class tokens: see stage_features())."""
        self._ensure_packed()
        B, _, Hn, Wn = image.shape
        with torch.cuda.device(self._device):
            plan = self._enc_plan(B, Hn, Wn)
            plan.img.copy_(image.to(self._device, torch.float32), non_blocking=True)
            n = sum(self._arch["depths"])
            outs: List[Optional[torch.Tensor]] = [None] * n
            if keep_all:
                pos = 0
                for name, at, fn in plan.tap_points:
                    if not name.startswith("block"):
                        continue
                    plan.prog.run(pos, at)
                    pos = at
                    outs[int(name[5:])] = fn()
                plan.prog.run(pos, len(plan.prog))
            else:
                plan.prog.run()
            cls: List[Optional[torch.Tensor]] = [None] * n
            for j, cb in enumerate(plan.cls):
                cls[n - 4 + j] = cb.clone().unsqueeze(1)
            if keep_all:                                   # class tokens of the other blocks: same reduction kernel on the copies
                for i, o in enumerate(outs):
                    if cls[i] is None and o is not None:
                        t = torch.empty(B, o.shape[-1], dtype=torch.float32, device=self._device)
                        ops.check(ops.lib.ud_spatial_mean_f32(o.data_ptr(), t.data_ptr(), B, o.shape[1] * o.shape[2], o.shape[3], o.shape[3],
                                                              ops.cur_stream()), "ud_spatial_mean_f32")
                        cls[i] = t.unsqueeze(1)
        return outs, cls

    @torch.no_grad()
    def stage_features(self, image: torch.Tensor):
        """What the V1 decoder consumes (decoder.py:366-377): per stage the element-wise max over its block outputs [B,h,w,C], and the
        class tokens of the last four blocks, deepest first."""
        self._ensure_packed()
        B, _, Hn, Wn = image.shape
        with torch.cuda.device(self._device):
            plan = self._enc_plan(B, Hn, Wn)
            plan.img.copy_(image.to(self._device, torch.float32), non_blocking=True)
            plan.prog.run()
            feats = [m.view(B, h, w, c).clone() for m, (h, w, c) in zip(plan.stage_max, plan.shapes)]
            cls = [plan.cls[-i - 1].clone().unsqueeze(1) for i in range(4)]
        return feats, cls

    @torch.no_grad()
    def infer(self, rgbs: torch.Tensor, intrinsics=None, skip_camera: bool = False):
        raise NotImplementedError(
            "UniDepthV1.infer on the MI355X engine: the ConvNeXt-L encoder runs (pixel_encoder / stage_features), the V1 decoder "
            "(unidepthv1/decoder.py: camera head, rsh_cart_8 ray embedding, Nystrom attention) is not built yet -- SURVEY.md 8f next-1")
