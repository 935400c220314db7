"""TEST INFRASTRUCTURE ONLY -- CPU fp32 restatement of the UniDepthV1 inference path (BASELINE.json configs[3]).
Nothing under unidepth_amd/ imports this file; it is the checker for the HIP engine's V1 components.

Encoder = ConvNeXt-L as the reference builds it (unidepth/models/encoder.py:127-136 -> backbones/convnext.py:301-471):
  stem Conv2d(3, 192, k=4, s=4) + LayerNorm2d                                 convnext.py:370-383
  4 stages, depths (3, 3, 27, 3), dims (192, 384, 768, 1536); stages 1..3 start with LayerNorm2d + Conv2d(k=2, s=2)   :245-266
  block: depth-wise 7x7 conv (pad 3) -> LayerNorm (channels last) -> Linear(C, 4C) -> GELU(erf) -> Linear(4C, C) -> * gamma -> + input   :130-223
  forward returns EVERY block's output in NHWC plus its spatial mean as a "class token"                           :447-458
LayerNorm / LayerNorm2d / Mlp / create_conv2d come from timm (un-vendored, un-pinned: requirements.txt:16): eps = 1e-6, symmetric
static padding.  PARITY UNPINNED at that boundary: no reference test or golden vector covers the ConvNeXt path; the pins of this
file are the reference's own code executed with the restated timm layers of oracle/stubs/timm (tests/test_oracle_v1_pins.py).
"""
from __future__ import annotations

import math
from typing import List, Tuple

import torch
import torch.nn.functional as F

CONVNEXT = {"convnext_large": dict(depths=(3, 3, 27, 3), dims=(192, 384, 768, 1536))}   # encoder.py:127-136


def convnext_arch(config: dict) -> dict:
    a = dict(CONVNEXT[config["model"]["pixel_encoder"]["name"]])
    a["output_idx"] = list(config["model"]["pixel_encoder"].get("output_idx", [3, 6, 33, 36]))   # cumulative stage ends
    return a


class OracleConvNeXt:
    """Functional ConvNeXt over a dict of fp32 tensors keyed like the reference state_dict (prefix `pixel_encoder.`)."""

    def __init__(self, config: dict, state_dict: dict, prefix: str = "pixel_encoder."):
        self.a = convnext_arch(config)
        self.p = prefix
        self.w = {k: v.detach().to(torch.float32).cpu() for k, v in state_dict.items() if k.startswith(prefix)}
        self.taps: dict = {}
        self.keep_taps = False

    def _ln2d(self, x, name):                      # LayerNorm2d: over C of NCHW, eps 1e-6 (timm)
        y = F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), self.w[name + ".weight"], self.w[name + ".bias"], 1e-6)
        return y.permute(0, 3, 1, 2)

    def block(self, x, name):
        """ConvNeXtBlock.forward, conv_mlp=False (convnext.py:206-223)."""
        w = self.w
        C = x.shape[1]
        y = F.conv2d(x, w[name + ".conv_dw.weight"], w[name + ".conv_dw.bias"], padding=3, groups=C)
        y = y.permute(0, 2, 3, 1)
        y = F.layer_norm(y, (C,), w[name + ".norm.weight"], w[name + ".norm.bias"], 1e-6)
        y = F.linear(F.gelu(F.linear(y, w[name + ".mlp.fc1.weight"], w[name + ".mlp.fc1.bias"])), w[name + ".mlp.fc2.weight"], w[name + ".mlp.fc2.bias"])
        y = y.permute(0, 3, 1, 2) * w[name + ".gamma"].reshape(1, -1, 1, 1)
        return y + x

    def encode(self, image: torch.Tensor) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
        """ConvNeXt.forward (convnext.py:447-458): image [B,3,H,W] (normalised) -> (36 block outputs [B,h,w,C], 36 means [B,1,C])."""
        p, w = self.p, self.w
        x = F.conv2d(image, w[p + "stem.0.weight"], w[p + "stem.0.bias"], stride=4)
        x = self._ln2d(x, p + "stem.1")
        if self.keep_taps:
            self.taps["stem"] = x.clone()
        outs = []
        for s, depth in enumerate(self.a["depths"]):
            if s > 0:
                x = self._ln2d(x, f"{p}stages.{s}.downsample.0")
                x = F.conv2d(x, w[f"{p}stages.{s}.downsample.1.weight"], w[f"{p}stages.{s}.downsample.1.bias"], stride=2)
            for b in range(depth):
                x = self.block(x, f"{p}stages.{s}.blocks.{b}")
                outs.append(x.permute(0, 2, 3, 1).contiguous())
        return outs, [o.mean(dim=(1, 2)).unsqueeze(1).contiguous() for o in outs]

    def stage_features(self, outs: List[torch.Tensor]) -> List[torch.Tensor]:
        """What the V1 decoder consumes (unidepthv1/decoder.py:366-373, 468-480): per stage the element-wise MAX over the stage's
        block outputs (utils/misc.py:18-21 max_stack over slices_encoder_range)."""
        ends = self.a["output_idx"]
        starts = [0] + ends[:-1]
        feats = []
        for i, j in zip(starts, ends):
            feats.append(outs[i] if j - i == 1 else torch.stack(outs[i:j], dim=-1).max(dim=-1).values)
        return feats


# =====================================================================================================================
# Decoder + infer().  Reference: unidepth/models/unidepthv1/decoder.py (Decoder.forward :364-463, CameraHead :39-111, DepthHead
# :114-330, ListAdapter :21-36), layers/attention.py:81-165 (AttentionBlock), layers/mlp.py:9-35, layers/upsample.py:13-45
# (ConvUpsample), layers/convnext.py:5-44 (CvnxtBlock), layers/positional_encoding.py:14-57 (PositionEmbeddingSine),
# utils/geometric.py:12-73,227-252 (generate_rays, spherical_zbuffer_to_euclidean, flat_interpolate), utils/sht.py:833 (rsh_cart_8),
# unidepthv1.py:28-98,288-373 (pre / post-processing, infer).
#
# layers_8 / layers_4 (NystromBlock, layers/nystrom_attention.py:22-84): their attention is xformers.components.attention.NystromAttention(
# num_landmarks=128) -- requirements.txt:24 `xformers>=0.0.26`, un-vendored and absent here.  oracle/stubs/xformers/components/attention is a
# statement-by-statement restatement of that module (nystrom.py forward + AvgPool, core.py scaled_query_key_softmax / scaled_dot_product_attention,
# utils.py iterative_pinv of xformers v0.0.26) and the live reference runs on it (tests/test_oracle_v1_pins.py).  The finding (round 5): the
# reference hands it 4-D [b, n, h, d] tensors (einops "b n (h d) -> b n h d"); the module reads `seq_len = k.size(-2)` = h (4 / 2 heads), finds
# `num_landmarks (128) >= seq_len` and takes its small-sequence branch, plain softmax attention over the last two axes: every token's h
# head-vectors attend to EACH OTHER, att = softmax(q k^T / sqrt d) in [b, n, h, h], out = att v.  No landmarks, no pseudo-inverse, no mixing
# between tokens.  `nystrom_block_attention()` below states that; rounds 2-4 had restated the PAPER's algorithm per head over the token axis
# (`nystrom_attention()`, kept because tests/test_oracle_nystrom_cpu.py triangulates it against Hugging Face's Nystromformer and the engine's
# split-key flash kernel is tested against it), which is NOT what the reference's module computes for this layout.
# The pin is as strong as the restatement of the two load-bearing lines (`seq_len = k.size(-2)`, `if self.num_landmarks >= seq_len:`): the
# package cannot be executed here.
# =====================================================================================================================
def real_sh_deg8(xyz: torch.Tensor) -> torch.Tensor:
    """Real spherical harmonics up to degree 8 at unit vectors xyz [..., 3] -> [..., 81], Y_n^m at index n(n+1)+m, Condon-Shortley
    phase included (the basis of utils/sht.py:833 rsh_cart_8, which is an auto-generated closed-form table).  Computed here from the
    standard recurrences instead: with Q_l^m(z) = P_l^m(z) / (1-z^2)^{m/2} (a polynomial), A_m + i B_m = (x + i y)^m,
        Y_l^0 = K_l^0 Q_l^0,   Y_l^m = (-1)^m sqrt2 K_l^m Q_l^m A_m,   Y_l^-m = (-1)^m sqrt2 K_l^m Q_l^m B_m,
        K_l^m = sqrt((2l+1)/(4 pi) (l-m)!/(l+m)!),  Q_m^m = (2m-1)!!,  Q_{m+1}^m = (2m+1) z Q_m^m,
        (l-m) Q_l^m = (2l-1) z Q_{l-1}^m - (l+m-1) Q_{l-2}^m."""
    x, y, z = xyz[..., 0], xyz[..., 1], xyz[..., 2]
    L = 8
    A = [torch.ones_like(x)]
    Bm = [torch.zeros_like(x)]
    for m in range(1, L + 1):
        A.append(A[-1] * x - Bm[-1] * y)
        Bm.append(A[-2] * y + Bm[-1] * x)
    out = [None] * ((L + 1) ** 2)
    for m in range(0, L + 1):
        q_prev2 = None
        dfact = 1.0
        for k in range(1, m + 1):
            dfact *= (2 * k - 1)
        q_prev = torch.full_like(z, dfact)                 # Q_m^m
        for l in range(m, L + 1):
            if l == m:
                q = q_prev
            elif l == m + 1:
                q = (2 * m + 1) * z * q_prev
                q_prev2, q_prev = q_prev, q
            else:
                q = ((2 * l - 1) * z * q_prev - (l + m - 1) * q_prev2) / (l - m)
                q_prev2, q_prev = q_prev, q
            K = math.sqrt((2 * l + 1) / (4 * math.pi) * math.factorial(l - m) / math.factorial(l + m))
            if m == 0:
                out[l * (l + 1)] = K * q
            else:
                sgn = -1.0 if m % 2 else 1.0
                out[l * (l + 1) + m] = sgn * math.sqrt(2.0) * K * q * A[m]
                out[l * (l + 1) - m] = sgn * math.sqrt(2.0) * K * q * Bm[m]
    return torch.stack(out, dim=-1)


def iterative_pinv(k: torch.Tensor, n_iter: int = 6) -> torch.Tensor:
    """Newton-Schulz pseudo-inverse of a row-stochastic matrix [..., n, n] as xformers' nystrom.py does it: Z0 = K^T / ||K||_1,
    Z <- 1/4 Z (13 I - KZ (15 I - KZ (7 I - KZ)))."""
    eye = torch.eye(k.shape[-1], dtype=k.dtype)
    v = k.transpose(-1, -2) / k.sum(dim=-2).max(dim=-1).values[..., None, None]
    for _ in range(n_iter):
        kv = k @ v
        v = (0.25 * v) @ (13 * eye - kv @ (15 * eye - kv @ (7 * eye - kv)))
    return v


def segment_means(x: torch.Tensor, n: int) -> torch.Tensor:
    """xformers AvgPool landmark pooling over the token axis of [B, N, D]: n segments, the first n - N % n of floor(N/n) tokens, the
    rest one token longer."""
    N = x.shape[1]
    seg = N // n
    assert seg > 0, "num_landmarks must not exceed the sequence length"
    if N % n == 0:
        return x.reshape(x.shape[0], n, seg, -1).mean(dim=-2)
    n_round = n - N % n
    a = x[:, : n_round * seg].reshape(x.shape[0], n_round, seg, -1).mean(dim=-2)
    b = x[:, n_round * seg:].reshape(x.shape[0], n - n_round, seg + 1, -1).mean(dim=-2)
    return torch.cat([a, b], dim=1)


def nystrom_block_attention(q, k, v):
    """What xformers' NystromAttention computes for the reference's 4-D call (see the note above): q, k, v [B, N, h, d] ->
    softmax(q k^T / sqrt d over the h heads of each token) v, [B, N, h, d].  xformers core.py scaled_dot_product_attention: q / sqrt(d) first."""
    att = torch.softmax((q / math.sqrt(k.shape[-1])) @ k.transpose(-2, -1), dim=-1)
    return att @ v


def nystrom_attention(q, k, v, num_landmarks: int = 128):
    """The PUBLISHED Nystrom algorithm per (image, head) over the token axis: q, k, v [BH, N, d] -> [BH, N, d].  Not what the reference's
    layers_8 / layers_4 compute (note above); kept for the kernel tests of the split-key flash attention and the HF triangulation."""
    N, d = k.shape[-2], k.shape[-1]
    if num_landmarks == N:
        return F.scaled_dot_product_attention(q, k, v)
    ql, kl = segment_means(q, num_landmarks), segment_means(k, num_landmarks)
    s = 1.0 / math.sqrt(d)
    k1 = torch.softmax(q @ kl.transpose(-1, -2) * s, dim=-1)          # [N, L]
    k2 = torch.softmax(ql @ kl.transpose(-1, -2) * s, dim=-1)         # [L, L]
    k3 = torch.softmax(ql @ k.transpose(-1, -2) * s, dim=-1) @ v       # [L, d]
    return (k1 @ iterative_pinv(k2)) @ k3


def pos_embed_sine(h: int, w: int, num_pos_feats: int, temperature: float = 10000.0) -> torch.Tensor:
    """PositionEmbeddingSine(num_pos_feats, normalize=True) on an unmasked h x w grid -> [h*w, 2*num_pos_feats] (y block, then x)."""
    eps, scale = 1e-6, 2 * math.pi
    yy = torch.arange(1, h + 1, dtype=torch.float32)[:, None].expand(h, w)
    xx = torch.arange(1, w + 1, dtype=torch.float32)[None, :].expand(h, w)
    yy = yy / (h + eps) * scale
    xx = xx / (w + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)
    px, py = xx[:, :, None] / dim_t, yy[:, :, None] / dim_t
    px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
    py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
    return torch.cat((py, px), dim=2).reshape(h * w, -1)


def generate_rays(K: torch.Tensor, shape):
    """utils/geometric.py:12-52: pixel centres (+0.5), inverse pinhole, unit rays [B,HW,3], angles (theta = atan2(x, z), phi = acos(y))."""
    H, W = shape
    xs = torch.linspace(0, W - 1, W) + 0.5
    ys = torch.linspace(0, H - 1, H) + 0.5
    u = xs.repeat(H, 1).reshape(-1)
    v = ys.repeat(W, 1).t().reshape(-1)
    x = (u[None] - K[:, 0, 2:3]) / K[:, 0, 0:1]
    y = (v[None] - K[:, 1, 2:3]) / K[:, 1, 1:2]
    d = F.normalize(torch.stack([x, y, torch.ones_like(x)], dim=1), dim=1).permute(0, 2, 1)
    return d, torch.stack([torch.atan2(d[..., 0], d[..., 2]), torch.acos(d[..., 1])], dim=-1)


def flat_interpolate(t: torch.Tensor, old, new, antialias: bool = True) -> torch.Tensor:
    if tuple(old) == tuple(new):
        return t
    x = t.view(t.shape[0], old[0], old[1], -1).permute(0, 3, 1, 2)
    x = F.interpolate(x, size=tuple(new), mode="bilinear", align_corners=False, antialias=antialias)
    return x.reshape(t.shape[0], -1, new[0] * new[1]).permute(0, 2, 1).contiguous()


def v1_shapes(image_shape, network_shape):
    """unidepthv1.py:38-47 _shapes and :29-35 _paddings."""
    h, w = image_shape
    if network_shape[1] / network_shape[0] > w / h:
        ratio = network_shape[0] / h
    else:
        ratio = network_shape[1] / w
    nh, nw = math.ceil(h * ratio - 0.5), math.ceil(w * ratio - 0.5)
    H, W = network_shape
    pt, pb = (H - nh) // 2, H - nh - (H - nh) // 2
    pl, pr = (W - nw) // 2, W - nw - (W - nw) // 2
    return (nh, nw), ratio, (pl, pr, pt, pb)


VIT = {"dinov2_vitl14": (1024, 24, 16, [5, 12, 18, 24])}       # models/encoder.py:171-186


class OracleV1(OracleConvNeXt):
    def __init__(self, config: dict, state_dict: dict):
        name = config["model"]["pixel_encoder"]["name"]
        self.vit = VIT.get(name)
        if self.vit is None:
            super().__init__(config, state_dict)
        else:               # UniDepthV1 on DINOv2 ViT-L/14 (configs/config_v1_vitl14.json)
            self.p = "pixel_encoder."
            self.a = {"output_idx": list(config["model"]["pixel_encoder"].get("output_idx", self.vit[3]))}
            self.taps, self.keep_taps = {}, False
        self.w = {k: v.detach().to(torch.float32).cpu() for k, v in state_dict.items()}
        self.cfg = config
        self.C = config["model"]["pixel_decoder"]["hidden_dim"]
        self.heads = config["model"]["num_heads"]
        self.dec_depths = list(config["model"]["pixel_decoder"]["depths"])
        self.image_shape = list(config["data"]["image_shape"])

    # ---- DINOv2 encoder as UniDepthV1 builds it (unidepthv1.py:412-421: interpolate_offset = 0.1, use_norm False -> NO final LayerNorm,
    # every block's output is returned; backbones/dinov2.py:267-347) followed by unidepthv1.py:324-328: patch tokens + class token
    def _vit_pos_embed(self, h, w):
        pe = self.w["pixel_encoder.pos_embed"]
        N = pe.shape[1] - 1
        M = int(math.sqrt(N))
        if h * w == N and h == w:
            return pe
        D = pe.shape[-1]
        grid = pe[:, 1:].reshape(1, M, M, D).permute(0, 3, 1, 2)
        # interpolate_offset != 0: scale factors (h + 0.1) / M instead of an output size (dinov2.py:283-296, "historical kludge")
        grid = F.interpolate(grid, scale_factor=(float(h + 0.1) / M, float(w + 0.1) / M), mode="bicubic", antialias=False)
        assert tuple(grid.shape[-2:]) == (h, w)
        return torch.cat([pe[:, :1], grid.permute(0, 2, 3, 1).reshape(1, -1, D)], dim=1)

    def encode(self, image):
        if self.vit is None:
            return super().encode(image)
        D, depth, heads, _ = self.vit
        w, pe = self.w, "pixel_encoder."
        B, _, Hn, Wn = image.shape
        h, wd = Hn // 14, Wn // 14
        x = F.conv2d(image, w[pe + "patch_embed.proj.weight"], w[pe + "patch_embed.proj.bias"], stride=14).flatten(2).transpose(1, 2)
        x = torch.cat([w[pe + "cls_token"].expand(B, -1, -1), x], dim=1) + self._vit_pos_embed(h, wd)
        outs, cls = [], []
        for i in range(depth):
            b = f"{pe}blocks.{i}"
            y = F.layer_norm(x, (D,), w[b + ".norm1.weight"], w[b + ".norm1.bias"], 1e-6)
            qkv = F.linear(y, w[b + ".attn.qkv.weight"], w[b + ".attn.qkv.bias"])
            N = qkv.shape[1]
            qkv = qkv.reshape(B, N, 3, heads, D // heads).permute(2, 0, 3, 1, 4)
            o = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2]).transpose(1, 2).reshape(B, N, D)
            x = x + F.linear(o, w[b + ".attn.proj.weight"], w[b + ".attn.proj.bias"]) * w[b + ".ls1.gamma"]
            y = F.layer_norm(x, (D,), w[b + ".norm2.weight"], w[b + ".norm2.bias"], 1e-6)
            y = F.linear(F.gelu(F.linear(y, w[b + ".mlp.fc1.weight"], w[b + ".mlp.fc1.bias"])), w[b + ".mlp.fc2.weight"], w[b + ".mlp.fc2.bias"])
            x = x + y * w[b + ".ls2.gamma"]
            c = x[:, :1]
            cls.append(c.contiguous())
            outs.append((x[:, 1:].reshape(B, h, wd, D) + c.unsqueeze(1)).contiguous())       # unidepthv1.py:324-328
        return outs, cls

    # ---- small blocks
    def _ln(self, x, name, eps=1e-5):
        return F.layer_norm(x, (x.shape[-1],), self.w[name + ".weight"], self.w[name + ".bias"], eps)

    def _lin(self, x, name):
        return F.linear(x, self.w[name + ".weight"], self.w.get(name + ".bias"))

    def _mlp(self, x, p):
        return self._lin(F.gelu(self._lin(self._ln(x, p + "norm"), p + "proj1")), p + "proj2")

    def _attn_block(self, x, p, heads, context=None, pos_embed=None, pos_embed_context=None, nystrom=False):
        ctx = x if context is None else context
        xn, cn = self._ln(x, p + "norm_attnx"), self._ln(ctx, p + "norm_attnctx")
        B, N, C = xn.shape
        d = C // heads
        kv = self._lin(cn, p + "kv")
        k, v = kv[..., :C], kv[..., C:]                    # "b n (kv h d)": K rows first, heads-major inside
        q = self._lin(xn, p + "q")
        if pos_embed is not None:
            q = q + pos_embed
        if pos_embed_context is not None:
            k = k + pos_embed_context

        def split(t):
            return t.reshape(t.shape[0], t.shape[1], heads, d).permute(0, 2, 1, 3)
        if nystrom:
            # layers/nystrom_attention.py:59-62,81: "b n (h d) -> b n h d" into xformers NystromAttention, read back as "b n h d"
            o = nystrom_block_attention(q.reshape(B, N, heads, d), k.reshape(B, -1, heads, d), v.reshape(B, -1, heads, d)).reshape(B, N, C)
        else:
            q, k, v = split(q), split(k), split(v)
            o = F.scaled_dot_product_attention(q, k, v).permute(0, 2, 1, 3).reshape(B, N, C)
        o = self._lin(o, p + "out")
        x = o * self.w[p + "ls1.gamma"] + x
        return self._mlp(x, p + "mlp.") * self.w[p + "ls2.gamma"] + x

    def _cvnxt(self, x, p):
        C = x.shape[1]
        y = F.conv2d(x, self.w[p + "dwconv.weight"], self.w[p + "dwconv.bias"], padding=3, groups=C).permute(0, 2, 3, 1)
        y = self._lin(F.gelu(self._lin(self._ln(y, p + "norm"), p + "pwconv1")), p + "pwconv2") * self.w[p + "gamma"]
        return x + y.permute(0, 3, 1, 2)

    def _conv_upsample(self, x, p):
        for c in range(2):
            x = self._cvnxt(x, f"{p}convs.{c}.")
        x = F.conv2d(x, self.w[p + "up.0.weight"], self.w[p + "up.0.bias"])
        x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
        x = F.conv2d(x, self.w[p + "up.2.weight"], self.w[p + "up.2.bias"], padding=1)
        return x.flatten(2).permute(0, 2, 1)                # b (h w) c

    def _adapter(self, x, p):
        return F.gelu(self._lin(self._ln(x, p + ".0"), p + ".1"))

    # ---- decoder
    def decode(self, outs, cls, H, W, rays_gt=None, K_gt=None, skip_camera=False):
        w, C = self.w, self.C
        pd = "pixel_decoder."
        B = outs[0].shape[0]
        feats = self.stage_features(outs)
        toks = [cls[-i - 1] for i in range(4)]
        res = [tuple(sorted([f.shape[1], f.shape[2]])) for f in feats]
        level_shapes = sorted(set(res))[::-1]
        if len(level_shapes) == 1:                         # ViT: one resolution for all levels (decoder.py:391-392)
            level_shapes = level_shapes * 4
        assert len(level_shapes) == 4
        common = level_shapes[-2]
        flat = [flat_interpolate(f.reshape(B, -1, f.shape[-1]), level_shapes[i], common) for i, f in enumerate(feats)]
        features = [self._adapter(x, f"{pd}input_adapter.input_adapters.{j}") for j, x in enumerate(flat)]      # 4 x [B, hw, C]
        hw = common[0] * common[1]
        lvl = self._ln(self._lin(F.gelu(self._lin(w[pd + "level_embeds"], pd + "level_embed_layer.0")), pd + "level_embed_layer.2"),
                       pd + "level_embed_layer.3")
        level_embed = torch.cat([lvl[i:i + 1].unsqueeze(0).repeat(B, hw, 1) for i in range(4)], dim=1)
        pos = pos_embed_sine(common[0], common[1], C // 2)[None].repeat(B, 4, 1)
        self.taps_v1 = {"features": features}
        if skip_camera:
            K, rays = K_gt, rays_gt
        else:
            ct = torch.cat([self._adapter(t, f"{pd}token_adapter.input_adapters.{j}") for j, t in enumerate(toks)], dim=1)   # [B,4,C]
            K = self._camera_head(features, ct, pos + level_embed)
            K = K.clone()
            K[:, 0, 0] = max(H, W) / 2 * K[:, 0, 0]
            K[:, 1, 1] = max(H, W) / 2 * K[:, 1, 1]
            K[:, 0, 2] = K[:, 0, 2] * W
            K[:, 1, 2] = K[:, 1, 2] * H
            rays = rays_gt if rays_gt is not None else generate_rays(K, (H, W))[0]
        outs_ms, depth_features = self._depth_head(features, rays, pos, level_embed, common, (H, W))
        return K, outs_ms, depth_features

    def _camera_head(self, features, cls_tokens, pos_embed):
        cl = "pixel_decoder.camera_layer."
        w = self.w
        c = self._ln(cls_tokens, cl + "cls_project.0")
        c = self._lin(F.gelu(self._lin(c, cl + "cls_project.1")), cl + "cls_project.3")
        fs = torch.cat(features, dim=1) + pos_embed
        lat = w[cl + "latents_pos"].expand(c.shape[0], -1, -1)
        fs = self._mlp(fs, cl + "in_features.")
        ctx = torch.cat((fs, c), dim=1)
        c = self._attn_block(c, cl + "aggregate.", 1, context=ctx, pos_embed=lat)
        for i in range(2):
            c = self._attn_block(c, f"{cl}layers.{i}.", self.heads, pos_embed=lat)
        x = self._mlp(c, cl + "out.").squeeze(-1)
        K = torch.zeros(x.shape[0], 3, 3)
        K[:, 0, 0], K[:, 1, 1] = x[:, 0].exp(), x[:, 1].exp()
        K[:, 0, 2], K[:, 1, 2] = x[:, 2].sigmoid(), x[:, 3].sigmoid()
        K[:, 2, 2] = 1.0
        return K

    def _depth_head(self, features, rays_hr, pos, level_embed, shapes, original):
        dl = "pixel_decoder.depth_layer."
        C, heads = self.C, self.heads
        B = features[0].shape[0]
        h, w_ = shapes
        emb = []
        for mult, nm in ((1, "project_rays16"), (2, "project_rays8"), (4, "project_rays4")):
            r = F.normalize(flat_interpolate(rays_hr, original, (h * mult, w_ * mult)), dim=-1)
            emb.append(self._mlp(real_sh_deg8(r), f"{dl}{nm}."))
        e16, e8, e4 = emb
        T = self.taps_v1
        T["rays_embedding_16"] = e16
        tokens = torch.cat(features, dim=1)
        f16 = self._lin(torch.cat(features, dim=-1), dl + "features_channel_cat")
        lat16 = self._mlp(f16, dl + "to_latents.")
        T["to_latents"] = lat16
        lat16 = self._attn_block(lat16, dl + "aggregate_16.", 1, context=tokens, pos_embed_context=pos + level_embed)
        T["aggregate_16"] = lat16
        lat16 = self._attn_block(lat16, dl + "prompt_camera.", 1, context=e16)
        T["prompt_camera"] = lat16
        for i in range(self.dec_depths[0]):
            lat16 = self._attn_block(lat16, f"{dl}layers_16.{i}.", heads, pos_embed=e16)
        self.taps_v1["latents_16"] = lat16

        def nchw(t, hh, ww):
            return t.permute(0, 2, 1).reshape(B, -1, hh, ww).contiguous()
        lat8 = self._conv_upsample(nchw(lat16 + e16, h, w_), dl + "up8.")
        T["up8"] = lat8
        out8 = F.conv2d(nchw(lat8, 2 * h, 2 * w_), self.w[dl + "out8.weight"], self.w[dl + "out8.bias"], padding=1)
        for i in range(self.dec_depths[1]):
            lat8 = self._attn_block(lat8, f"{dl}layers_8.{i}.", heads // 2, pos_embed=e8, nystrom=True)
        T["layers_8"] = lat8
        lat4 = self._conv_upsample(nchw(lat8 + e8, 2 * h, 2 * w_), dl + "up4.")
        T["up4"] = lat4
        out4 = F.conv2d(nchw(lat4, 4 * h, 4 * w_), self.w[dl + "out4.weight"], self.w[dl + "out4.bias"], padding=1)
        for i in range(self.dec_depths[2]):
            lat4 = self._attn_block(lat4, f"{dl}layers_4.{i}.", heads // 4, pos_embed=e4, nystrom=True)
        T["layers_4"] = lat4
        lat2 = self._conv_upsample(nchw(lat4 + e4, 4 * h, 4 * w_), dl + "up2.")
        T["up2"] = lat2
        out2 = F.conv2d(nchw(lat2, 8 * h, 8 * w_), self.w[dl + "out2.weight"], self.w[dl + "out2.bias"], padding=1)
        ms = [o.clamp(-10.0, 10.0).exp() for o in (out8, out4, out2)]
        T["out8"], T["out4"], T["out2"] = [m[:, 0] for m in ms]
        return ms, nchw(lat16, h, w_)

    # ---- infer (unidepthv1.py:288-373)
    def infer(self, rgbs: torch.Tensor, intrinsics=None, skip_camera: bool = False):
        if rgbs.ndim == 3:
            rgbs = rgbs.unsqueeze(0)
        if intrinsics is not None and intrinsics.ndim == 2:
            intrinsics = intrinsics.unsqueeze(0)
        B, _, H, W = rgbs.shape
        x = rgbs
        if x.max() > 5 or x.dtype == torch.uint8:
            x = x.to(torch.float32).div(255)
        if x.min() >= 0.0 and x.max() <= 1.0:
            x = (x - torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)) / torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
        (h, w), ratio, pads = v1_shapes((H, W), self.image_shape)
        pl, pr, pt, pb = pads
        x = F.interpolate(x, size=(h, w), mode="bilinear", align_corners=False, antialias=True)
        x = F.pad(x, (pl, pr, pt, pb), mode="constant")
        gtK = None
        if intrinsics is not None:
            gtK = intrinsics.clone().float()
            gtK[:, 0, 0] *= ratio; gtK[:, 1, 1] *= ratio
            gtK[:, 0, 2] = gtK[:, 0, 2] * ratio + pl
            gtK[:, 1, 2] = gtK[:, 1, 2] * ratio + pt
        outs, cls = self.encode(x)
        Hn, Wn = self.image_shape
        rays_gt = generate_rays(gtK, (Hn, Wn))[0] if gtK is not None else None
        skip = bool(skip_camera and gtK is not None)
        K, ms, _ = self.decode(outs, cls, Hn, Wn, rays_gt=rays_gt, K_gt=gtK, skip_camera=skip)
        pred = sum(F.interpolate(m, size=(Hn, Wn), mode="bilinear", align_corners=False, antialias=True) for m in ms) / len(ms)
        pred = pred[..., pt: Hn - pb, pl: Wn - pr]
        pred = F.interpolate(pred, size=(H, W), mode="bilinear", align_corners=False, antialias=True)
        K = K.clone()
        K[:, 0, 0] /= ratio; K[:, 1, 1] /= ratio
        K[:, 0, 2] = (K[:, 0, 2] - pl) / ratio
        K[:, 1, 2] = (K[:, 1, 2] - pt) / ratio
        # NB reference quirk (unidepthv1.py:343-360): with GT intrinsics the back-projection uses the NETWORK-resolution matrix on the
        # ORIGINAL pixel grid -- except with skip_camera, where the "predicted" matrix IS the GT tensor and _postprocess rescaled it in place
        Kuse = K if (skip or gtK is None) else gtK
        ang = generate_rays(Kuse, (H, W))[1].reshape(B, H, W, 2)
        theta, phi, z = ang[..., 0], ang[..., 1], pred[:, 0]
        pts = torch.stack((z * torch.tan(theta), z / torch.tan(phi) / torch.cos(theta), z), dim=1)
        return {"intrinsics": K, "points": pts, "depth": pred[:, -1:]}


IMAGENET_MEAN = (0.485, 0.456, 0.406)     # unidepth/utils/constants.py:12-13
IMAGENET_STD = (0.229, 0.224, 0.225)
