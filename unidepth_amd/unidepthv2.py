"""UniDepthV2 on MI355X: same Python surface as the reference model wrapper
(unidepth/models/unidepthv2/unidepthv2.py:111-467), all device arithmetic in libunidepth_hip.so.

    model = UniDepthV2.from_pretrained(dir_or_repo).to("cuda").eval()
    model.resolution_level = 9            # optional, as in the reference (unidepthv2.py:252-260)
    out = model.infer(rgb, camera=None, normalize=True)
    # -> dict(confidence, intrinsics, radius, depth, points, rays, depth_features)   (unidepthv2.py:311-339)

How a call runs: the shape policy (pure integer/float host logic, unidepthv2.py:36-77) picks the network
resolution; a *launch program* for (batch, input shape, camera batch) is recorded once -- ~260 kernel descriptors
over a fixed set of device buffers -- and replayed by one C call (ops.Program / csrc/program.cpp); only the output
assembly kernel is issued per call because it writes into fresh caller-owned tensors.
"""
from __future__ import annotations

import collections
import json
import math
import os
import warnings
from typing import Optional

import torch

from . import ops
from .ops import (UD_A_CONV3_REFLECT, UD_A_CONV3_REFLECT_UP, UD_A_CONV3_ZERO, UD_ACT_GELU, UD_ACT_LRELU, UD_ACT_NONE, UD_EPI_D2S, UD_EPI_F16,
                  UD_EPI_F32, UD_EPI_HEAD, UD_EPI_QKV)
from .cameras import GT_OPENCV, GT_PINHOLE, BatchCamera, as_camera

GT_GIVEN_RAYS = 15          # plan tag: the ray map itself is supplied (pixel_decoder seam, decoder.py:400 `rays_gt`)
from .module import EngineModule
from .weights import arch_of, pack

IMAGENET_DATASET_MEAN = (0.485, 0.456, 0.406)      # unidepth/utils/constants.py:12
IMAGENET_DATASET_STD = (0.229, 0.224, 0.225)       # unidepth/utils/constants.py:13


# ------------------------------------------------------------------------------------------ shape policy (host)
def get_paddings(original_shape, aspect_ratio_range):
    """Aspect-ratio padding policy; mirrors unidepthv2.py:36-58 (returns (l, r, t, b), (H, W))."""
    H, W = original_shape
    ratio = W / H
    lo, hi = aspect_ratio_range
    target = min(hi, max(lo, ratio))
    if ratio > target:                     # too wide -> pad rows
        Hn = int(W / target)
        top = (Hn - H) // 2
        return (0, 0, top, Hn - H - top), (Hn, W)
    Wn = int(H * target)                   # too tall -> pad columns
    left = (Wn - W) // 2
    return (left, Wn - W - left, 0, 0), (H, Wn)


def get_resize_factor(original_shape, pixels_range, shape_multiplier=14):
    """Area clamp + round up to the patch multiple; mirrors unidepthv2.py:61-77."""
    H, W = original_shape
    n = H * W
    lo, hi = pixels_range
    rf = (min(hi, max(lo, n)) / n) ** 0.5
    new_w, new_h = int(W * rf), int(H * rf)
    return rf, (math.ceil(new_h / shape_multiplier) * shape_multiplier, math.ceil(new_w / shape_multiplier) * shape_multiplier)


def _rup(x, m):
    return (x + m - 1) // m * m


class _Plan:
    """Device buffers + recorded launch program for one (batch, image shape, camera batch, dtype) signature."""

    def __init__(self, model: "UniDepthV2", B, H, W, cam_nb, is_u8, normalize, pixels_bounds, gt_mode=1, net=False):
        w, a, dev = model._w, model._arch, model.device
        meta = w["meta"]
        D, C, heads, Hd = a["D"], a["C"], a["heads"], a["dec_heads"]
        sc = model.shape_constraints
        self.B, self.H, self.W = B, H, W
        if net:                            # module-seam plans (pixel_encoder / pixel_decoder): the input IS the network image
            assert H % 14 == 0 and W % 14 == 0, "network-resolution image: H and W must be multiples of the patch size (14)"
            self.paddings, (self.Hp, self.Wp) = (0, 0, 0, 0), (H, W)
            self.rf, (Hn, Wn) = 1.0, (H, W)
        else:
            self.paddings, (self.Hp, self.Wp) = get_paddings((H, W), sc["ratio_bounds"])
            self.rf, (Hn, Wn) = get_resize_factor((self.Hp, self.Wp), pixels_bounds)
        pl, pr, pt, pb = self.paddings
        self.tap_points = []               # (name, number of ops after which it is valid, getter -> tensor in the reference's layout)

        def tap(name, fn):
            self.tap_points.append((name, len(self.prog), fn))
        self.Hn, self.Wn = Hn, Wn
        h, wg = Hn // 14, Wn // 14
        self.h, self.w = h, wg
        hw = h * wg
        N = hw + 1
        Np = _rup(N, 16)              # token rows per image (encoder stream; 16: V^T block order of the QKV epilogue)
        hwp = _rup(hw, 8)             # token rows per image (decoder streams)
        Nkp = _rup(N, 64)
        hwkp = _rup(hw, 64)
        self.cam_nb = cam_nb
        nb = cam_nb if cam_nb else B  # batch of the ray tensors (a single GT camera broadcasts: decoder.py:400)
        f16, f32 = torch.float16, torch.float32

        def z(*shape, dtype=f16):
            return torch.zeros(*shape, dtype=dtype, device=dev)

        P = ops.Program()
        self.prog = P
        wname = {id(v): k for k, v in w.items() if torch.is_tensor(v)}
        _gemm = P.gemm

        def tagged_gemm(**kw):
            if "tag" not in kw and id(kw.get("W")) in wname:
                kw["tag"] = wname[id(kw["W"])]
            return _gemm(**kw)
        P.gemm = tagged_gemm
        zeros = z(256)
        # ---------------- inputs
        self.rgb = torch.zeros(B, 3, H, W, dtype=torch.uint8 if is_u8 else f32, device=dev)
        patches = z(B * hw, 640)
        P.preprocess(rgb=self.rgb, patches=patches, B=B, H=H, W=W, pad_l=pl, pad_t=pt, Hp=self.Hp, Wp=self.Wp, Hn=Hn, Wn=Wn,
                     ldp=640, is_u8=int(is_u8), normalize=int(normalize), mean=IMAGENET_DATASET_MEAN,
                     inv_std=tuple(1.0 / s for s in IMAGENET_DATASET_STD))
        # ---------------- encoder (dinov2.py:306-347; block.py:84-109; attention.py:51-62; mlp.py:35-41)
        pos = model._pos_embed(h, wg).to(dev)                                # [N, D] fp32 (bicubic-resampled per grid)
        cls_row = (w["host.cls_token"] + pos[0].cpu()).to(dev)
        x = z(B * Np, D, dtype=f32)
        M = B * Np
        P.gemm(A=patches, W=w["patch.w"], bias=w["patch.b"], out=x, add=pos, M=B * hw, N=D, K=640, lda=640, ldw=640, ldc=D,
               ldadd=D, epi=UD_EPI_F32, rows_in=hw, rows_out=Np, row_off=1, add_row_off=1)
        P.fill_rows(x, cls_row, B, Np, 0, D, D)
        tap("tokens0", lambda: x.view(B, Np, D)[:, :N].clone())                 # cls + patches + pos_embed (dinov2.py:306-322)
        xn = z(M, D)
        vt = z(B, heads, 64, Nkp)
        hid = z(M, 4 * D)
        # Q|K and the attention output live inside `hid`: between fc2 of block i and fc1 of block i+1 the hidden activations are dead, and
        # q|k / ao are dead while fc1 / fc2 run.  The block's working set drops from 272 MB to 204 MB at bs = 8 (ViT-L) -- under the 256 MB
        # Infinity Cache, so what one launch writes the next one reads on-die (tools/r4_insitu.py: the step's launches ran 36 us per block
        # behind the same launches on warm operands).  V^T keeps its own buffer: its pad columns must stay zero.
        flat = hid.view(-1)
        qk = flat[: M * 2 * D].view(M, 2 * D)
        ao = flat[M * 2 * D: M * 3 * D].view(M, D)
        featn_all = z(4, B * hwp, D)                                  # stacked: the 4 levels are processed by grouped launches
        featn = [featn_all[j] for j in range(4)]
        clsn = [z(_rup(B, 8), D, dtype=f32) for _ in range(4)]       # final-LN'd cls tokens stay fp32: they feed the fp32 camera head
        self.enc_first = len(P)
        lvl = 0
        # LayerNorm folded into the neighbouring GEMMs (UdGemm.row_stats_out / row_stats_in): proj / fc2 write the raw fp16 copy of the
        # residual stream and per-row partial sums with their fp32 accumulate, qkv / fc1 normalise in their epilogues -- no LayerNorm
        # launch, no second pass over x.  Only where all four GEMMs run on the large-tile kernel (its epilogues hold the statistics
        # code): bs >= 4 or so for ViT-L; smaller problems keep the LayerNorm kernel ...
        slabs = D // 64
        x16 = z(M, D)
        rpart = z(M, slabs, 2, dtype=f32)                           # per 64-column slab (sum, sum of squares) written by proj / fc2
        rstats = z(M, 2, dtype=f32)                                 # (rstd, -mean * rstd) per row, reduced by the producer's last workgroup per row tile
        rticket = torch.zeros(2, M // 128 + 2, dtype=torch.int32, device=dev)     # one set per producer (proj, fc2): a set counts arrivals of ONE tiling

        def _pick(**kw):
            import ctypes as _C
            return ops.lib.ud_gemm_pick(_C.byref(ops.mk(ops.UdGemm, **kw)))
        big = all(_pick(A=xn, W=w[f"enc.0.{nm}.w"], out=xn, M=M, N=n_, K=k_, lda=k_, ldw=k_, ldc=n_, epi=e_, vsplit=2 * D, tok_per_img=Np,
                        kv_ld=Nkp, heads_v=heads, out2=vt, accumulate=int(e_ == UD_EPI_F32),
                        **(dict(row_stats_in=rstats, wsum=w[f"enc.0.{nm}.wsum"]) if nm in ("qkv", "fc1") else {})) & 15 in (3, 4, 8)
                  for nm, n_, k_, e_ in (("qkv", 3 * D, D, UD_EPI_QKV), ("proj", D, D, UD_EPI_F32), ("fc1", 4 * D, D, UD_EPI_F16), ("fc2", D, 4 * D, UD_EPI_F32)))
        # ... and where the producers (proj / fc2, N = D) run about one tile per workgroup: measured on one box (gpurun_out r3c12), bs = 8:
        # +2.5 %, 644x966 bs = 4 (264 tiles): +0.3 %, bs = 16 (460 tiles): +-0, bs = 32 (916 tiles): -1.2 % -- with several tiles per
        # workgroup the per-tile drain + ticket of the in-kernel reduction sits inside the tile stream, and the LayerNorm kernels it
        # replaces are efficient HBM streams at that size.  model.ln_fold_force (tests): True forces it on, False off.
        prod_tiles = -(-M // 192) * -(-D // 256)
        force = getattr(model, "ln_fold_force", None)
        fold = big and force is not False and (prod_tiles <= 320 or force is True)
        self.ln_fold = fold
        self.row_tickets = rticket          # tests: every completed launch leaves its ticket set at zero
        lnc = dict(row_stats_in=rstats, ln_slabs=slabs, ln_D=D, ln_eps=1e-6)
        for i in range(a["depth"]):
            if fold and i > 0:
                P.gemm(A=x16, W=w[f"enc.{i}.qkv.w"], bias=w[f"enc.{i}.qkv.b"], out=qk, out2=vt, M=M, N=3 * D, K=D, lda=D, ldw=D,
                       ldc=2 * D, epi=UD_EPI_QKV, vsplit=2 * D, tok_per_img=Np, kv_ld=Nkp, heads_v=heads, tag="enc.qkv",
                       flops=2.0 * B * N * 3 * D * D, wsum=w[f"enc.{i}.qkv.wsum"], **lnc)
            else:
                P.layernorm(x=x, y=xn, rows=M, D=D, ldx=D, ldy=D, eps=1e-6, rows_per_img=M, in_rows_per_img=M, out_rows_per_img=M, tag="enc.ln")
                P.gemm(A=xn, W=w[f"enc.{i}.qkv.w"], bias=w[f"enc.{i}.qkv.b"], out=qk, out2=vt, M=M, N=3 * D, K=D, lda=D, ldw=D,
                       ldc=2 * D, epi=UD_EPI_QKV, vsplit=2 * D, tok_per_img=Np, kv_ld=Nkp, heads_v=heads, tag="enc.qkv",
                       flops=2.0 * B * N * 3 * D * D)
            if i == 0:
                def _qkv0():                                                    # [B, N, 3D] = Q | K | V (attention.py:53-55 layout)
                    cols = ((torch.arange(N) & ~15) | ((torch.arange(N) & 4) << 1) | ((torch.arange(N) & 8) >> 1) | (torch.arange(N) & 3)).to(dev)
                    v = vt[:, :, :, cols].permute(0, 3, 1, 2).reshape(B, N, D)
                    qkv = torch.cat([qk.view(B, Np, 2 * D)[:, :N], v], dim=2).float()
                    qkv[..., :D] /= (D // heads) ** -0.5 * 1.4426950408889634      # Q is stored pre-scaled for the attention kernel
                    return qkv
                tap("blocks.0.attn.qkv", _qkv0)
            P.attention(Q=qk, K=qk.data_ptr() + D * 2, Vt=vt, O=ao, B=B, H=heads, Nq=N, Nk=N, ldq=2 * D, ldk=2 * D, ldo=D,
                        kv_ld=Nkp, q_rows_per_img=Np, k_rows_per_img=Np, scale=(D // heads) ** -0.5, q_prescaled=1, tag="enc.attn")
            prod = dict(out2=x16, ldc2=D, row_stats_out=rpart, row_stats_final=rstats, row_stats_ticket=rticket[0], ln_D=D, ln_eps=1e-6) if fold else {}
            P.gemm(A=ao, W=w[f"enc.{i}.proj.w"], bias=w[f"enc.{i}.proj.b"], out=x, M=M, N=D, K=D, lda=D, ldw=D, ldc=D,
                   epi=UD_EPI_F32, accumulate=1, tag="enc.proj", flops=2.0 * B * N * D * D, **prod)
            if fold:
                P.gemm(A=x16, W=w[f"enc.{i}.fc1.w"], bias=w[f"enc.{i}.fc1.b"], out=hid, M=M, N=4 * D, K=D, lda=D, ldw=D, ldc=4 * D,
                       epi=UD_EPI_F16, act=UD_ACT_GELU, tag="enc.fc1", flops=8.0 * B * N * D * D, wsum=w[f"enc.{i}.fc1.wsum"], **lnc)
            else:
                P.layernorm(x=x, y=xn, rows=M, D=D, ldx=D, ldy=D, eps=1e-6, rows_per_img=M, in_rows_per_img=M, out_rows_per_img=M, tag="enc.ln")
                P.gemm(A=xn, W=w[f"enc.{i}.fc1.w"], bias=w[f"enc.{i}.fc1.b"], out=hid, M=M, N=4 * D, K=D, lda=D, ldw=D, ldc=4 * D,
                       epi=UD_EPI_F16, act=UD_ACT_GELU, tag="enc.fc1", flops=8.0 * B * N * D * D)
            last = i == a["depth"] - 1
            prod2 = dict(prod, row_stats_ticket=rticket[1]) if (fold and not last) else {}            # nothing consumes the last block's raw copy
            P.gemm(A=hid, W=w[f"enc.{i}.fc2.w"], bias=w[f"enc.{i}.fc2.b"], out=x, M=M, N=D, K=4 * D, lda=4 * D, ldw=4 * D, ldc=D,
                   epi=UD_EPI_F32, accumulate=1, tag="enc.fc2", flops=8.0 * B * N * D * D, **prod2)
            if i in (0, 5, 11, 17, 23) or i == a["depth"] - 1:
                tap(f"block{i}", lambda: x.view(B, Np, D)[:, :N].clone())      # residual stream after block i
            if (i + 1) in a["output_idx"]:
                # final LayerNorm (eps 1e-5, dinov2.py:254) only on the 4 consumed outputs; patch rows and cls row separately
                # (one launch: the class-token row in front of an image's patch rows goes out as fp32, UdLayerNorm.cls_y -- it was a launch of B rows)
                P.layernorm(x=x, y=featn[lvl], rows=B * (hw + 1), D=D, ldx=D, ldy=D, eps=1e-5, rows_per_img=hw, in_rows_per_img=Np,
                            in_row_off=1, out_rows_per_img=hwp, out_row_off=0, cls_y=clsn[lvl], ldcls=D)
                lvl += 1
        self.enc_last = len(P)
        self.x, self.featn, self.clsn = x, featn, clsn

        # ---------------- decoder: adapters (decoder.py:418,434-435)
        Md = B * hwp
        feat_all = z(4, Md, C, dtype=f32)
        ct = z(B * 4, C, dtype=f32)
        # The camera branch (4 token adapters, the fp32 camera head, intrinsics, rays, ray embedding: ~30 dependent launches of a few workgroups
        # each, 0.35-0.45 ms) and the feature branch (grouped adapter GEMM, LayerNorm, the q projection of the four cross-attention blocks) are
        # independent until the K / V projection of the ray embedding.  Round 4 ran them on two streams (fork / join by events): same bits,
        # one-call p50 14.22 ms on one stream against 14.25 ms forked (profiles/r04_side_branch_ab.txt) -- the feature-branch launches fill every
        # CU's LDS, the camera kernels wait for them instead of running beside them.  One stream; the mechanism was removed in round 5.
        # Round 6 re-measured it with the camera head as ONE 128-workgroup launch (side section of the launch program, fork / join by events):
        # p50 14.06 vs 14.01 ms at bs 8, 4.84 vs 4.67 ms at bs 1 (profiles/r06_side_section_ab.txt) -- the spinning grid holds half of the CUs and
        # the two cross-stream waits cost more than the overlap hides.  Dropped again.
        self.dec_first = self.enc_last

        def feature_branch_head():
            P.gemm(A=featn_all, W=w["dec.adapterg.w"], bias=w["dec.adapterg.b"], out=feat_all, M=Md, N=C, K=D, lda=D, ldw=D, ldc=C,
                   epi=UD_EPI_F32, groups=4, gA=Md * D, gW=C * D, gBias=C, gOut=Md * C, tag="dec.adapters(x4)",
                   )
            for j in range(4):
                tap(f"input_adapter.{j}", lambda j=j: feat_all[j].view(B, hwp, C)[:, :hw].clone())
        feature_branch_head()
        # ---- camera token adapters + camera head (decoder.py:34-45,48-114) on the 4 camera tokens per image: an fp32 island (UdLinearF32
        # explains why).  ONE launch (UdCameraHead: a persistent grid walks the ~18 dependent layers as phases between grid barriers) where
        # the kernel's limits allow it, otherwise the per-layer launches it replaces (same arithmetic, ~26 launches).
        Mc = B * 4
        ch = z(Mc, 4 * C, dtype=f32); cqkv = z(Mc, 3 * C, dtype=f32); cao = z(Mc, C, dtype=f32); t = z(Mc, C, dtype=f32); raw = z(Mc, 1, dtype=f32)
        scale_d = meta["hd"] ** -0.5
        self.cam_sync = z(16, dtype=torch.int32)
        # host-mapped word the one-launch camera head sets (system-scope store) when one of its grid barriers times out: polled by the next
        # infer() without a device synchronisation (UniDepthV2._check_camera_head); the kernel also turns that call's camera parameters into NaN
        self.cam_fail = torch.zeros(1, dtype=torch.int32).pin_memory() if dev.type == "cuda" else None

        def lin(x, pre, out, M, N, K, ldx, ldc, **kw):
            d = dict(x=x, W=w[pre + ".w"], out=out, M=M, N=N, K=K, ldx=ldx, ldc=ldc, kind=0, sync=1)
            if kw.pop("bias", True):
                d["bias"] = w[pre + ".b"]
            d.update(kw)
            return d

        phases = [lin(clsn[j], f"dec.camadapter.{j}", ct.data_ptr() + j * C * 4, B, C, D, D, 4 * C, sync=int(j == 3)) for j in range(4)]

        def mlp_phases(pre, src, dst, accumulate, n_out=C):
            nh = w[pre + "fc1.w"].shape[0]
            return [lin(src, pre + "fc1", ch, Mc, nh, C, C, 4 * C, ln=1, act=UD_ACT_GELU),
                    lin(ch, pre + "fc2", dst, Mc, n_out, nh, 4 * C, n_out if n_out == 1 else C, accumulate=accumulate)]

        phases += mlp_phases("cam.project.", ct, t, 0)
        for blk in ("cam.agg1.", "cam.agg2."):
            phases += [lin(t, blk + "qkv", cqkv, Mc, 3 * C, C, C, 3 * C, ln=1, add=w["cam.pos"], ldadd=C, add_mod=4, add_cols=C),   # norm_attnx / norm_attnctx share statistics
                       dict(x=cqkv, out=cao, M=Mc, ldx=3 * C, ldc=C, kind=1, sync=1),
                       lin(cao, blk + "out", t, Mc, C, C, C, C, accumulate=1, bias=False)]
            phases += mlp_phases(blk, t, t, 1)
        phases += mlp_phases("cam.out.", t, raw, 0, n_out=1)
        phases[-1]["sync"] = 0
        head = ops.camera_head_desc(phases, 4, Hd, C, scale_d, 1e-5, self.cam_sync, fail_host=self.cam_fail, spin_limit=model._cam_spin_limit)
        self.cam_one_launch = model._cam_one_launch and ops.camera_head_supported(head)
        if self.cam_one_launch:
            nw = sum(ph["N"] * ph["K"] for ph in phases if ph["kind"] == 0)
            P.camera_head(head, keep=[*clsn, ct, ch, cqkv, cao, t, raw, self.cam_sync], flops=2.0 * Mc * nw, nbytes=4.0 * nw)
        else:
            cn = z(Mc, C, dtype=f32); cq = z(Mc, C, dtype=f32); ckv = z(Mc, 2 * C, dtype=f32)
            for j in range(4):
                P.linear_f32(x=clsn[j], W=w[f"dec.camadapter.{j}.w"], bias=w[f"dec.camadapter.{j}.b"], out=ct.data_ptr() + j * C * 4,
                             M=B, N=C, K=D, ldx=D, ldw=D, ldc=4 * C, tag="cam.adapter")

            def ln32(src, dst, rows):
                P.layernorm(x=src, y=dst, rows=rows, D=C, ldx=C, ldy=C, eps=1e-5, rows_per_img=rows, in_rows_per_img=rows,
                            out_rows_per_img=rows, out_f32=1)

            def lin32(xb, pre, out, n, k, ldx, ldc, act=UD_ACT_NONE, accumulate=0, bias=True, **kw):
                P.linear_f32(x=xb, W=w[pre + ".w"], out=out, M=Mc, N=n, K=k, ldx=ldx, ldw=k, ldc=ldc, act=act, accumulate=accumulate,
                             tag="cam." + pre, **({"bias": w[pre + ".b"]} if bias else {}), **kw)

            def mlp32(pre, stream, out, accumulate, n_out=C):
                ln32(stream, cn, Mc)
                nh = w[pre + "fc1.w"].shape[0]
                lin32(cn, pre + "fc1", ch, nh, C, C, nh, act=UD_ACT_GELU)
                lin32(ch, pre + "fc2", out, n_out, nh, nh, n_out if n_out == 1 else C, accumulate=accumulate)

            mlp32("cam.project.", ct, t, 0)
            for blk in ("cam.agg1.", "cam.agg2."):
                ln32(t, cn, Mc)                                                   # norm_attnx and norm_attnctx share statistics
                lin32(cn, blk + "q", cq, C, C, C, C, add=w["cam.pos"], ldadd=C, add_mod=4)
                lin32(cn, blk + "kv", ckv, 2 * C, C, C, 2 * C)
                P.attention_small_f32(cq, ckv, cao, B, 4, Hd, C, scale_d)
                lin32(cao, blk + "out", t, C, C, C, C, accumulate=1, bias=False)
                mlp32(blk, t, t, 1)
            mlp32("cam.out.", t, raw, 0, n_out=1)

        def ln(src, dst, rows, dim=C):
            P.layernorm(x=src, y=dst, rows=rows, D=dim, ldx=dim, ldy=dim, eps=1e-5, rows_per_img=rows, in_rows_per_img=rows,
                        out_rows_per_img=rows)

        def mlp(pre, stream, rows, nrm, hidbuf, out=None, accumulate=1, out2=None, act2=UD_ACT_NONE, n_out=C, ldc=C):
            ln(stream, nrm, rows)
            nh = w[pre + "fc1.w"].shape[0]
            P.gemm(A=nrm, W=w[pre + "fc1.w"], bias=w[pre + "fc1.b"], out=hidbuf, M=rows, N=nh, K=C, lda=C, ldw=C, ldc=nh,
                   epi=UD_EPI_F16, act=UD_ACT_GELU)
            kw = dict(out2=out2, ldc2=C, act2=act2) if out2 is not None else {}
            P.gemm(A=hidbuf, W=w[pre + "fc2.w"], bias=w[pre + "fc2.b"], out=stream if out is None else out, M=rows, N=n_out, K=nh,
                   lda=nh, ldw=nh, ldc=ldc, epi=UD_EPI_F32, accumulate=accumulate, **kw)

        self.intr4 = z(B, 4, dtype=f32); self.K33 = z(B, 9, dtype=f32); kinv = z(B, 9, dtype=f32); self.Kpost = z(B, 9, dtype=f32)
        P.camera_intrinsics(raw, 1, self.intr4, self.K33, kinv, self.Kpost, B, Hn, Wn, float(self.rf), pl, pt)
        tap("intrinsics4", lambda: self.intr4.clone())
        # ---------------- rays (decoder.py:361-403 / GT camera: unidepthv2.py:299-303,361-362)
        self.rays = z(nb, 3, Hn, Wn, dtype=f32)
        if cam_nb and gt_mode == GT_GIVEN_RAYS:                             # pixel_decoder(inputs={"rays": ...}): the caller fills self.rays
            pass
        elif cam_nb and isinstance(gt_mode, tuple):                         # BatchCamera of mixed / iterative models: image i has its own model
            # (utils/camera.py:1166-1171: BatchCamera.unproject concatenates every member's own unproject)
            self.kinv_gt = z(nb, 16, dtype=f32)
            self.cam_scratch = z(4 * Hn * Wn + 16, dtype=f32)
            for i, gm in enumerate(gt_mode):
                if gm >= GT_OPENCV:
                    P.rays_camera(self.kinv_gt[i:i + 1], self.rays[i:i + 1], self.cam_scratch, Hn, Wn, gm)
                else:
                    P.rays(self.kinv_gt[i:i + 1], self.rays[i:i + 1], 1, Hn, Wn, gm)
        elif cam_nb and gt_mode >= GT_OPENCV:                               # iterative models: OPENCV, Fisheye624, MEI (one camera)
            self.kinv_gt = z(1, 16, dtype=f32)
            self.cam_scratch = z(4 * Hn * Wn + 16, dtype=f32)
            P.rays_camera(self.kinv_gt, self.rays, self.cam_scratch, Hn, Wn, gt_mode)
        elif cam_nb:
            self.kinv_gt = z(nb, 9, dtype=f32)
            P.rays(self.kinv_gt, self.rays, nb, Hn, Wn, gt_mode or 1)      # 1 pinhole K^-1, 2 EUCM / 3 Spherical parameters
        else:
            P.rays(kinv, self.rays, nb, Hn, Wn, 0)
        # ---------------- ray embedding + 4 camera-prompt cross-attention blocks (decoder.py:234-260)
        nbands = C // 2
        scales = (2.0 ** torch.linspace(0.0, math.log2(max(h, wg) // 2), steps=nbands)).to(dev)
        emb = z(nb * hwp, C)
        P.ray_embed(rays=self.rays, scales=scales, xhat=emb, nb=nb, Hn=Hn, Wn=Wn, h=h, w=wg, C=C, ldy=C, rows_per_img=hwp, eps=1e-5)
        tap("rays_embedding_normed", lambda: emb.view(nb, hwp, C)[:, :hw].float())   # the embedding after LayerNorm statistics (eps 1e-5)
        # the 4 blocks (one per encoder level) are independent: every step is ONE grouped launch (blockIdx.z = level)
        HC = Hd * 64
        Mk = nb * hwp
        fn = z(4, Md, C); qd = z(4, Md, HC); kd = z(4, Mk, HC); vtd = z(4, nb, Hd, 64, hwkp); aod = z(4, Md, HC); hidd = z(4, Md, 4 * C)
        c16_all = z(4, Md, C)
        c16 = [c16_all[j] for j in range(4)]
        G4 = dict(groups=4)
        ln(feat_all, fn, 4 * Md)
        P.gemm(A=fn, W=w["dhg.q.w"], bias=w["dhg.q.b"], out=qd, M=Md, N=HC, K=C, lda=C, ldw=C, ldc=HC, epi=UD_EPI_F16,
               gA=Md * C, gW=HC * C, gBias=HC, gOut=Md * HC, tag="dh.q(x4)", **G4)
        P.gemm(A=emb, W=w["dhg.kv.w"], bias=w["dhg.kv.b"], out=kd, out2=vtd, M=Mk, N=2 * HC, K=C, lda=C, ldw=C, ldc=HC, epi=UD_EPI_QKV,
               vsplit=HC, tok_per_img=hwp, kv_ld=hwkp, heads_v=Hd, gA=0, gW=2 * HC * C, gBias=2 * HC, gOut=Mk * HC,
               gOut2=nb * Hd * 64 * hwkp, tag="dh.kv(x4)", **G4)
        bc = int(nb == 1 and B > 1)
        P.attention(Q=qd, K=kd, Vt=vtd, O=aod, B=4 * B, H=Hd, Nq=hw, Nk=hw, ldq=HC, ldk=HC, ldo=HC, kv_ld=hwkp, q_rows_per_img=hwp,
                    k_rows_per_img=hwp, scale=scale_d, kv_broadcast=bc, kv_group=B, q_prescaled=1, tag="dh.attn(x4)")
        P.gemm(A=aod, W=w["dhg.out.w"], out=feat_all, M=Md, N=C, K=HC, lda=HC, ldw=HC, ldc=C, epi=UD_EPI_F32, accumulate=1,
               gA=Md * HC, gW=C * HC, gOut=Md * C, tag="dh.out(x4)", **G4)
        ln(feat_all, fn, 4 * Md)
        P.gemm(A=fn, W=w["dhg.fc1.w"], bias=w["dhg.fc1.b"], out=hidd, M=Md, N=4 * C, K=C, lda=C, ldw=C, ldc=4 * C, epi=UD_EPI_F16,
               act=UD_ACT_GELU, gA=Md * C, gW=4 * C * C, gBias=4 * C, gOut=Md * 4 * C, tag="dh.fc1(x4)", **G4)
        P.gemm(A=hidd, W=w["dhg.fc2.w"], bias=w["dhg.fc2.b"], out=feat_all, out2=c16_all, M=Md, N=C, K=4 * C, lda=4 * C, ldw=4 * C,
               ldc=C, ldc2=C, epi=UD_EPI_F32, accumulate=1, gA=Md * 4 * C, gW=C * 4 * C, gBias=C, gOut=Md * C, gOut2=Md * C,
               tag="dh.fc2(x4)", **G4)
        for j in range(4):
            tap(f"prompt_camera.{j}", lambda j=j: feat_all[j].view(B, hwp, C)[:, :hw].clone())
        # ---------------- latents + 3 x (ConvT inject, 2 RCU, 1x1 + x2 up) (decoder.py:262-282; upsample.py:137-223)
        lat = z(Md, C, dtype=f32)
        P.gemm(A=c16[0], W=w["dh.to_latents.w"], bias=w["dh.to_latents.b"], out=lat, M=Md, N=C, K=C, lda=C, ldw=C, ldc=C, epi=UD_EPI_F32)
        self.depth_features = z(B, C, h, wg, dtype=f32)
        P.nhwc_to_nchw(lat, self.depth_features, B, hw, C, C, hwp)
        gh, gw, rows_img = h, wg, hwp
        xh = None
        # The x2 up-sampling behind a stage's 1x1 conv feeds only the next stage's ConvTranspose accumulate: the ConvTranspose
        # accumulate interpolates the 1x1 conv's output itself (UdGemm.up_src) -- the up-sampled fp32 map is never written and read back
        # (45 + 90 MB at bs = 8, two launches).
        upfuse = True
        pending_up = {}
        for i in range(3):
            cur, outd = meta["chans"][i]
            k = max(1, 2 * i)
            Ms = B * rows_img
            l16 = z(Ms, cur); t16 = z(Ms, cur)
            P.gemm(A=c16[i + 1], W=w[f"dh.convt.{i}.w"], bias=w[f"dh.convt.{i}.b"], out=lat, out2=l16, M=Md, N=k * k * cur, K=C, lda=C,
                   ldw=C, ldc=cur, ldc2=cur, epi=UD_EPI_D2S, act2=UD_ACT_LRELU, d2s_k=k, d2s_Co=cur, d2s_Hin=h, d2s_Win=wg,
                   d2s_rows_in_img=hwp, d2s_out_img_pix=rows_img, **pending_up)
            pending_up = {}
            kp = w[f"dh.ups.{i}.0.conv1.w"].shape[1]
            conv = dict(zeros=zeros, M=Ms, N=cur, K=kp, ldw=kp, amode=UD_A_CONV3_ZERO, Himg=gh, Wimg=gw, Cin=cur, cstride=cur, coff=0,
                        rows_img=rows_img, img_stride=rows_img * cur)
            for c in range(2):
                P.gemm(A=l16, W=w[f"dh.ups.{i}.{c}.conv1.w"], bias=w[f"dh.ups.{i}.{c}.conv1.b"], out=t16, ldc=cur, epi=UD_EPI_F16,
                       act=UD_ACT_LRELU, **conv)
                P.gemm(A=t16, W=w[f"dh.ups.{i}.{c}.conv2.w"], bias=w[f"dh.ups.{i}.{c}.conv2.b"], out=lat, out2=l16, ldc=cur, ldc2=cur,
                       epi=UD_EPI_F32, accumulate=1 if c == 0 else 2, act2=UD_ACT_LRELU if c == 0 else UD_ACT_NONE, **conv)
            u = z(Ms, outd, dtype=f32)
            P.gemm(A=l16, W=w[f"dh.ups.{i}.up.w"], bias=w[f"dh.ups.{i}.up.b"], out=u, M=Ms, N=outd, K=_rup(cur, 64), lda=cur,
                   ldw=_rup(cur, 64), ldc=outd, epi=UD_EPI_F32)
            if i < 2:
                nlat = z(B * 4 * gh * gw, outd, dtype=f32)
                if upfuse:
                    pending_up = dict(up_src=u, up_H=gh, up_W=gw, up_ld=outd, up_img_rows=rows_img)      # consumed by the next stage's ConvT
                    # the tap (the stage's output = the up-sampled map BEFORE the next injection) is formed from u on demand
                    tap(f"ups.{i}", lambda u=u, gh=gh, gw=gw, outd=outd, ri=rows_img: torch.nn.functional.interpolate(
                        u.view(B, ri, outd)[:, :gh * gw].reshape(B, gh, gw, outd).permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=False))
                else:
                    P.upsample2x(in_=u, out=nlat, B=B, H=gh, W=gw, C=outd, ldin=outd, ldy=outd, mode=0, in_img_rows=rows_img)
                    tap(f"ups.{i}", lambda t=nlat, gh=gh, gw=gw, outd=outd: t.view(B, 2 * gh, 2 * gw, outd).permute(0, 3, 1, 2).clone())
                lat = nlat
            else:
                ldx = _rup(outd, 64)
                xh = z(B * 4 * gh * gw, ldx)
                P.upsample2x(in_=u, out=xh, B=B, H=gh, W=gw, C=outd, ldin=outd, ldy=ldx, mode=1, eps=1e-5, in_img_rows=rows_img)
                tap(f"ups.{i}_normed", lambda t=xh, gh=gh, gw=gw, outd=outd: t.view(B, 2 * gh, 2 * gw, -1)[..., :outd].permute(0, 3, 1, 2).float())
            gh, gw = 2 * gh, 2 * gw
            rows_img = gh * gw
        # ---------------- heads (decoder.py:284-318): LN+Linear (both branches), 3x3 reflect, AC resize, 3x3 reflect + 1x1 + exp
        nd, od = meta["nd"], meta["od"]
        Mh = B * gh * gw
        ldx = _rup(nd, 64)
        o2 = od // 2
        lr = z(2, Mh, o2)
        kp = w["dh.lr.w"].shape[2]
        # both branches read the same normalised map; their Linear layers live inside the composed conv filters (weights.py)
        P.gemm(A=xh, W=w["dh.lr.w"], bias=w["dh.lr.b"], out=lr, zeros=zeros, M=Mh, N=o2, K=kp, ldw=kp, ldc=o2, amode=UD_A_CONV3_REFLECT,
               epi=UD_EPI_F16, Himg=gh, Wimg=gw, Cin=nd, cstride=ldx, coff=0, rows_img=gh * gw, img_stride=gh * gw * ldx,
               groups=2, gA=0, gW=o2 * kp, gBias=o2, gOut=Mh * o2, tag="dh.lr(mlp folded)", flops=2.0 * 2 * Mh * o2 * 9 * nd)
        self.net = z(2, B, Hn, Wn, dtype=f32)              # [0] radius, [1] confidence at network resolution
        kp = w["dh.hr.w"].shape[2]
        b2 = w["dh.hr.b2"]
        head = dict(W=w["dh.hr.w"], bias=w["dh.hr.b1"], w2=w["dh.hr.w2"], out=self.net, zeros=zeros, M=B * Hn * Wn, N=32, K=kp, ldw=kp, epi=UD_EPI_HEAD,
                    Himg=Hn, Wimg=Wn, Cin=o2, cstride=o2, coff=0, rows_img=Hn * Wn, b2=b2[0], post_add=2.0, b2_g1=b2[1], post_add_g1=0.0, groups=2,
                    gW=32 * kp, gBias=32, gOut=B * Hn * Wn, gW2=32, flops=2.0 * 2 * B * Hn * Wn * 32 * 9 * o2)
        if o2 % 64 == 0:
            # ViT-L: the align_corners=True up-sampling to network resolution (decoder.py:299-301,309-311) happens inside the head conv's halo
            # loader -- the 518 x 518 x 64-channel maps of both branches (549 MB at bs = 8) are never written
            P.gemm(A=lr, amode=UD_A_CONV3_REFLECT_UP, Hsrc=gh, Wsrc=gw, img_stride=gh * gw * o2, gA=Mh * o2, tag="dh.hr(up fused)", **head)
        else:                                              # narrower heads (ViT-S / ViT-B: 32 / 48 channels): materialised up-sampling + implicit GEMM
            hr = z(2, B * Hn * Wn, o2)
            P.resize_ac(in_=lr, out=hr, G=2, B=B, Hin=gh, Win=gw, Hout=Hn, Wout=Wn, C=o2)
            P.gemm(A=hr, amode=UD_A_CONV3_REFLECT, img_stride=Hn * Wn * o2, gA=B * Hn * Wn * o2, **head)
        tap("logdepth", lambda: (torch.log(self.net[0]) - 2.0).view(B, 1, Hn, Wn))     # pre-exp head output (valid while |log| < 8: no clip)
        tap("logconf", lambda: torch.log(self.net[1]).view(B, 1, Hn, Wn))
        self.nb = nb
        self.Ho, self.Wo = self.Hp - pt - pb, self.Wp - pl - pr
        self.graph = None

    def finalize(self, out: dict, mode: int = 0):
        import ctypes as C
        pl, _, pt, _ = self.paddings
        d = ops.mk(ops.UdFinalize, radius_net=self.net[0], conf_net=self.net[1], rays_net=self.rays, confidence=out["confidence"],
                   radius=out["radius"], depth=out["depth"], points=out["points"], rays=out["rays"], B=self.B, nb_rays=self.nb,
                   Hn=self.Hn, Wn=self.Wn, Hp=self.Hp, Wp=self.Wp, pad_l=pl, pad_t=pt, Ho=self.Ho, Wo=self.Wo, mode=mode)
        ops.check(ops.lib.ud_finalize_outputs(C.byref(d), ops.cur_stream()), "ud_finalize_outputs")


class UniDepthV2(EngineModule):
    """Drop-in for the reference class (unidepthv2.py:111-117: nn.Module + PyTorchModelHubMixin): from_pretrained / to / eval / infer,
    attributes `resolution_level`, `interpolation_mode`, `shape_constraints`, `device`; the nn.Module surface is unidepth_amd/module.py."""

    def __init__(self, config: dict, eps: float = 1e-6, **kwargs):
        super().__init__()
        self.config = config
        self._arch = arch_of(config)
        self.shape_constraints = config["data"]["augmentations"]["shape_constraints"]   # unidepthv2.py:459
        self.interpolation_mode = "bilinear"                                            # unidepthv2.py:460
        self._sd = None
        self._w = None
        self._plans: "collections.OrderedDict" = collections.OrderedDict()
        self.max_plans = int(os.environ.get("UNIDEPTH_MAX_PLANS", "6"))   # LRU bound on cached (batch, shape, camera, slot) plans
        self._pos_cache: dict = {}
        self._cam_one_launch = True        # False after a reported grid-barrier time-out of the one-launch camera head (_check_camera_head)
        self._cam_spin_limit = 0           # 0 = the kernel's default (seconds); tests force the time-out with 1
        # True: a plan's launch program is replayed as ONE hipGraph launch (recorded on the second call of a signature).  For the launch-bound
        # small-batch calls (bs = 1: ~280 kernels of 2-10 us each); at bs = 8 the stream is never idle and eager replay is as fast.

    # ---- checkpoint I/O (HF mixin layout: config.json + model.safetensors / pytorch_model.bin) ----
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, **kwargs):
        path = str(pretrained_model_name_or_path)
        if not os.path.isdir(path):
            from huggingface_hub import snapshot_download     # e.g. "lpiccinelli/unidepth-v2-vitl14" (needs network / cache)
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
        model.load_state_dict(sd)
        return model

    def load_state_dict(self, state_dict: dict, strict: bool = False):
        if "model" in state_dict and not torch.is_tensor(state_dict["model"]):
            state_dict = state_dict["model"]                                            # unidepthv2.py:386-388
        self._sd = {k.replace("module.", ""): v.detach().float().cpu() for k, v in state_dict.items()}
        self._w = None
        self._plans.clear()
        self._pos_cache.clear()          # the resampled position embeddings are a function of the weights
        return self

    def clear_plans(self):
        """Drop every cached launch plan (device activation buffers of all (batch, shape, camera, slot) signatures seen so far)."""
        if self._plans and self._device.type == "cuda":
            torch.cuda.synchronize(self._device)          # programs of these plans may still be queued on pipeline streams
        self._plans.clear()

    def reserve_plans(self, n: int):
        """Make room for `n` distinct plan signatures visited cyclically (dist.infer_mixed knows its set up front: micro-batch sizes x
        shapes x camera modes x pipeline slots); an LRU smaller than the cycle would rebuild a ~2.6 GB plan on every call."""
        self.max_plans = max(self.max_plans, int(n))

    def trim_plans(self, n: int):
        """Set the LRU bound back to `n` and evict least-recently-used plans down to it (dist.infer_mixed after its call: the outputs it
        returned are fresh tensors, no plan buffer is referenced by them)."""
        self.max_plans = max(1, int(n))
        if len(self._plans) > self.max_plans:
            if self._device.type == "cuda":
                torch.cuda.synchronize(self._device)      # evicted plans' programs may still be queued on pipeline streams
            while len(self._plans) > self.max_plans:
                self._plans.popitem(last=False)

    def load_pretrained(self, model_file):
        return self.load_state_dict(torch.load(model_file, map_location="cpu", weights_only=False))

    def _move(self, device):          # EngineModule.to(): packed weights, plans and resampled position embeddings belong to a device
        self._w = None
        self._plans.clear()
        self._pos_cache.clear()

    def _ensure_packed(self):
        if self._device.type != "cuda":
            raise RuntimeError("UniDepthV2 (MI355X engine) runs on a ROCm GPU only: call .to('cuda') first; there is no CPU path")
        if self._sd is None:
            raise RuntimeError("no weights loaded (use from_pretrained or load_state_dict)")
        if self._w is None:
            with torch.cuda.device(self._device):
                self._w = pack(self.config, self._sd, self._device)

    def _pos_embed(self, h, w):
        """Absolute position embedding for an (h, w) patch grid; bicubic resample of the 37x37 table in fp32 when the grid
        differs (dinov2.py:267-304; computed once per grid on the host, it is a constant of the weights)."""
        key = (h, w)
        if key not in self._pos_cache:
            pe = self._w["host.pos_embed"]
            n = pe.shape[1] - 1
            m = int(math.sqrt(n))
            if h * w == n and h == w:
                out = pe[0]
            else:
                grid = pe[:, 1:].reshape(1, m, m, -1).permute(0, 3, 1, 2)
                grid = torch.nn.functional.interpolate(grid, size=(h, w), mode="bicubic", antialias=False)
                out = torch.cat([pe[0, :1], grid.permute(0, 2, 3, 1).reshape(h * w, -1)], 0)
            self._pos_cache[key] = out.contiguous()
        return self._pos_cache[key]

    def _pixels_bounds(self):
        lo, hi = self.shape_constraints["pixels_min"], self.shape_constraints["pixels_max"]
        if hasattr(self, "resolution_level"):
            assert 0 <= self.resolution_level < 10, "resolution_level should be in [0, 10)"
            step = (hi - lo) / 10
            return (self.resolution_level * step + lo, (self.resolution_level + 1) * step + lo)
        warnings.warn("!! self.resolution_level not set, using default bounds !!")
        return (lo, hi)

    def _plan(self, B, H, W, cam_nb, is_u8, normalize, slot=0, gt_mode=0, net=False) -> _Plan:
        bounds = (0.0, 0.0) if net else self._pixels_bounds()
        key = (B, H, W, cam_nb, is_u8, normalize, bounds, slot, gt_mode, net)
        plan = self._plans.get(key)
        if plan is None:
            # a plan owns the full activation set of its signature (~2.6 GB for ViT-L at bs=8): the cache is an LRU of `max_plans`
            # entries, so a stream of many image shapes (KITTI / nuScenes style) cannot grow device memory without bound
            if len(self._plans) >= max(1, self.max_plans):
                # the evicted plan's buffers go back to the caching allocator of whatever stream allocated them, while its launch
                # program may still be queued on ANOTHER stream (pipeline slots): drain the device first -- rare, and a plan rebuild
                # costs far more than this sync
                if self._device.type == "cuda":
                    torch.cuda.synchronize(self._device)
                while len(self._plans) >= max(1, self.max_plans):
                    self._plans.popitem(last=False)
            with torch.cuda.device(self._device):
                plan = _Plan(self, B, H, W, cam_nb, is_u8, normalize, bounds, gt_mode, net)
            self._plans[key] = plan
        else:
            self._plans.move_to_end(key)
        return plan

    # ---- the hot path ----
    @torch.no_grad()
    def infer(self, rgb: torch.Tensor, camera=None, normalize: bool = True, *, slot: int = 0):
        """Same contract as the reference infer() (unidepthv2.py:239-339).  `slot` (engine extension, keyword only) selects an
        independent set of activation buffers: calls with different slots may be in flight at the same time on different HIP
        streams (unidepth_amd/pipeline.py); calls with the same slot must be stream-ordered, as with the reference module."""
        return self._infer(rgb, camera, normalize, slot, None)

    @torch.no_grad()
    def infer_with_taps(self, rgb: torch.Tensor, camera=None, normalize: bool = True, names=None):
        """infer() that also returns intermediate tensors (SURVEY.md 8c tap list) in the reference's layouts: the launch program is
        replayed in segments and the tapped buffers are copied out between segments (buffers are reused along the network).
        Returns (outputs, {tap name: tensor}).  Parity tooling -- the product path is infer()."""
        taps: dict = {}
        out = self._infer(rgb, camera, normalize, 0, (taps, None if names is None else set(names)))
        return out, taps

    def _infer(self, rgb, camera, normalize, slot, taps):
        # F.interpolate(..., align_corners=False) in the reference's _postprocess (unidepthv2.py:80-89) accepts exactly these two for 4-D input
        if self.interpolation_mode not in ("bilinear", "bicubic"):
            raise ValueError(f"interpolation_mode {self.interpolation_mode!r}: 'bilinear' or 'bicubic' (align_corners=False) expected")
        self._ensure_packed()
        if rgb.ndim == 3:
            rgb = rgb.unsqueeze(0)
        B, _, H, W = rgb.shape
        Kc = None
        cam_obj = None                                             # non-pinhole models: parameters go to the ray kernels as they are
        if camera is not None:
            if isinstance(camera, torch.Tensor):
                Kc = camera
            else:
                cam_obj = as_camera(camera)                        # reference-class objects are matched by class name
                if isinstance(cam_obj, BatchCamera) and cam_obj.uniform() is not None:
                    cam_obj = cam_obj.uniform()                    # one closed-form model for every image: the batched ray kernel
                if cam_obj.gt_mode == GT_PINHOLE:
                    Kc, cam_obj = cam_obj.K, None
            if Kc is not None:
                assert Kc.shape[-1] == 3 and Kc.shape[-2] == 3, "camera tensor should be of shape (..., 3, 3): assume pinhole"
                Kc = Kc.detach().reshape(-1, 3, 3).float().cpu()
        is_u8 = rgb.dtype == torch.uint8
        with torch.cuda.device(self._device):
            cam_nb = 0 if camera is None else (Kc.shape[0] if Kc is not None else cam_obj.params.shape[0])
            # one camera broadcasts over the batch, otherwise one per image (the reference fails with a shape error here too)
            assert cam_nb in (0, 1, B), f"camera batch {cam_nb} does not match the image batch {B} (one camera, or one per image)"
            gt_mode = 0 if camera is None else (GT_PINHOLE if Kc is not None else cam_obj.gt_mode)
            mixed = isinstance(cam_obj, BatchCamera)               # one camera per image, models differ (or are iterative): one ray launch per image
            unsort = None
            if mixed:
                assert cam_nb == B, f"BatchCamera of {cam_nb} cameras for a batch of {B} images (one per image)"
                # canonical order: images sorted by camera model (stable), so a plan (2.6 GB at bs = 8, keyed on the tuple of per-image models)
                # is shared by every ORDERING of the same models instead of being rebuilt per ordering; infer() is bit-exactly
                # batch-permutation equivariant, the outputs are put back in the caller's order below
                modes = cam_obj.gt_modes
                order = sorted(range(B), key=lambda i: modes[i])
                if order != list(range(B)):
                    idx = torch.tensor(order, device=rgb.device)
                    rgb = rgb.index_select(0, idx)
                    cam_obj = BatchCamera([cam_obj.cameras[i] for i in order])
                    unsort = torch.empty(B, dtype=torch.long)
                    unsort[torch.tensor(order)] = torch.arange(B)
                gt_mode = cam_obj.gt_modes
            self._check_camera_head()
            plan = self._plan(B, H, W, cam_nb, is_u8, bool(normalize), int(slot), gt_mode)
            plan.rgb.copy_(rgb if is_u8 else rgb.float(), non_blocking=True)
            if mixed:
                buf = torch.zeros(B, 16)
                pl, _, pt, _ = plan.paddings
                for i, c in enumerate(cam_obj.cameras):
                    if c.gt_mode == GT_PINHOLE:                    # camera.crop(-pad) then .resize(rf), K^-1 in the first nine slots
                        Kn = c.K.clone()
                        Kn[:, 0, 2] += pl
                        Kn[:, 1, 2] += pt
                        Kn[:, :2, :] *= plan.rf
                        buf[i, :9] = torch.inverse(Kn).reshape(9)
                    else:
                        pn = c.network_params(plan.paddings, plan.rf)
                        buf[i, : pn.shape[1]] = pn[0]
                plan.kinv_gt.copy_(buf)
            elif cam_obj is not None:
                pn = cam_obj.network_params(plan.paddings, plan.rf)               # [n, <= 16] -> the parameter slots of the ray kernel
                buf = torch.zeros(pn.shape[0], plan.kinv_gt.shape[1])
                buf[:, :pn.shape[1]] = pn
                plan.kinv_gt.copy_(buf)
            if Kc is not None:
                pl, _, pt, _ = plan.paddings
                Kn = Kc.clone()                                    # camera.crop(-pad) then .resize(rf): utils/camera.py:78-81,115-120
                Kn[:, 0, 2] += pl
                Kn[:, 1, 2] += pt
                Kn[:, :2, :] *= plan.rf
                plan.kinv_gt.copy_(torch.inverse(Kn).reshape(-1, 9))
            self._run(plan, 0, len(plan.prog), taps)
            out = self._collect(plan, B)
            if unsort is not None:
                ui = unsort.to(self._device)
                out = {k: (v.index_select(0, ui) if v.shape[0] == B else v) for k, v in out.items()}
            return out

    def _check_camera_head(self):
        """The one-launch camera head (csrc/camera_f32.hip) needs its grid co-resident; a barrier that times out (CU mask, partitioned device,
        foreign kernels holding CUs) sets a host-mapped word and turns that call's intrinsics / rays / depth into NaN.  Polled here, at the
        start of the next call, without a device synchronisation: the failure is reported once, the model switches to the per-layer launches
        (same arithmetic, no grid barrier) for every later call."""
        for plan in self._plans.values():
            if plan.cam_fail is not None and int(plan.cam_fail[0]):
                torch.cuda.synchronize(self._device)
                self._cam_one_launch = False
                self.clear_plans()
                raise RuntimeError("unidepth_amd: a grid barrier of the one-launch camera head timed out in an earlier infer() call (its workgroups "
                                   "were not co-resident on the device); that call returned NaN intrinsics, rays and depth.  This model now runs the "
                                   "camera head as per-layer launches; repeat the failed call.")

    @staticmethod
    def _run(plan: _Plan, first: int, last: int, taps=None):
        """Replay ops [first, last) of the plan; with `taps` = (dict, names or None) the replay stops at every tap point in range."""
        if taps is None:
            plan.prog.run(first, last)
            return
        store, names = taps
        pos = first
        for name, at, fn in sorted(plan.tap_points, key=lambda t: t[1]):
            if at < first or at > last or (names is not None and name not in names):
                continue
            if at > pos:
                plan.prog.run(pos, at)
                pos = at
            store[name] = fn()
        if last > pos:
            plan.prog.run(pos, last)

    def _collect(self, plan: _Plan, B: int):
        dev, f32 = self._device, torch.float32
        out = {
            "confidence": torch.empty(B, 1, plan.Ho, plan.Wo, dtype=f32, device=dev),
            "radius": torch.empty(B, 1, plan.Ho, plan.Wo, dtype=f32, device=dev),
            "depth": torch.empty(B, 1, plan.Ho, plan.Wo, dtype=f32, device=dev),
            "points": torch.empty(B, 3, plan.Ho, plan.Wo, dtype=f32, device=dev),
            "rays": torch.empty(plan.nb, 3, plan.Ho, plan.Wo, dtype=f32, device=dev),
        }
        plan.finalize(out, 1 if self.interpolation_mode == "bicubic" else 0)
        out["intrinsics"] = plan.Kpost.view(B, 3, 3).clone()
        out["depth_features"] = plan.depth_features.clone()
        return {k: out[k] for k in ("confidence", "intrinsics", "radius", "depth", "points", "rays", "depth_features")}

    __call__ = infer

    # ---- module seams: the two halves of encode_decode() (unidepthv2.py:341-379) with the reference's signatures,
    # each a sub-range of the launch program of a network-resolution plan, so either half can be swapped for the reference's /
    # the oracle's during parity bisection.
    @property
    def embed_dim(self):
        return self._arch["D"]

    @property
    def embed_dims(self):
        return [self._arch["D"]] * self._arch["depth"]

    @property
    def depths(self):
        return list(self._arch["output_idx"])

    patch_size = 14

    @torch.no_grad()
    def pixel_encoder(self, image: torch.Tensor, *, slot: int = 0):
        """image [B,3,Hn,Wn] (normalised, network resolution, multiples of 14) -> (outputs, class_tokens) like the reference's
        DINOv2 wrapper (backbones/dinov2.py:324-347): lists indexed by block, entries [B,h,w,D] / [B,1,D] after the final LayerNorm.
        Only the blocks the decoder consumes (stacking_fn 'last' over the `depths` ranges: blocks output_idx - 1) are filled; the
        other entries are None -- the reference computes and discards them."""
        self._ensure_packed()
        B, _, Hn, Wn = image.shape
        with torch.cuda.device(self._device):
            plan = self._plan(B, Hn, Wn, 0, False, False, int(slot), 0, net=True)
            plan.rgb.copy_(image.float(), non_blocking=True)
            plan.prog.run(0, plan.enc_last)
            D, hw = self._arch["D"], plan.h * plan.w
            hwp = _rup(hw, 8)
            outs, cls = [None] * self._arch["depth"], [None] * self._arch["depth"]
            # the engine's final LayerNorm stores statistics only (its affine is folded into the adapters' weights): apply it here
            gn, bn = self._w["enc.norm.g"], self._w["enc.norm.b"]
            for j, li in enumerate(self._arch["output_idx"]):
                outs[li - 1] = (plan.featn[j].view(B, hwp, D)[:, :hw].float() * gn + bn).reshape(B, plan.h, plan.w, D)
                cls[li - 1] = (plan.clsn[j][:B] * gn + bn).view(B, 1, D)
        return outs, cls

    @torch.no_grad()
    def encode(self, image: torch.Tensor):
        """The 4 feature maps / class tokens encode_decode() hands to the decoder (unidepthv2.py:365-372)."""
        outs, cls = self.pixel_encoder(image)
        idx = [i - 1 for i in self._arch["output_idx"]]
        return [outs[i] for i in idx], [cls[i] for i in idx]

    @torch.no_grad()
    def pixel_decoder(self, inputs: dict, image_metas=None, *, slot: int = 0):
        """Decoder.forward (unidepthv2/decoder.py:405-462): inputs = {image [B,3,H,W] (shape only), features: 4 x [B,h,w,D],
        tokens: 4 x [B,1,D], optional rays [B|1,3,H,W] (GT rays replace the predicted ones, decoder.py:400)} ->
        {radius [B,1,H,W], depth_features [B,C,h,w], confidence [B,1,H,W], intrinsics [B,3,3], rays [B|1,H*W,3]}."""
        self._ensure_packed()
        B, _, H, W = inputs["image"].shape
        feats, toks, rays = inputs["features"], inputs["tokens"], inputs.get("rays", None)
        assert len(feats) == 4 and len(toks) == 4
        with torch.cuda.device(self._device):
            nb = 0 if rays is None else int(rays.shape[0])
            assert nb in (0, 1, B)
            plan = self._plan(B, H, W, nb, False, False, int(slot), GT_GIVEN_RAYS if nb else 0, net=True)
            D, hw = self._arch["D"], plan.h * plan.w
            hwp = _rup(hw, 8)
            # the decoder program consumes LayerNorm STATISTICS (the affine of the encoder's final norm is folded into the adapters):
            # undo the affine of the reference-semantics inputs.  Seam / bisection path only -- infer() never does this.
            gn, bn = self._w["enc.norm.g"], self._w["enc.norm.b"]
            if float(gn.abs().min()) < 1e-6:
                raise NotImplementedError("pixel_decoder seam: the encoder's final LayerNorm has a zero scale, its affine cannot be undone")
            for j in range(4):
                assert tuple(feats[j].shape) == (B, plan.h, plan.w, D), (tuple(feats[j].shape), (B, plan.h, plan.w, D))
                plan.featn[j].view(B, hwp, D)[:, :hw].copy_((feats[j].reshape(B, hw, D).to(self._device, torch.float32) - bn) / gn)
                plan.clsn[j][:B].copy_((toks[j].reshape(B, D).to(self._device, torch.float32) - bn) / gn)
            if nb:
                plan.rays.copy_(rays.reshape(nb, 3, H, W))
            plan.prog.run(plan.dec_first, len(plan.prog))
            return {"radius": plan.net[0].view(B, 1, H, W).clone(), "depth_features": plan.depth_features.clone(),
                    "confidence": plan.net[1].view(B, 1, H, W).clone(), "intrinsics": plan.K33.view(B, 3, 3).clone(),
                    "rays": plan.rays.reshape(plan.nb, 3, H * W).permute(0, 2, 1).contiguous()}

    @torch.no_grad()
    def forward_export(self, rgbs: torch.Tensor, rays: Optional[torch.Tensor] = None):
        """Pure-tensor entries of the reference's ONNX wrappers (unidepthv2/export.py:27-45 `forward(rgbs)` and :57-76
        `forward(rgbs, rays)`): rgbs is the NETWORK image [B,3,H,W] (normalised, multiples of 14; no pre-/post-processing) ->
        (pts_3d [B,3,H,W], confidence [B,1,H,W], intrinsics [B,3,3] at network resolution)."""
        B, _, H, W = rgbs.shape
        features, tokens = self.encode(rgbs)
        inputs = {"image": rgbs, "features": features, "tokens": tokens}
        if rays is not None:
            inputs["rays"] = rays
        out = self.pixel_decoder(inputs, [])
        r = out["rays"].permute(0, 2, 1).reshape(-1, 3, H, W)
        return r * out["radius"], out["confidence"], out["intrinsics"]

    # ---- quick look at the last call's network-resolution tensors (kept from round 1; infer_with_taps() is the full tap list) ----
    @torch.no_grad()
    def debug_taps(self, plan: Optional[_Plan] = None):
        """Tensors of the last infer() in reference layout: final-normed features/cls tokens, network-res maps."""
        plan = plan or next(reversed(self._plans.values()))
        hw = plan.h * plan.w
        hwp = _rup(hw, 8)
        D = self._arch["D"]
        gn, bn = self._w["enc.norm.g"], self._w["enc.norm.b"]          # reference semantics: final LayerNorm WITH its affine
        feats = [(f.view(plan.B, hwp, D)[:, :hw].float() * gn + bn).view(plan.B, plan.h, plan.w, D) for f in plan.featn]
        cls = [(c[: plan.B].float() * gn + bn).view(plan.B, 1, D) for c in plan.clsn]
        return dict(features=feats, tokens=cls, radius_net=plan.net[0], confidence_net=plan.net[1], rays_net=plan.rays,
                    intrinsics_net=plan.K33.view(-1, 3, 3), intr4=plan.intr4)
