"""Thin host-side helpers over the C-ABI: build descriptors from torch tensors (device memory + streams are
PyTorch-ROCm's; all arithmetic is in libunidepth_hip.so) and record / replay launch programs."""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib
from ._lib import (UD_A_CONV3_REFLECT, UD_A_CONV3_REFLECT_UP, UD_A_CONV3_ZERO, UD_A_DENSE, UD_ACT_GELU, UD_ACT_LRELU, UD_ACT_NONE,  # noqa: F401
                   UD_EPI_D2S, UD_EPI_F16, UD_EPI_F32, UD_EPI_HEAD, UD_EPI_QKV, UdAttention, UdFinalize, UdGemm,
                   UdCameraHead, UdCamPhase, UdDwConv7, UdLayerNorm, UdLinearF32, UdPreprocess, UdRayEmbed, UdResizeAC, UdUpsample2x, check, lib)


def ptr(t):
    if t is None:
        return None
    if isinstance(t, int):
        return t
    assert t.is_cuda, "device tensor expected"
    return t.data_ptr()


def cur_stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def mk(struct, **kw):
    """Fill a descriptor struct; tensors become device pointers (an int is taken as a raw address)."""
    d = struct()
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            v = v.data_ptr()
        elif isinstance(v, (tuple, list)):
            v = (C.c_float * len(v))(*v)
        setattr(d, k, v)
    return d


def camera_head_desc(phases, T, H, Cc, scale, eps, sync_ws, workgroups=0, fail_host=None, spin_limit=0):
    """UdCameraHead from a list of phase dicts (UdCamPhase fields; tensors become device pointers, an int is a raw address)."""
    d = UdCameraHead()
    assert len(phases) <= len(d.ph), "too many phases for UdCameraHead"
    for i, ph in enumerate(phases):
        for k, v in ph.items():
            setattr(d.ph[i], k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    d.n_phases, d.T, d.H, d.C, d.scale, d.eps, d.workgroups = len(phases), T, H, Cc, scale, eps, workgroups
    d.sync_ws = sync_ws.data_ptr() if isinstance(sync_ws, torch.Tensor) else sync_ws
    d.fail_host = fail_host.data_ptr() if isinstance(fail_host, torch.Tensor) else fail_host
    d.spin_limit = spin_limit
    return d


def camera_head_supported(desc) -> bool:
    """Inside the limits of the one-launch camera head (include/unidepth_hip.h UdCameraHead)?  Host-side check, nothing is launched."""
    return lib.ud_camera_head_supported(C.byref(desc)) == 0


def camera_head(desc):
    check(lib.ud_camera_head_f32(C.byref(desc), cur_stream()), "ud_camera_head_f32")


def v1_desc(kind, a=None, b=None, c=None, out=None, out2=None, i=(), f=()):
    d = _lib.UdV1Op()
    d.kind = kind
    d.a, d.b, d.c, d.out, d.out2 = ptr(a), ptr(b), ptr(c), ptr(out), ptr(out2)
    for k, v in enumerate(i):
        d.i[k] = int(v)
    for k, v in enumerate(f):
        d.f[k] = float(v)
    return d


def v1_op(kind, **kw):
    check(lib.ud_v1_op(C.byref(v1_desc(kind, **kw)), cur_stream()), "ud_v1_op")


# ---- eager single-op entry points (used by the kernel-level tests) -------------------------------------
def gemm(**kw):
    check(lib.ud_gemm_f16(C.byref(mk(UdGemm, **kw)), cur_stream()), "ud_gemm_f16")


def layernorm(**kw):
    check(lib.ud_layernorm_f32_f16(C.byref(mk(UdLayerNorm, **kw)), cur_stream()), "ud_layernorm_f32_f16")


def row_stats_finalize(partials, stats, M, slabs, D, eps):
    check(lib.ud_row_stats_finalize(ptr(partials), ptr(stats), M, slabs, D, eps, cur_stream()), "ud_row_stats_finalize")


def attention(**kw):
    check(lib.ud_attention_f16(C.byref(mk(UdAttention, **kw)), cur_stream()), "ud_attention_f16")


class Program:
    """Recorded list of kernel launches (UdProgram): built once per (batch, shape), replayed per infer()."""

    def __init__(self):
        self.h = lib.ud_program_create()
        if not self.h:
            raise MemoryError("ud_program_create")
        self.keep = []          # keep tensors referenced by raw pointers alive
        self.meta = []          # per op: (kernel class, tag, algorithmic flops, algorithmic bytes) for bench / profiling
        self._splitk = None     # scratch of the two-way K split (UdGemm.splitk_ws / splitk_cnt), one per program = per stream

    def __del__(self):
        try:
            if self.h:
                lib.ud_program_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def __len__(self):
        return lib.ud_program_size(self.h)

    def _k(self, kw, cls="misc", flops=0.0, nbytes=0.0):
        self.keep.extend(v for v in kw.values() if isinstance(v, torch.Tensor))
        self.meta.append((cls, kw.pop("tag", cls), float(kw.pop("flops", flops)), float(nbytes)))

    def gemm(self, **kw):
        g = max(1, kw.get("groups", 0))
        n = kw["N"]
        # algorithmic reduction length: a split-fp16 product (a_wrap / w_wrap: K concatenation, include/unidepth_hip.h) multiplies
        # every (m, n, k) of the layer two or three times -- that is the price of the precision, not additional algorithmic work
        k_alg = kw["K"]
        if kw.get("a_wrap") or kw.get("w_wrap"):      # (a conv amode with w_wrap is rejected by the C side: leave k_alg alone and let it say so)
            if not kw.get("amode", 0):
                k_alg = kw.get("a_wrap") or kw.get("w_wrap")
            elif kw.get("a_wrap") and kw.get("Cin"):
                k_alg = kw["K"] * kw["a_wrap"] // kw["Cin"]
        tag, flops = kw.pop("tag", None), kw.pop("flops", 2.0 * kw["M"] * n * k_alg * g)
        tiles = -(-kw["M"] // 128) * -(-n // 128)
        if (g == 1 and kw.get("amode", 0) == UD_A_DENSE and kw.get("epi", 0) in (UD_EPI_F16, UD_EPI_F32) and tiles <= 128
                and kw["K"] >= 1024 and kw["K"] % 128 == 0 and "splitk_ws" not in kw):
            # small problems (small batches): see UdGemm.splitk_ws in include/unidepth_hip.h; one workspace per program
            key = 0
            if not isinstance(self._splitk, dict):
                self._splitk = {}
            if key not in self._splitk:
                dev = kw["A"].device
                self._splitk[key] = (torch.empty(256 * 16384, dtype=torch.float32, device=dev), torch.zeros(128, dtype=torch.int32, device=dev))
                self.keep += list(self._splitk[key])
            kw["splitk_ws"], kw["splitk_cnt"] = self._splitk[key]
        tiles192 = -(-kw["M"] // 192) * -(-n // 256)
        if (g == 1 and kw.get("amode", 0) in (UD_A_DENSE, UD_A_CONV3_ZERO) and kw.get("epi", 0) in (UD_EPI_F16, UD_EPI_F32) and 32 <= tiles192 <= 128
                and kw["K"] >= 2048 and kw["K"] % 128 == 0 and "splitk_ws" not in kw and kw.get("row_stats_out") is None
                and kw.get("row_stats_in") is None and kw.get("max_out") is None and not kw.get("a_wrap") and not kw.get("w_wrap")):
            # two-way K split of the large-tile list (UdGemm.splitk_ws_bytes): one scratch per program, sized for its largest user
            if not isinstance(self._splitk, dict):
                self._splitk = {}
            need = 2 * tiles192 * 192 * 256 * 4
            if 1 not in self._splitk or self._splitk[1][0].numel() * 4 < need:
                dev = kw["A"].device
                self._splitk[1] = (torch.empty(need // 4, dtype=torch.float32, device=dev), torch.zeros(128, dtype=torch.int32, device=dev))
                self.keep += list(self._splitk[1])
            kw["splitk_ws"], kw["splitk_cnt"] = self._splitk[1]
            kw["splitk_ws_bytes"] = self._splitk[1][0].numel() * 4
        d = mk(UdGemm, **kw)
        pick, epi, amode = lib.ud_gemm_pick(C.byref(d)), kw.get("epi", 0), kw.get("amode", 0)
        lnc, grp = "true" if pick & 16 else "false", "true" if pick & 32 else "false"
        pick &= 15
        if pick <= 2:       # names as rocprofv3 prints them (template arguments), so profiles and bench lines can be joined
            cls = "gemm_kernel<Cfg<%s>, %d, %d, 2, false>" % (("128, 64, 64", "64, 64, 32", "32, 32, 32")[pick], epi, amode)
        elif pick in (6, 7):     # 128x128 tiles, 4-stage pipelined ring (7: + two-way K split)
            cls = "gemm_kernel<Cfg<128, 64, 64>, %d, %d, 4, %s>" % (epi, amode, "true" if pick == 7 else "false")
        elif pick <= 4:
            # 7th template argument: the 3-deep weight ring of the 192-row tile list (csrc/gemm.hip launch256: dense A, not grouped, K >= 128)
            w3 = pick == 3 and amode == UD_A_DENSE and grp == "false" and kw["K"] >= 128 and kw.get("tile_hint", 0) != 9
            cls = "gemm256_kernel<%d, %d, %d, false, %s, %s, %s, false>" % (pick, epi, amode, lnc, grp, "true" if w3 else "false")
        elif pick == 8:     # row-balanced schedule of the 256-column kernel
            cls = "gemm256_kernel<4, %d, %d, true, %s, false, false, false>" % (epi, amode, lnc)
        elif pick == 11:    # 192-row tiles, ping-pong schedule (csrc/gemm_pp.hip)
            cls = "gemm_pp_f32_kernel<3, 8>"
        elif pick == 10:    # 192-row tile list, two-way K split (2 * tiles workgroups)
            cls = "gemm256_kernel<3, %d, %d, false, false, false, false, true>" % (epi, amode)
        elif amode == 3 and kw.get("Cin", 0) == 64 and epi == UD_EPI_HEAD:
            cls = "conv_head_regw_kernel"                       # head conv with its weights in registers
        else:
            cls = "conv_tile_kernel<%d, %d, %s, %s>" % (n // 16, epi, "true" if amode >= 2 else "false", "true" if amode == 3 else "false")
        self.keep.extend(v for v in kw.values() if isinstance(v, torch.Tensor))
        # algorithmic HBM bytes (every operand element once): A (conv modes: the image, not the 9x gathered rows), W, outputs,
        # + the old fp32 values of an accumulating epilogue
        m, k = kw["M"], kw["K"]
        a_bytes = 2.0 * m * ((kw.get("a_wrap") or kw["Cin"]) if amode else (kw.get("a_wrap") or k)) * (1 if kw.get("gA", 1) else 1.0 / g)
        o_bytes = {UD_EPI_F16: 2.0, UD_EPI_QKV: 2.0, UD_EPI_F32: 4.0, UD_EPI_D2S: 4.0, UD_EPI_HEAD: 4.0 / max(n, 1)}[epi] * m * n
        if epi in (UD_EPI_F32, UD_EPI_D2S):
            o_bytes = (0.0 if kw.get("accumulate", 0) == 2 else 4.0 * m * n) + (4.0 * m * n if kw.get("accumulate", 0) or epi == UD_EPI_D2S else 0.0)
            o_bytes += 2.0 * m * n if kw.get("out2") is not None else 0.0
        nbytes = g * (a_bytes + 2.0 * n * k + o_bytes)
        self.meta.append((cls, tag or cls, float(flops), float(nbytes)))
        return check(lib.ud_program_add_gemm(self.h, C.byref(d)))

    def layernorm(self, **kw):
        self._k(kw, "layernorm", 0.0, 6.0 * kw["rows"] * kw["D"])
        return check(lib.ud_program_add_layernorm(self.h, C.byref(mk(UdLayerNorm, **kw))))

    def row_stats_finalize(self, partials, stats, M, slabs, D, eps, tag="row_stats_finalize"):
        self.keep += [partials, stats]
        self.meta.append(("row_stats_finalize", tag, 0.0, 8.0 * M * (slabs + 1)))
        return check(lib.ud_program_add_row_stats_finalize(self.h, ptr(partials), ptr(stats), M, slabs, D, eps))

    def attention(self, **kw):
        self._k(kw, "attention", 4.0 * kw["B"] * kw["H"] * kw["Nq"] * kw["Nk"] * 64)
        return check(lib.ud_program_add_attention(self.h, C.byref(mk(UdAttention, **kw))))

    def linear_f32(self, **kw):
        self._k(kw, "camera_f32", 2.0 * kw["M"] * kw["N"] * kw["K"], 4.0 * kw["N"] * kw["K"])
        return check(lib.ud_program_add_linear_f32(self.h, C.byref(mk(UdLinearF32, **kw))))

    def camera_head(self, desc, keep=(), flops=0.0, nbytes=0.0):
        self.keep.extend(keep)
        self.meta.append(("camera_f32", "cam.head", float(flops), float(nbytes)))
        return check(lib.ud_program_add_camera_head(self.h, C.byref(desc)))

    def attention_small_f32(self, q, kv, out, B, T, H, Cc, scale):
        self.keep += [q, kv, out]
        self.meta.append(("camera_f32", "attention_small", 0.0, 0.0))
        return check(lib.ud_program_add_attention_small_f32(self.h, ptr(q), ptr(kv), ptr(out), B, T, H, Cc, scale))

    def preprocess(self, **kw):
        self._k(kw, "preprocess"); return check(lib.ud_program_add_preprocess(self.h, C.byref(mk(UdPreprocess, **kw))))

    def fill_rows(self, dst, src, n_img, rows_per_img, row_off, D, ld):
        self.keep += [dst, src]
        self.meta.append(("misc", "fill_rows", 0.0, 0.0))
        return check(lib.ud_program_add_fill_rows(self.h, ptr(dst), ptr(src), n_img, rows_per_img, row_off, D, ld))

    def camera_intrinsics(self, raw, raw_stride, intr4, K33, Kinv33, Kpost33, B, Hn, Wn, rf, pad_l, pad_t):
        self.keep += [raw, intr4, K33, Kinv33, Kpost33]
        self.meta.append(("misc", "camera_intrinsics", 0.0, 0.0))
        return check(lib.ud_program_add_camera_intrinsics(self.h, ptr(raw), raw_stride, ptr(intr4), ptr(K33), ptr(Kinv33),
                                                          ptr(Kpost33), B, Hn, Wn, rf, pad_l, pad_t))

    def rays(self, Kinv33, rays, nb, Hn, Wn, gt_mode):
        self.keep += [Kinv33, rays]
        self.meta.append(("misc", "rays", 0.0, 12.0 * nb * Hn * Wn))
        return check(lib.ud_program_add_rays(self.h, ptr(Kinv33), ptr(rays), nb, Hn, Wn, gt_mode))

    def rays_camera(self, params, rays, scratch, Hn, Wn, model):
        self.keep += [params, rays, scratch]
        self.meta.append(("misc", "rays_camera", 0.0, 12.0 * Hn * Wn))
        return check(lib.ud_program_add_rays_camera(self.h, ptr(params), ptr(rays), ptr(scratch), Hn, Wn, model))

    def ray_embed(self, **kw):
        self._k(kw, "ray_embed"); return check(lib.ud_program_add_ray_embed(self.h, C.byref(mk(UdRayEmbed, **kw))))

    def upsample2x(self, **kw):
        self._k(kw, "upsample2x", 0.0, kw["B"] * kw["H"] * kw["W"] * kw["C"] * (4.0 + 4 * (4.0 if kw.get("mode", 0) == 0 else 2.0))); return check(lib.ud_program_add_upsample2x(self.h, C.byref(mk(UdUpsample2x, **kw))))

    def resize_ac(self, **kw):
        self._k(kw, "resize_ac", 0.0, 2.0 * kw["G"] * kw["B"] * kw["C"] * (kw["Hin"] * kw["Win"] + kw["Hout"] * kw["Wout"])); return check(lib.ud_program_add_resize_ac(self.h, C.byref(mk(UdResizeAC, **kw))))

    def finalize(self, **kw):
        self._k(kw, "finalize"); return check(lib.ud_program_add_finalize(self.h, C.byref(mk(UdFinalize, **kw))))

    def nhwc_to_nchw(self, src, dst, B, hw, Cc, ld, rows_per_img):
        self.keep += [src, dst]
        self.meta.append(("misc", "nhwc_to_nchw", 0.0, 8.0 * B * hw * Cc))
        return check(lib.ud_program_add_nhwc_to_nchw(self.h, ptr(src), ptr(dst), B, hw, Cc, ld, rows_per_img))

    # ---- ConvNeXt-side ops (UniDepthV1)
    def dwconv7(self, **kw):
        self._k(kw, "dwconv7", 98.0 * kw["B"] * kw["H"] * kw["W"] * kw["C"], 8.0 * kw["B"] * kw["H"] * kw["W"] * kw["C"])
        return check(lib.ud_program_add_dwconv7(self.h, C.byref(mk(UdDwConv7, **kw))))

    def layernorm_patchify2(self, x, out, B, H, W, Cc, ldo, eps):
        self.keep += [x, out]
        self.meta.append(("ln_patchify2", "ln_patchify2", 0.0, 6.0 * B * H * W * Cc))
        return check(lib.ud_program_add_layernorm_patchify2(self.h, ptr(x), ptr(out), B, H, W, Cc, ldo, eps))

    def patchify4(self, img, out, B, H, W, ldo):
        self.keep += [img, out]
        self.meta.append(("misc", "patchify4", 0.0, 0.0))
        return check(lib.ud_program_add_patchify4(self.h, ptr(img), ptr(out), B, H, W, ldo))

    def max_(self, dst, src, n, init):
        self.keep += [dst, src]
        self.meta.append(("max", "max_stack", 0.0, (8.0 if init else 12.0) * n))
        return check(lib.ud_program_add_max(self.h, ptr(dst), ptr(src), n, int(init)))

    def spatial_mean(self, x, out, B, HW, Cc, ldo):
        self.keep += [x, out]
        self.meta.append(("misc", "spatial_mean", 0.0, 4.0 * B * HW * Cc))
        return check(lib.ud_program_add_spatial_mean(self.h, ptr(x), ptr(out), B, HW, Cc, ldo))

    def v1(self, kind, a=None, b=None, c=None, out=None, out2=None, i=(), f=(), tag="v1"):
        """One decoder-side op of the UniDepthV1 path (include/unidepth_hip.h UdV1Op)."""
        self.keep += [t for t in (a, b, c, out, out2) if isinstance(t, torch.Tensor)]
        self.meta.append(("v1." + tag, tag, 0.0, 0.0))
        return check(lib.ud_program_add_v1_op(self.h, C.byref(v1_desc(kind, a, b, c, out, out2, i, f))))

    def run(self, first=0, last=None, stream=None):
        """Replay ops [first, last) on the current (or given) stream."""
        last = len(self) if last is None else last
        check(lib.ud_program_run(self.h, first, last, cur_stream() if stream is None else stream), "ud_program_run")
