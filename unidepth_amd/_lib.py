"""ctypes binding of the C-ABI declared in include/unidepth_hip.h (libunidepth_hip.so, built in-tree by
unidepth_amd/csrc/build.sh).  There is NO fallback: if the library is missing the import fails loudly."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UNIDEPTH_HIP_LIB", os.path.join(_HERE, "libunidepth_hip.so"))   # env override: A/B two builds in one process tree

UD_EPI_F16, UD_EPI_F32, UD_EPI_QKV, UD_EPI_D2S, UD_EPI_HEAD = 0, 1, 2, 3, 4
UD_ACT_NONE, UD_ACT_GELU, UD_ACT_LRELU = 0, 1, 2
UD_A_DENSE, UD_A_CONV3_ZERO, UD_A_CONV3_REFLECT, UD_A_CONV3_REFLECT_UP = 0, 1, 2, 3

vp, fp, i32, i64, f32 = C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_float


class UdGemm(C.Structure):
    _fields_ = [
        ("A", vp), ("W", vp), ("bias", fp), ("out", vp), ("out2", vp), ("add", fp), ("zeros", vp), ("w2", fp),
        ("M", i32), ("N", i32), ("K", i32),
        ("lda", i32), ("ldw", i32), ("ldc", i32), ("ldc2", i32), ("ldadd", i32),
        ("amode", i32), ("epi", i32), ("act", i32), ("act2", i32), ("accumulate", i32),
        ("rows_in", i32), ("rows_out", i32), ("row_off", i32), ("add_row_off", i32),
        ("Himg", i32), ("Wimg", i32), ("Cin", i32), ("cstride", i32), ("coff", i32), ("rows_img", i32),
        ("img_stride", i64),
        ("vsplit", i32), ("tok_per_img", i32), ("kv_ld", i32), ("heads_v", i32),
        ("d2s_k", i32), ("d2s_Co", i32), ("d2s_Hin", i32), ("d2s_Win", i32), ("d2s_rows_in_img", i32),
        ("d2s_out_img_pix", i64),
        ("b2", f32), ("post_add", f32),
        ("groups", i32),
        ("gA", i64), ("gW", i64), ("gBias", i64), ("gOut", i64), ("gOut2", i64), ("gW2", i64),
        ("b2_g1", f32), ("post_add_g1", f32), ("tile_hint", i32),
        ("splitk_ws", vp), ("splitk_cnt", vp), ("Hsrc", i32), ("Wsrc", i32), ("a_wrap", i32), ("w_wrap", i32), ("max_out", fp), ("max_init", i32), ("grp_rows", i32),
        ("row_stats_out", fp), ("row_stats_final", fp), ("row_stats_ticket", vp), ("row_stats_in", fp), ("wsum", fp), ("ln_slabs", i32), ("ln_D", i32), ("ln_eps", f32),
        ("up_src", fp), ("up_H", i32), ("up_W", i32), ("up_ld", i32), ("up_img_rows", i32),
        ("splitk_ws_bytes", i64),
    ]


class UdLayerNorm(C.Structure):
    _fields_ = [("x", fp), ("y", vp), ("rows", i32), ("D", i32), ("ldx", i32), ("ldy", i32), ("eps", f32),
                ("rows_per_img", i32), ("in_rows_per_img", i32), ("in_row_off", i32), ("out_rows_per_img", i32),
                ("out_row_off", i32), ("out_f32", i32), ("gamma", fp), ("beta", fp), ("add", fp), ("cls_y", fp), ("ldcls", i32)]


class UdLinearF32(C.Structure):
    _fields_ = [("x", fp), ("W", fp), ("bias", fp), ("add", fp), ("out", fp), ("M", i32), ("N", i32), ("K", i32), ("ldx", i32),
                ("ldw", i32), ("ldc", i32), ("ldadd", i32), ("add_mod", i32), ("act", i32), ("accumulate", i32)]


class UdCamPhase(C.Structure):
    _fields_ = [("x", fp), ("W", fp), ("bias", fp), ("add", fp), ("out", fp), ("M", i32), ("N", i32), ("K", i32), ("ldx", i32), ("ldc", i32),
                ("ldadd", i32), ("add_mod", i32), ("add_cols", i32), ("kind", i32), ("ln", i32), ("act", i32), ("accumulate", i32), ("sync", i32)]


UD_CAM_MAX_PHASES = 24


class UdCameraHead(C.Structure):
    _fields_ = [("ph", UdCamPhase * UD_CAM_MAX_PHASES), ("n_phases", i32), ("T", i32), ("H", i32), ("C", i32), ("scale", f32), ("eps", f32),
                ("sync_ws", vp), ("workgroups", i32), ("fail_host", vp), ("spin_limit", C.c_uint)]


class UdDwConv7(C.Structure):
    _fields_ = [("x", fp), ("w", fp), ("bias", fp), ("y", fp), ("B", i32), ("H", i32), ("W", i32), ("C", i32), ("ldx", i32), ("ldy", i32),
                ("y16", vp), ("stats_out", fp), ("ldy16", i32), ("stats_final", fp), ("stats_ticket", vp), ("ln_eps", f32)]


class UdV1Op(C.Structure):
    _fields_ = [("kind", i32), ("a", vp), ("b", vp), ("c", vp), ("out", vp), ("out2", vp), ("i", i32 * 12), ("f", f32 * 4)]


class UdKnn(C.Structure):
    _fields_ = [("p1", vp), ("p2", vp), ("lengths1", vp), ("lengths2", vp), ("dists", vp), ("idx", vp), ("work", vp),
                ("N", i32), ("P1", i32), ("P2", i32), ("D", i32), ("K", i32), ("norm", i32)]


class UdExtractPatches(C.Structure):
    _fields_ = [("in_", vp), ("out", vp), ("centers", vp), ("B", i32), ("C", i32), ("H", i32), ("W", i32), ("N", i32),
                ("h", i32), ("w", i32), ("pad_h", i32), ("pad_w", i32)]


(UD_V1_RESIZE_AA, UD_V1_SH_EMBED, UD_V1_SOFTMAX, UD_V1_ATTN_FEWQ, UD_V1_HEAD_MIX) = range(1, 6)
(UD_V1_ADD, UD_V1_COPY_ROWS) = (8, 9)
(UD_V1_CAMERA, UD_V1_POINTS, UD_V1_MEAN3, UD_V1_PREPROCESS, UD_V1_VIT_TAP) = range(11, 16)
UD_V1_RESIZE_AC_SPLIT = 17
UD_V1_OUT_CONV3 = 18
UD_ACT_CLAMPEXP = 3


class UdAttention(C.Structure):
    _fields_ = [("Q", vp), ("K", vp), ("Vt", vp), ("O", vp), ("B", i32), ("H", i32), ("Nq", i32), ("Nk", i32),
                ("ldq", i32), ("ldk", i32), ("ldo", i32), ("kv_ld", i32), ("q_rows_per_img", i32),
                ("k_rows_per_img", i32), ("scale", f32), ("kv_broadcast", i32), ("kv_group", i32), ("q_prescaled", i32)]


class UdPreprocess(C.Structure):
    _fields_ = [("rgb", vp), ("patches", vp), ("B", i32), ("H", i32), ("W", i32), ("pad_l", i32), ("pad_t", i32),
                ("Hp", i32), ("Wp", i32), ("Hn", i32), ("Wn", i32), ("ldp", i32), ("is_u8", i32), ("normalize", i32),
                ("mean", f32 * 3), ("inv_std", f32 * 3)]


class UdRayEmbed(C.Structure):
    _fields_ = [("rays", fp), ("scales", fp), ("xhat", vp), ("nb", i32), ("Hn", i32), ("Wn", i32), ("h", i32),
                ("w", i32), ("C", i32), ("ldy", i32), ("rows_per_img", i32), ("eps", f32)]


class UdUpsample2x(C.Structure):
    _fields_ = [("in_", vp), ("out", vp), ("B", i32), ("H", i32), ("W", i32), ("C", i32), ("ldin", i32), ("ldy", i32),
                ("mode", i32), ("eps", f32), ("in_img_rows", i32)]


class UdResizeAC(C.Structure):
    _fields_ = [("in_", vp), ("out", vp), ("G", i32), ("B", i32), ("Hin", i32), ("Win", i32), ("Hout", i32),
                ("Wout", i32), ("C", i32)]


class UdFinalize(C.Structure):
    _fields_ = [("radius_net", fp), ("conf_net", fp), ("rays_net", fp), ("confidence", fp), ("radius", fp),
                ("depth", fp), ("points", fp), ("rays", fp), ("B", i32), ("nb_rays", i32), ("Hn", i32), ("Wn", i32),
                ("Hp", i32), ("Wp", i32), ("pad_l", i32), ("pad_t", i32), ("Ho", i32), ("Wo", i32), ("mode", i32)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP kernel library is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` or unidepth_amd/csrc/build.sh (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    P = C.POINTER
    sig = {
        "ud_gemm_f16": [P(UdGemm), vp],
        "ud_gemm_pick": [P(UdGemm)],
        "ud_layernorm_f32_f16": [P(UdLayerNorm), vp],
        "ud_rccl_unique_id": [vp],
        "ud_rccl_init": [vp, i32, i32],
        "ud_rccl_allgather_outputs": [vp, vp, C.c_size_t, i32, vp],
        "ud_rccl_finalize": [],
        "ud_attention_f16": [P(UdAttention), vp],
        "ud_row_stats_finalize": [vp, vp, i32, i32, i32, f32, vp],
        "ud_program_add_row_stats_finalize": [vp, vp, vp, i32, i32, i32, f32],
        "ud_linear_f32": [P(UdLinearF32), vp],
        "ud_attention_small_f32": [vp, vp, vp, i32, i32, i32, i32, f32, vp],
        "ud_program_add_linear_f32": [vp, P(UdLinearF32)],
        "ud_camera_head_f32": [P(UdCameraHead), vp],
        "ud_camera_head_supported": [P(UdCameraHead)],
        "ud_program_add_camera_head": [vp, P(UdCameraHead)],
        "ud_program_add_attention_small_f32": [vp, vp, vp, vp, i32, i32, i32, i32, f32],
        "ud_preprocess_patches": [P(UdPreprocess), vp],
        "ud_fill_rows_f32": [vp, vp, i32, i32, i32, i32, i32, vp],
        "ud_camera_intrinsics": [vp, i32, vp, vp, vp, vp, i32, i32, i32, f32, i32, i32, vp],
        "ud_rays_from_kinv": [vp, vp, i32, i32, i32, i32, vp],
        "ud_rays_from_camera": [vp, vp, vp, i32, i32, i32, vp],
        "ud_ray_embed": [P(UdRayEmbed), vp],
        "ud_upsample2x_nhwc": [P(UdUpsample2x), vp],
        "ud_resize_ac_nhwc_f16": [P(UdResizeAC), vp],
        "ud_finalize_outputs": [P(UdFinalize), vp],
        "ud_nhwc_to_nchw_f32": [vp, vp, i32, i32, i32, i32, i32, vp],
        "ud_program_destroy": [vp],
        "ud_program_size": [vp],
        "ud_program_add_gemm": [vp, P(UdGemm)],
        "ud_program_add_layernorm": [vp, P(UdLayerNorm)],
        "ud_program_add_attention": [vp, P(UdAttention)],
        "ud_program_add_preprocess": [vp, P(UdPreprocess)],
        "ud_program_add_fill_rows": [vp, vp, vp, i32, i32, i32, i32, i32],
        "ud_program_add_camera_intrinsics": [vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, f32, i32, i32],
        "ud_program_add_rays": [vp, vp, vp, i32, i32, i32, i32],
        "ud_program_add_rays_camera": [vp, vp, vp, vp, i32, i32, i32],
        "ud_program_add_ray_embed": [vp, P(UdRayEmbed)],
        "ud_program_add_upsample2x": [vp, P(UdUpsample2x)],
        "ud_program_add_resize_ac": [vp, P(UdResizeAC)],
        "ud_program_add_finalize": [vp, P(UdFinalize)],
        "ud_program_add_nhwc_to_nchw": [vp, vp, vp, i32, i32, i32, i32, i32],
        "ud_dwconv7_nhwc_f32": [P(UdDwConv7), vp],
        "ud_layernorm_patchify2": [vp, vp, i32, i32, i32, i32, i32, f32, vp],
        "ud_patchify4_nchw": [vp, vp, i32, i32, i32, i32, vp],
        "ud_max_f32": [vp, vp, i64, i32, vp],
        "ud_spatial_mean_f32": [vp, vp, i32, i32, i32, i32, vp],
        "ud_program_add_dwconv7": [vp, P(UdDwConv7)],
        "ud_program_add_layernorm_patchify2": [vp, vp, vp, i32, i32, i32, i32, i32, f32],
        "ud_program_add_patchify4": [vp, vp, vp, i32, i32, i32, i32],
        "ud_program_add_max": [vp, vp, vp, i64, i32],
        "ud_program_add_spatial_mean": [vp, vp, vp, i32, i32, i32, i32],
        "ud_v1_op": [P(UdV1Op), vp],
        "ud_program_add_v1_op": [vp, P(UdV1Op)],
        "ud_knn_points": [P(UdKnn), vp],
        "ud_knn_split": [P(UdKnn)],
        "ud_extract_patches": [P(UdExtractPatches), vp],
        "ud_program_run": [vp, i32, i32, vp],
        "ud_calib_mfma_stream": [vp, i32, i32, vp, C.POINTER(C.c_double), vp],
        "ud_calib_mfma_stream16": [vp, i32, i32, vp, C.POINTER(C.c_double), vp],
        "ud_version": [],
        "ud_struct_size": [i32],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = None if name == "ud_program_destroy" else i32
    lib.ud_program_create.argtypes = []
    lib.ud_program_create.restype = vp
    lib.ud_last_error.argtypes = []
    lib.ud_last_error.restype = C.c_char_p
    for i, st in enumerate([UdGemm, UdLayerNorm, UdAttention, UdPreprocess, UdRayEmbed, UdUpsample2x, UdResizeAC, UdFinalize, UdLinearF32, UdDwConv7, UdV1Op, UdKnn, UdExtractPatches, UdCameraHead]):
        # a library whose descriptors differ from this mirror in ANY way is a hard error (A/B runs rebuild both arms from one tree:
        # an older .so would read the appended fields -- a_wrap, row_stats_* -- as garbage or not at all)
        if lib.ud_struct_size(i) != C.sizeof(st):
            raise ImportError(f"ctypes mirror of {st.__name__} is out of sync with include/unidepth_hip.h "
                              f"({C.sizeof(st)} vs {lib.ud_struct_size(i)} bytes)")
    return lib


lib = _load()


def check(rc: int, what: str = ""):
    if rc < 0:
        raise RuntimeError(f"unidepth_hip {what} failed (code {rc}): {lib.ud_last_error().decode()}")
    return rc
