// fp16 MFMA GEMM family for gfx950:  C[M,N] = A[M,K] * W[N,K]^T  (+ fused epilogues, + implicit-GEMM 3x3 conv A-gather).
//
// Design (MI355X-first, see DESIGN.md section "GEMM"):
//  * 128 x BN x 64 tiles, 4 waves (wave64), v_mfma_f32_16x16x32_f16, fp32 accumulators in registers.
//  * both operands are K-contiguous and go HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR round trip);
//    the LDS image is lane-linear [row][64 halves]; bank conflicts of the ds_read_b128 fragment reads are removed
//    by an XOR swizzle of the 16-byte chunk index, applied on the *global source* address and on the read address:
//    chunk' = chunk ^ ((row >> 1) & 7)  -> the 16 lanes of every ds_read_b128 service group hit 16 distinct slots.
//  * 2-stage LDS ring: tile kt+1 streams in while tile kt is multiplied; one s_barrier per K-tile (two workgroups per CU
//    cover each other's stalls).  With at most one workgroup per CU (tile count <= CU count, small batches) the 128 x 128
//    configuration switches to a 4-stage ring, software-pipelined over K-tiles (gemm_body, NST = 4).
//  * operands are issued as mfma(W_frag, A_frag): the accumulator then holds C^T fragments, i.e. each lane owns
//    4 consecutive n for one m -> 8-byte fp16 / 16-byte fp32 row-major stores.  Tiles that produce V^T for the
//    attention kernel use the opposite order (4 consecutive tokens per lane -> 8-byte stores into [head][d][token]).
//  * workgroup -> tile map is XCD-aware: the 8 XCDs get contiguous ranges of the tile list (private L2 reuse).
#include "ud_common.h"

#ifdef UD_TRACE
// timeline instrumentation (tools/trace_gemm.py; never in the product build): per workgroup and tile, 100 MHz wall-clock stamps
__device__ unsigned long long* ud_trace_ptr = nullptr;
#define UD_STAMP(slot)                                                                           \
  do {                                                                                           \
    if (ud_trace_ptr && threadIdx.x == 0 && trace_tile < 8)                                      \
      ud_trace_ptr[((size_t)blockIdx.x * 8 + trace_tile) * 8 + (slot)] = wall_clock64();         \
  } while (0)
#else
#define UD_STAMP(slot)
#endif

namespace {

constexpr int UD_LNC_TILES = 2;                  // statistics tables of the folded-LayerNorm consumer: the current tile's and the next one's
constexpr int UD_LNC_PLANE = UD_LNC_TILES * 256; // floats per plane of its LDS table: [2][256] rstd, then [2][256] -mean * rstd

template <int BN_, int WM_, int WN_>
struct Cfg {
  static constexpr int BM = 128, BN = BN_, BK = 64, WM = WM_, WN = WN_;
  static constexpr int NWM = BM / WM, NWN = BN / WN;
  static constexpr int TM = WM / 16, TN = WN / 16;
  static constexpr int A_BYTES = BM * BK * 2;
  static constexpr int B_BYTES = BN * BK * 2;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int A_INSTR = BM / 32;  // glds per wave per tile (8 rows each)
  static constexpr int B_INSTR = BN / 32;
  static_assert(NWM * NWN == 4, "4 waves");
};

template <int I> struct IntTag { static constexpr int value = I; };

struct ConvLane {
  long long base;  // element offset of the image
  int y, x;
  int valid;
};

// ---------------- epilogue (shared by all tile configurations) ----------------
// acc[i][j] is the 16x16 MFMA accumulator of m-tile i / n-tile j of this wave's sub-tile whose origin is (mbase, nbase).
// SWAP (default): lane owns m = mbase + 16i + (lane & 15), n = nbase + 16j + 4*(lane >> 4) + r.
// !SWAP (V^T tiles): lane owns m = mbase + 16i + 4*(lane >> 4) + r, n = nbase + 16j + (lane & 15).
// residual-as-accumulator-init: for `out += A W^T` epilogues the old fp32 values are loaded into the accumulators BEFORE the
// K loop (overlapping the first operand DMA) instead of being re-read in the epilogue, where every tile of a one-round GEMM
// would hit HBM at the same moment; the epilogue then only writes.
// ---- LayerNorm folded into a producer / consumer pair of GEMMs (UdGemm.row_stats_out / row_stats_in, include/unidepth_hip.h).
// Every formula below is written with explicit fma / add order and shared by the straight-line and the edge-tile epilogues: an
// element's bits must not depend on which path its tile took (batch-permutation equivariance of infer(), bit-identical schedules).
__device__ __forceinline__ void ud_row_stats_acc(const f32x4 v, float& s1, float& s2) {     // one lane's 4 columns of a 16-column tile
  s1 += (v[0] + v[1]) + (v[2] + v[3]);
  s2 = __builtin_fmaf(v[0], v[0], s2);
  s2 = __builtin_fmaf(v[1], v[1], s2);
  s2 = __builtin_fmaf(v[2], v[2], s2);
  s2 = __builtin_fmaf(v[3], v[3], s2);
}
__device__ __forceinline__ void ud_row_stats_store(const UdGemm& p, float s1, float s2, int m, int nbase, int lane, bool ok) {
  // the 4 lanes (lane & 15, lane >> 4 = 0..3) of a row hold its 64-column slab in 16-column pieces: (s[0] + s[1]) + (s[2] + s[3])
  s1 += __shfl_xor(s1, 16, 64);
  s2 += __shfl_xor(s2, 16, 64);
  s1 += __shfl_xor(s1, 32, 64);
  s2 += __shfl_xor(s2, 32, 64);
  if (ok && nbase < p.N && (lane >> 4) == 0) {
    f32x2 o;
    o[0] = s1; o[1] = s2;
    // system-scope write-through: the workgroup that finishes the row tile LAST (another XCD, in general) reduces these right away
    // (UdGemm.row_stats_final); same fence-free exchange as the two-way K split (sc0 sc1 stores, vmcnt(0), one ticket atomic)
    float* dst = p.row_stats_out + ((size_t)m * (p.N >> 6) + (nbase >> 6)) * 2;
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_nop 2" ::"v"(dst), "v"(o) : "memory");
  }
}
// consumer: out = rstd * (x W^T - mean wsum) + bias as two fmas in this order: fma(acc, rstd, fma(nmr, wsum, bias)), nmr = -mean * rstd.
// (Starting the accumulators at -mean * wsum instead -- one fma in the epilogue -- makes all of them live from the top of the tile,
// where the zero-initialised ones come to life one MFMA phase at a time: 70-350 spilled registers in the Q|K / V^T instantiations.)
__device__ __forceinline__ float ud_ln_apply(float acc, float rs, float nmr, float ws, float bias) {
  return __builtin_fmaf(acc, rs, __builtin_fmaf(nmr, ws, bias));
}
__device__ __forceinline__ f32x4 ud_ln_apply4(f32x4 acc, float rs, float nmr, f32x4 ws, f32x4 bias) {
  f32x4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = ud_ln_apply(acc[r], rs, nmr, ws[r], bias[r]);
  return o;
}

// UdGemm.up_src (UD_EPI_D2S): the old value of output pixel (img, Y, X), channels o .. o + 3, as the bilinear x2 up-sampling of the source map
// (align_corners=False; the same expression as upsample2x_kernel<0> in pointwise.hip, evaluated where the transposed convolution adds to it)
__device__ __forceinline__ f32x4 ud_up2_fetch(const UdGemm& p, int img, int Y, int X, int o) {
  float fy = 0.5f * ((float)Y + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
  float fx = 0.5f * ((float)X + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < p.up_H - 1), x1 = x0 + (x0 < p.up_W - 1);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float* in = p.up_src + (size_t)img * p.up_img_rows * p.up_ld + o;
  const f32x4 v00 = *(const f32x4*)(in + ((size_t)y0 * p.up_W + x0) * p.up_ld);
  const f32x4 v01 = *(const f32x4*)(in + ((size_t)y0 * p.up_W + x1) * p.up_ld);
  const f32x4 v10 = *(const f32x4*)(in + ((size_t)y1 * p.up_W + x0) * p.up_ld);
  const f32x4 v11 = *(const f32x4*)(in + ((size_t)y1 * p.up_W + x1) * p.up_ld);
  return (1.0f - ly) * ((1.0f - lx) * v00 + lx * v01) + ly * ((1.0f - lx) * v10 + lx * v11);
}

template <int TM, int TN, int TMA>      // TM row tiles used of an accumulator array of TMA (deduced)
__device__ __forceinline__ void gemm_preload_acc(const UdGemm& p, f32x4 (&acc)[TMA][TN], int mbase, int nbase, int lane, const char* out) {
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = mbase + i * 16 + (lane & 15);
    int orow = m;
    if (p.rows_in > 0) {
      const int img = m / p.rows_in;
      orow = img * p.rows_out + (m - img * p.rows_in) + p.row_off;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nb = nbase + j * 16 + 4 * (lane >> 4);
      if (m < p.M && nb < p.N) acc[i][j] = *(const f32x4*)((const float*)out + (size_t)orow * p.ldc + nb);
    }
  }
}

template <int ACT>
__device__ __forceinline__ float ud_act_t(float x) {
  if constexpr (ACT == UD_ACT_GELU) return ud_gelu_erf(x);
  else if constexpr (ACT == UD_ACT_LRELU) return ud_lrelu(x);
  else return x;
}

template <int ACT>
__device__ __forceinline__ f32x4 ud_act4_t(f32x4 v) {      // GELU on packed fp32 pairs (same bits as the scalar function)
  if constexpr (ACT == UD_ACT_GELU) {
    const f32x2 g0 = ud_gelu_erf2((f32x2){v[0], v[1]}), g1 = ud_gelu_erf2((f32x2){v[2], v[3]});
    return (f32x4){g0[0], g0[1], g1[0], g1[1]};
  } else {
    return (f32x4){ud_act_t<ACT>(v[0]), ud_act_t<ACT>(v[1]), ud_act_t<ACT>(v[2]), ud_act_t<ACT>(v[3])};
  }
}

// ACT = activation of the fp16 output (EPI_F16/QKV) or of the fp16 copy (EPI_F32/D2S), resolved ONCE per kernel by
// gemm_epilogue below: a per-element switch on the runtime value compiled to ~700 scalar branches in the unrolled epilogue.
// lnst: LayerNorm-folded consumer (UD_EPI_F16 / UD_EPI_QKV): rstd of the wave's rows, lnst[r] = row mbase + r (LDS)
template <int TM, int TN, int EPI, bool SWAP, bool PRELOADED, int ACT, int TMA>
__device__ __forceinline__ void gemm_epilogue_impl(const UdGemm& p, f32x4 (&acc)[TMA][TN], int mbase, int nbase, int lane, const float* bias,
                                                   char* out, char* out2, const float* w2, float b2, float post_add, char* stage,
                                                   const float* lnst = nullptr) {
  if constexpr (!SWAP) {
    // V^T tiles of UD_EPI_QKV: lane owns tokens mb..mb+3 (one image: tok_per_img % 4 == 0) for column n.
    static_assert(EPI == UD_EPI_QKV, "non-swapped orientation only for V^T");
    half_t* vt = (half_t*)out2;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mb = mbase + i * 16 + 4 * (lane >> 4);
      if (mb >= p.M) continue;
      const int img = mb / p.tok_per_img;
      const int t = mb - img * p.tok_per_img;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = nbase + j * 16 + (lane & 15);
        if (n >= p.N) continue;
        const float bv = bias ? bias[n] : 0.0f;
        const int nv = n - p.vsplit;
        const int hd = nv >> 6, d = nv & 63;
        half4 h;
        if (lnst) {
          const float wsn = p.wsum[n];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int lr = i * 16 + 4 * (lane >> 4) + r;
            h[r] = (half_t)ud_ln_apply(acc[i][j][r], lnst[lr], lnst[UD_LNC_PLANE + lr], wsn, bv);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) h[r] = (half_t)(acc[i][j][r] + bv);
        }
        // V^T key order inside every aligned 16-key group: 4-key blocks [0, 2, 1, 3] (include/unidepth_hip.h, UD_EPI_QKV)
        const int tp = (t & ~15) | ((t & 4) << 1) | ((t & 8) >> 1);
        *(half4*)(vt + (((size_t)img * p.heads_v + hd) * 64 + d) * p.kv_ld + tp) = h;
      }
    }
    return;
  } else {
    // ---- fp16 outputs: stage the wave's sub-tile through LDS so that global stores are 16 B per lane and every
    //      8 lanes write one full 128-byte row segment (direct MFMA-layout stores are 8 B per lane in 32-byte pieces)
    if constexpr ((EPI == UD_EPI_F16 || EPI == UD_EPI_QKV) && TN == 4 && (TM % 2) == 0) {
      if (stage != nullptr && (p.N & 7) == 0 && (p.ldc & 7) == 0 && !(p.ldc2 >> 30)) {
        constexpr int CH = (TM % 4) == 0 ? 4 : 2;      // m-tiles staged per pass (64 or 32 rows x 64 columns)
#pragma unroll
        for (int c = 0; c < TM / CH; ++c) {
#pragma unroll
          for (int il = 0; il < CH; ++il) {
            const int i = c * CH + il;
            const int m = mbase + i * 16 + (lane & 15);
            int arow = m;
            if (p.rows_in > 0) arow = m - (m / p.rows_in) * p.rows_in + p.add_row_off;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int nb = nbase + j * 16 + 4 * (lane >> 4);
              f32x4 v = acc[i][j];
              if (nb < p.N) {
                if (bias) v += *(const f32x4*)(bias + nb);
                if (p.add && m < p.M) v += *(const f32x4*)(p.add + (size_t)arow * p.ldadd + nb);
              }
              half4 h;
#pragma unroll
              for (int r = 0; r < 4; ++r) h[r] = (half_t)ud_act4_t<ACT>(v)[r];
              *(half4*)(stage + (il * 16 + (lane & 15)) * 144 + (j * 16 + 4 * (lane >> 4)) * 2) = h;
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int it = 0; it < 2 * CH; ++it) {
            const int id = it * 64 + lane;
            const int row = id >> 3, ch = id & 7;
            const int m = mbase + c * CH * 16 + row;
            const int n = nbase + ch * 8;
            if (m < p.M && n < p.N) {
              int orow = m;
              if (p.rows_in > 0) {
                const int img = m / p.rows_in;
                orow = img * p.rows_out + (m - img * p.rows_in) + p.row_off;
              }
              const half8 val = *(const half8*)(stage + row * 144 + ch * 16);
              *(half8*)((half_t*)out + (size_t)orow * p.ldc + n) = val;
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        return;
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = mbase + i * 16 + (lane & 15);
      const bool mok = m < p.M;
      // output row remap
      int orow = m, arow = m;
      if (p.rows_in > 0) {
        const int img = m / p.rows_in;
        const int pr = m - img * p.rows_in;
        orow = img * p.rows_out + pr + p.row_off;
        arow = pr + p.add_row_off;
      }
      if constexpr (EPI == UD_EPI_HEAD) {
        float part = 0.0f;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int nb = nbase + j * 16 + 4 * (lane >> 4);
          if (nb < p.N) {
            const f32x4 bv = *(const f32x4*)(bias + nb);
            const f32x4 wv2 = *(const f32x4*)(w2 + nb);
#pragma unroll
            for (int r = 0; r < 4; ++r) part += ud_lrelu(acc[i][j][r] + bv[r]) * wv2[r];
          }
        }
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        if (mok && (lane >> 4) == 0) {
          float y = part + b2;
          y = fminf(fmaxf(y, -8.0f), 8.0f);
          ((float*)out)[orow] = __expf(y + post_add);
        }
        continue;
      }
      float rs1 = 0.f, rs2 = 0.f;                   // row statistics of the stored fp32 values (UD_EPI_F32 producer of a folded LayerNorm)
      if constexpr (EPI == UD_EPI_D2S) {
        // m -> (img, y, x) on the input grid
        const int img = m / p.d2s_rows_in_img;
        const int pp = m - img * p.d2s_rows_in_img;
        const int y = pp / p.d2s_Win, x = pp - y * p.d2s_Win;
        const bool ok = mok && pp < p.d2s_Hin * p.d2s_Win;
        const int k = p.d2s_k;
        const int Wout = p.d2s_Win * k;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int nb = nbase + j * 16 + 4 * (lane >> 4);
          if (!ok || nb >= p.N) continue;
          const int ac = nb / p.d2s_Co;
          const int o = nb - ac * p.d2s_Co;
          const int a = ac / k, c = ac - a * k;
          const long long pix = (long long)img * p.d2s_out_img_pix + (long long)(y * k + a) * Wout + (x * k + c);
          float* dst = (float*)out + pix * p.ldc + o;
          f32x4 v = p.up_src ? ud_up2_fetch(p, img, y * k + a, x * k + c, o) : *(f32x4*)dst;
          const f32x4 bv = *(const f32x4*)(bias + o);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += acc[i][j][r] + bv[r];
          *(f32x4*)dst = v;
          if (out2) {
            half4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = (half_t)ud_act4_t<ACT>(v)[r];
            *(half4*)((half_t*)out2 + pix * p.ldc2 + o) = h;
          }
        }
        continue;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int nb = nbase + j * 16 + 4 * (lane >> 4);
        if (!mok || nb >= p.N) continue;
        f32x4 v = acc[i][j];
        if (lnst) {
          const int lr = i * 16 + (lane & 15);
          v = ud_ln_apply4(v, lnst[lr], lnst[UD_LNC_PLANE + lr], *(const f32x4*)(p.wsum + nb), bias ? *(const f32x4*)(bias + nb) : (f32x4){0.f, 0.f, 0.f, 0.f});
        } else if (bias) {
          const f32x4 bv = *(const f32x4*)(bias + nb);
          v += bv;
        }
        if (p.add) {
          const f32x4 av = *(const f32x4*)(p.add + (size_t)arow * p.ldadd + nb);
          v += av;
        }
        if constexpr (EPI == UD_EPI_F16 || EPI == UD_EPI_QKV) {
          half4 h;
#pragma unroll
          for (int r = 0; r < 4; ++r) h[r] = (half_t)ud_act4_t<ACT>(v)[r];
          *(half4*)((half_t*)out + (size_t)orow * p.ldc + nb) = h;
        } else if constexpr (EPI == UD_EPI_F32) {
          float* dst = (float*)out + (size_t)orow * p.ldc + nb;
          if constexpr (!PRELOADED) {
            if (p.accumulate) {
              const f32x4 old = *(const f32x4*)dst;
              v += old;
            }
          }
          if (p.act == UD_ACT_GELU) {
            v = ud_act4_t<UD_ACT_GELU>(v);
          } else if (p.act == UD_ACT_CLAMPEXP) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = ud_clampexp(v[r]);
          }
          if (p.accumulate != 2) *(f32x4*)dst = v;     // accumulate == 2: the fp32 stream dies here, only the fp16 copy is consumed
          if (p.max_out) {                             // running element-wise maximum over successive launches (utils/misc.py:18-21 max_stack)
            float* mp = p.max_out + (size_t)orow * p.ldc + nb;
            f32x4 mv = v;
            if (!p.max_init) {
              const f32x4 mo = *(const f32x4*)mp;
#pragma unroll
              for (int r = 0; r < 4; ++r) mv[r] = fmaxf(mo[r], v[r]);
            }
            *(f32x4*)mp = mv;
          }
          ud_row_stats_acc(v, rs1, rs2);
          if (out2) {
            half4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = (half_t)ud_act4_t<ACT>(v)[r];
            *(half4*)((half_t*)out2 + (size_t)orow * p.ldc2 + nb) = h;
          }
        }
      }
      if constexpr (EPI == UD_EPI_F32 && TN == 4) {
        if (p.row_stats_out) ud_row_stats_store(p, rs1, rs2, orow, nbase, lane, mok);
      }
    }
  }
}

template <int TM, int TN, int EPI, bool SWAP, bool PRELOADED = false, int TMA = TM>
__device__ __forceinline__ void gemm_epilogue(const UdGemm& p, f32x4 (&acc)[TMA][TN], int mbase, int nbase, int lane, const float* bias,
                                              char* out, char* out2, const float* w2, float b2, float post_add, char* stage,
                                              const float* lnst = nullptr) {
  const int a = (EPI == UD_EPI_F16 || EPI == UD_EPI_QKV) ? p.act : p.act2;
  if constexpr (EPI == UD_EPI_HEAD || EPI == UD_EPI_QKV) {
    gemm_epilogue_impl<TM, TN, EPI, SWAP, PRELOADED, UD_ACT_NONE>(p, acc, mbase, nbase, lane, bias, out, out2, w2, b2, post_add, stage, lnst);
  } else if constexpr (EPI == UD_EPI_F16) {
    if (a == UD_ACT_GELU) gemm_epilogue_impl<TM, TN, EPI, SWAP, PRELOADED, UD_ACT_GELU>(p, acc, mbase, nbase, lane, bias, out, out2, w2, b2, post_add, stage, lnst);
    else if (a == UD_ACT_LRELU) gemm_epilogue_impl<TM, TN, EPI, SWAP, PRELOADED, UD_ACT_LRELU>(p, acc, mbase, nbase, lane, bias, out, out2, w2, b2, post_add, stage, lnst);
    else gemm_epilogue_impl<TM, TN, EPI, SWAP, PRELOADED, UD_ACT_NONE>(p, acc, mbase, nbase, lane, bias, out, out2, w2, b2, post_add, stage, lnst);
  } else {
    if (a == UD_ACT_LRELU) gemm_epilogue_impl<TM, TN, EPI, SWAP, PRELOADED, UD_ACT_LRELU>(p, acc, mbase, nbase, lane, bias, out, out2, w2, b2, post_add, stage);
    else gemm_epilogue_impl<TM, TN, EPI, SWAP, PRELOADED, UD_ACT_NONE>(p, acc, mbase, nbase, lane, bias, out, out2, w2, b2, post_add, stage);
  }
}

// LDS fragment read the compiler does not see as an LDS access.  With ordinary ds_reads in the loop, the waitcnt pass puts
// `s_waitcnt vmcnt(0)` in front of them whenever a global_load_lds is in flight (it cannot tell the ring stages apart), which
// turns any ring deeper than two stages back into a two-stage one.  The caller owns the lgkmcnt wait (ud_lds_wait16).
template <int OFF>
__device__ __forceinline__ half8 ud_lds_read16_raw(unsigned addr) {
  half8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
__device__ __forceinline__ void ud_lds_wait16(half8 (&a)[2][4], half8 (&b)[2][4]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3]),
                 "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[0][3]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2]), "+v"(b[1][3]));
}

// NST = stages of the operand ring (NST - 1 K-tiles in flight).  Two stages are enough when two workgroups share a CU; with one
// workgroup per CU (fewer tiles than CUs) the loop waits on the DMA round trip every K-tile, and a deeper ring hides it.
template <class C, int EPI, int AMODE, bool SWAP, int NST = 2, bool SPLIT = false>
__device__ __forceinline__ void gemm_body(const UdGemm& p, char* smem, int m0, int n0, const half_t* A, const half_t* W,
                                          const float* bias, char* out, char* out2, const float* w2, float b2, float post_add,
                                          int tile = 0, int half_k = 0) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv / C::NWN, wn = wv % C::NWN;
  const int nk = SPLIT ? (p.K >> 7) : (p.K >> 6);          // SPLIT: this workgroup multiplies K-tiles [kbase, kbase + nk)
  const int kbase = SPLIT ? half_k * nk : 0;

  // ---------------- loader geometry: lane -> (row within 8-row group, chunk position) ----------------
  const int lrow = lane >> 3;   // 0..7
  const int cpos = lane & 7;    // chunk slot inside the 128-byte LDS row
  const half_t* aptr[C::A_INSTR];
  const half_t* bptr[C::B_INSTR];
  int a_csrc[C::A_INSTR];
  ConvLane cl[C::A_INSTR];
  const float inv_cc = (AMODE != UD_A_DENSE) ? 1.0f / (float)(p.Cin >> 3) : 0.0f;
#pragma unroll
  for (int i = 0; i < C::A_INSTR; ++i) {
    const int r = (wv * C::A_INSTR + i) * 8 + lrow;
    const int cs = cpos ^ ((r >> 1) & 7);
    a_csrc[i] = cs;
    int m = m0 + r;
    m = m < p.M ? m : p.M - 1;
    if constexpr (AMODE == UD_A_DENSE) {
      aptr[i] = A + (size_t)m * p.lda + cs * 8;
    } else {
      const int img = m / p.rows_img;
      const int pp = m - img * p.rows_img;
      const int y = pp / p.Wimg;
      cl[i].base = (long long)img * p.img_stride;
      cl[i].y = y;
      cl[i].x = pp - y * p.Wimg;
      cl[i].valid = pp < p.Himg * p.Wimg;
    }
  }
#pragma unroll
  for (int i = 0; i < C::B_INSTR; ++i) {
    const int r = (wv * C::B_INSTR + i) * 8 + lrow;
    const int cs = cpos ^ ((r >> 1) & 7);
    int n = n0 + r;
    n = n < p.N ? n : p.N - 1;
    bptr[i] = W + (size_t)n * p.ldw + cs * 8;
  }

  // split-fp16 products by K concatenation (UdGemm.a_wrap / w_wrap): the K index wraps around once inside the narrower operand
  const int a_wt = p.a_wrap > 0 ? (AMODE == UD_A_DENSE ? p.a_wrap >> 6 : p.a_wrap) : 0x7fffffff;
  const int w_wt = p.w_wrap > 0 ? p.w_wrap >> 6 : 0x7fffffff;
  auto issue = [&](int kt, int stage) {
    char* sb = smem + stage * C::STAGE_BYTES;
    const int kb = kt >= w_wt ? kt - w_wt : kt;
#pragma unroll
    for (int i = 0; i < C::A_INSTR; ++i) {
      const half_t* src;
      if constexpr (AMODE == UD_A_DENSE) {
        src = aptr[i] + (kt >= a_wt ? kt - a_wt : kt) * 64;
      } else {
        const int kc = kt * 8 + a_csrc[i];                       // 8-channel chunk index along K = (tap, cin)
        const int tap = (int)(((float)kc + 0.5f) * inv_cc);
        int cch = (kc - tap * (p.Cin >> 3)) << 3;
        cch = cch >= a_wt ? cch - a_wt : cch;
        const int t3 = (tap * 11) >> 5;                           // tap / 3 for tap < 12
        int yy = cl[i].y + t3 - 1;
        int xx = cl[i].x + (tap - t3 * 3) - 1;
        bool ok = cl[i].valid && tap < 9;
        if constexpr (AMODE == UD_A_CONV3_ZERO) {
          ok = ok && (unsigned)yy < (unsigned)p.Himg && (unsigned)xx < (unsigned)p.Wimg;
        } else {
          yy = yy < 0 ? -yy : (yy >= p.Himg ? 2 * p.Himg - 2 - yy : yy);
          xx = xx < 0 ? -xx : (xx >= p.Wimg ? 2 * p.Wimg - 2 - xx : xx);
        }
        src = ok ? A + cl[i].base + ((long long)(yy * p.Wimg + xx)) * p.cstride + p.coff + cch : (const half_t*)p.zeros;
      }
      ud_glds16(src, sb + (wv * C::A_INSTR + i) * 1024);
    }
#pragma unroll
    for (int i = 0; i < C::B_INSTR; ++i) ud_glds16(bptr[i] + kb * 64, sb + C::A_BYTES + (wv * C::B_INSTR + i) * 1024);
  };

  // ---------------- fragment read geometry ----------------
  const int frow = lane & 15;
  const int fq = lane >> 4;              // k-chunk within a 32-wide k-step
  const int fswz = frow >> 1;            // (row >> 1) & 7 for row = 16*t + frow
  const int a_row_off = (wm * C::WM + frow) * 128;
  const int b_row_off = C::A_BYTES + (wn * C::WN + frow) * 128;

  f32x4 acc[C::TM][C::TN];
#pragma unroll
  for (int i = 0; i < C::TM; ++i)
#pragma unroll
    for (int j = 0; j < C::TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  constexpr bool PRE = (EPI == UD_EPI_F32) && SWAP;
  constexpr int LOADS = C::A_INSTR + C::B_INSTR;           // DMA instructions per thread per K-tile (vmcnt units)
  static_assert(NST == 2 || NST == 4, "ring depth");
  auto mfma_kstep = [&](half8 (&af)[2][C::TM], half8 (&bf)[2][C::TN], auto KS_) {
    constexpr int ks = decltype(KS_)::value;
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
      for (int j = 0; j < C::TN; ++j) {
        if constexpr (SWAP)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
        else
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ks][i], bf[ks][j], acc[i][j], 0, 0, 0);
      }
  };
  if constexpr (NST > 2) {
    // ---- one workgroup per CU (fewer tiles than CUs): 4-stage ring, software-pipelined over K-tiles.
    // Iteration kt:  [fragments of kt are in registers, read during kt-1]  16 MFMAs (k-step 0)  ->  wait for the DMA of kt+1,
    // barrier  ->  DMA of kt+3 into the stage kt-1 left, fragment reads of kt+1 into the other register set  ->  16 MFMAs
    // (k-step 1) covering those reads.  Nothing of the LDS or DMA round trips is exposed unless a K-tile takes longer than
    // 1.5 iterations to arrive.  Fragment reads are raw asm (ud_lds_read16_raw explains why), the barrier is the bare s_barrier.
    static_assert(C::TM == 4 && C::TN == 4, "deep ring: 128 x 128 tiles only");
    if constexpr (PRE) {   // residual first: loads retire in order, so waiting for K-tile 0 below also covers it
      if (p.accumulate && half_k == 0) gemm_preload_acc<C::TM, C::TN>(p, acc, m0 + wm * C::WM, n0 + wn * C::WN, lane, out);
    }
#pragma unroll
    for (int st = 0; st < NST - 1; ++st)
      if (st < nk) issue(kbase + st, st);
    const unsigned lds0 = (unsigned)(size_t)smem;          // low half of the flat address of an LDS object = its LDS offset
    unsigned aa[2], ba[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      aa[ks] = lds0 + a_row_off + (((ks * 4 + fq) ^ fswz) << 4);
      ba[ks] = lds0 + b_row_off + (((ks * 4 + fq) ^ fswz) << 4);
    }
    auto read_frags = [&](half8 (&af)[2][4], half8 (&bf)[2][4], int stage) {
      const unsigned so = (unsigned)stage * C::STAGE_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const unsigned a = aa[ks] + so, b = ba[ks] + so;
        af[ks][0] = ud_lds_read16_raw<0>(a); af[ks][1] = ud_lds_read16_raw<2048>(a);
        af[ks][2] = ud_lds_read16_raw<4096>(a); af[ks][3] = ud_lds_read16_raw<6144>(a);
        bf[ks][0] = ud_lds_read16_raw<0>(b); bf[ks][1] = ud_lds_read16_raw<2048>(b);
        bf[ks][2] = ud_lds_read16_raw<4096>(b); bf[ks][3] = ud_lds_read16_raw<6144>(b);
      }
    };
    auto wait_tile = [&](int kt) {       // own DMA of K-tile kt has landed; K-tiles kt+1 .. (at most two) may still be in flight
      const int later = nk - 1 - kt;
      if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS) : "memory");
      else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    half8 fa0[2][4], fb0[2][4], fa1[2][4], fb1[2][4];
    wait_tile(0);
    __builtin_amdgcn_s_barrier();
    read_frags(fa0, fb0, 0);
    auto step = [&](half8 (&ca)[2][4], half8 (&cb)[2][4], half8 (&na)[2][4], half8 (&nb)[2][4], int kt) {
      ud_lds_wait16(ca, cb);
      __builtin_amdgcn_sched_barrier(0);
      mfma_kstep(ca, cb, IntTag<0>{});
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < nk) {
        // at this point K-tile kt+2 is the only later one issued (kt+3 goes out below)
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 3 < nk) issue(kbase + kt + 3, (kt + 3) & 3);
        read_frags(na, nb, (kt + 1) & 3);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma_kstep(ca, cb, IntTag<1>{});
      __builtin_amdgcn_sched_barrier(0);
    };
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      step(fa0, fb0, fa1, fb1, kt);
      step(fa1, fb1, fa0, fb0, kt + 1);
    }
    if (kt < nk) step(fa0, fb0, fa1, fb1, kt);
  } else {
    issue(0, 0);
    if constexpr (PRE) {
      if (p.accumulate) gemm_preload_acc<C::TM, C::TN>(p, acc, m0 + wm * C::WM, n0 + wn * C::WN, lane, out);
    }
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
      const char* sb = smem + (kt & 1) * C::STAGE_BYTES;
      // all 16 fragment reads of the K-tile go out back to back, then the 32 MFMAs: one exposed LDS round trip per K-tile and
      // wave.  Left alone the compiler serialises four read -> lgkmcnt(0) -> 8 MFMA groups per K-tile (fewest registers).
      half8 af[2][C::TM], bf[2][C::TN];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        __builtin_amdgcn_sched_barrier(0);
        const int coff = ((ks * 4 + fq) ^ fswz) << 4;
#pragma unroll
        for (int i = 0; i < C::TM; ++i) af[ks][i] = *(const half8*)(sb + a_row_off + i * 2048 + coff);
#pragma unroll
        for (int j = 0; j < C::TN; ++j) bf[ks][j] = *(const half8*)(sb + b_row_off + j * 2048 + coff);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma_kstep(af, bf, IntTag<0>{});
      __builtin_amdgcn_sched_barrier(0);
      mfma_kstep(af, bf, IntTag<1>{});
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if constexpr (SPLIT) {
    // ---- join of the two K halves (workgroups on different CUs, in general on different XCDs with private L2s).
    // Partials travel with system-scope cache policy (sc0 sc1: stores write through, loads bypass the non-coherent L2), ordered
    // by vmcnt(0) around one ticket atomic per workgroup: no agent-scope fence -- __threadfence() writes back / invalidates the
    // whole L2 and costs 20-50 us per launch here (tools/ubench/partial_exchange.hip: 56 us vs 7.8 us for 256 x 64 KB).
    static_assert(C::TM == 4 && C::TN == 4 && NST == 4, "split: 128 x 128 ring kernel only");
    f32x4* mine = (f32x4*)p.splitk_ws + ((size_t)(tile * 2 + half_k) * 4 + wv) * 1024 + lane;
    const f32x4* other = (const f32x4*)p.splitk_ws + ((size_t)(tile * 2 + (half_k ^ 1)) * 4 + wv) * 1024 + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 v = acc[i][j];
        // s_nop: the data VGPRs of a 16-byte store must not be rewritten in the next cycles (the compiler guards only its own stores)
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 2" ::"v"(mine + (i * 4 + j) * 64), "v"(v) : "memory");
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                       // every wave's partial is out (and nobody reads operand tiles any more)
    unsigned* tk = (unsigned*)smem;
    if (tid == 0) *tk = atomicAdd((unsigned*)p.splitk_cnt + tile, 1u);
    __syncthreads();
    const unsigned ticket = *tk;
    if ((ticket & 1u) == 0) return;        // first of the pair: the partner finishes the tile (tickets: parity, never reset)
    f32x4 o[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(o[q]) : "v"(other + q * 64) : "memory");
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7]), "+v"(o[8]), "+v"(o[9]),
                   "+v"(o[10]), "+v"(o[11]), "+v"(o[12]), "+v"(o[13]), "+v"(o[14]), "+v"(o[15])
                 :
                 : "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] += o[i * 4 + j];     // a + b == b + a: the sum does not depend on which half came last
    __syncthreads();                       // the ticket word is read: LDS may become the store staging area
  }
  char* stage = nullptr;
  if constexpr ((EPI == UD_EPI_F16 || EPI == UD_EPI_QKV) && SWAP && C::TN == 4 && C::TM == 4) {
    __syncthreads();                       // every wave is done reading operand tiles: LDS becomes the store staging area
    stage = smem + wv * 9216;
  }
  gemm_epilogue<C::TM, C::TN, EPI, SWAP, PRE>(p, acc, m0 + wm * C::WM, n0 + wn * C::WN, lane, bias, out, out2, w2, b2, post_add, stage);
}

template <class C, int EPI, int AMODE, int NST = 2, bool SPLIT = false>
__global__ __launch_bounds__(256) void gemm_kernel(const UdGemm p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_n = (p.N + C::BN - 1) / C::BN;
  const int tiles_m = (p.M + C::BM - 1) / C::BM;
  const int nblk = tiles_m * tiles_n;
  int bid = blockIdx.x;
  if constexpr (SPLIT) {                // blocks 2t / 2t+1 (different XCDs) = K halves 0 / 1 of tile t; fewer blocks than CUs
    const int tile_m = (bid >> 1) / tiles_n, tile_n = (bid >> 1) - tile_m * tiles_n;
    gemm_body<C, EPI, AMODE, true, NST, true>(p, smem, tile_m * C::BM, tile_n * C::BN, (const half_t*)p.A, (const half_t*)p.W, p.bias,
                                              (char*)p.out, (char*)p.out2, p.w2, p.b2, p.post_add, bid >> 1, bid & 1);
    return;
  }
  {  // bijective XCD-aware remap: block b runs on XCD b % 8 -> give each XCD a contiguous range of tiles
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = bid / tiles_n;
  const int tile_n = bid - tile_m * tiles_n;
  const int m0 = tile_m * C::BM, n0 = tile_n * C::BN;
  const int g = blockIdx.z;
  const half_t* A = (const half_t*)p.A + (long long)g * p.gA;
  const half_t* W = (const half_t*)p.W + (long long)g * p.gW;
  const float* bias = p.bias ? p.bias + (long long)g * p.gBias : nullptr;
  char* out = (char*)p.out + (long long)g * p.gOut * ((EPI == UD_EPI_F16 || EPI == UD_EPI_QKV) ? 2 : 4);
  char* out2 = p.out2 ? (char*)p.out2 + (long long)g * p.gOut2 * 2 : nullptr;
  const float* w2 = p.w2 ? p.w2 + (long long)g * p.gW2 : nullptr;
  const float b2 = g == 0 ? p.b2 : p.b2_g1;
  const float post_add = g == 0 ? p.post_add : p.post_add_g1;
  if constexpr (EPI == UD_EPI_QKV) {
    if (n0 >= p.vsplit) {
      gemm_body<C, EPI, AMODE, false, NST, false>(p, smem, m0, n0, A, W, bias, out, out2, w2, b2, post_add);
      return;
    }
  }
  gemm_body<C, EPI, AMODE, true, NST, false>(p, smem, m0, n0, A, W, bias, out, out2, w2, b2, post_add);
}

// ================================================================================================================
// Large-tile kernel: 256 x 256 (MH = 4) or 192 x 256 (MH = 3) output tile, 8 waves (2 along m x 4 along n), K-tile = 64,
// 2-stage LDS ring (A [BM][64] + B [256][64] halves per stage; one workgroup per CU), persistent over its tile list.
//  * full 128-byte rows per DMA lane group (a 64-byte-row variant re-fetched every L2 line twice and was L2-bound);
//  * the whole next K-tile is issued (global_load_lds_dwordx4) at the top of the current one, three of the four 16-MFMA
//    phases ahead of its first use; one vmcnt(0) + raw s_barrier per K-tile (64 MFMAs per wave);
//  * the K-tile is walked as 2 k-steps x 2 m-halves; fragment reads run one phase ahead of their MFMAs in registers
//    (A m-half sets alternate, B k-step sets alternate) -- 0.375 ds_read_b128 per MFMA;
//  * same source-side XOR swizzle as the 128x128 kernel (chunk ^= (row >> 1) & 7): conflict-free ds_read_b128;
//  * the K-tile stream is CONTINUOUS across the workgroup's tiles: the last K-tile of tile t issues K-tile 0 of tile t+1, so
//    the epilogue of t (VALU + fire-and-forget stores, no LDS) runs while the next operands are already landing
//    (measured with tools/trace_gemm.py: prologue 1.1 us + epilogue 7-10 us per 21-26 us K loop before);
//  * fp16 outputs: v_permlane16_swap pairs neighbouring 16-column MFMA tiles so every lane stores 16 contiguous bytes
//    (64-byte row segments) straight from registers -- no LDS staging pass, no lgkmcnt waits.
// ================================================================================================================
// MH = m-tiles (of 16 rows) per wave per m-half: 4 -> 256-row tiles, 3 -> 192-row tiles (tile-count quantisation on the
// 256 CUs decides which one a GEMM gets: e.g. M = 11008, N = 1024 is 172 tiles of 256 rows (67 % of the CUs busy) but 232
// tiles of 192 rows (91 %)).
template <int MH>
struct BigCfg {
  static constexpr int BM = 64 * MH;                 // 2 waves along m x 2 halves x MH x 16
  static constexpr int A_BYTES = BM * 128;
  static constexpr int STAGE = A_BYTES + 32768;
  static constexpr int A_LOADS = BM / 64;            // 16-byte chunks per thread per K-tile (512 threads)
};

template <bool B> struct BoolTag { static constexpr bool value = B; };
constexpr int LNC_TILES = UD_LNC_TILES;
constexpr int LNC_LDS = 2 * UD_LNC_PLANE * 4;    // the two planes, behind the operand ring

// two packed-fp16 dwords of column tiles j / j+1 -> 8 consecutive columns per lane (see the layout note in the kernel)
__device__ __forceinline__ void ud_pair16(unsigned& a, unsigned& b) {
  const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
__device__ __forceinline__ void ud_pair32(unsigned& a, unsigned& b) {     // a: lanes 32-63 <-> b: lanes 0-31
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
__device__ __forceinline__ unsigned ud_pack2(float x, float y) {
  f32x2 v; v[0] = x; v[1] = y;
  const half2v h = __builtin_convertvector(v, half2v);
  return __builtin_bit_cast(unsigned, h);
}
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// scheduling-region pattern: R ds_reads spread between M MFMAs (q MFMAs, 1 read, q MFMAs, 1 read, ..., rest), so the matrix pipe is
// not left idle while a cluster of fragment reads issues (reads of a phase target registers that phase's MFMAs do not use)
template <int R, int M>
__device__ __forceinline__ void ud_interleave_reads() {
  constexpr int q = M / (R + 1);
  if constexpr (R >= 1) { __builtin_amdgcn_sched_group_barrier(0x8, q, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
  if constexpr (R >= 2) { __builtin_amdgcn_sched_group_barrier(0x8, q, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
  if constexpr (R >= 3) { __builtin_amdgcn_sched_group_barrier(0x8, q, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
  if constexpr (R >= 4) { __builtin_amdgcn_sched_group_barrier(0x8, q, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
  if constexpr (R >= 5) { __builtin_amdgcn_sched_group_barrier(0x8, q, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
  if constexpr (R >= 6) { __builtin_amdgcn_sched_group_barrier(0x8, q, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
  if constexpr (R >= 7) { __builtin_amdgcn_sched_group_barrier(0x8, q, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
  if constexpr (R >= 8) { __builtin_amdgcn_sched_group_barrier(0x8, q, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
  __builtin_amdgcn_sched_group_barrier(0x8, M - q * R, 0);
}

// BAL (row-balanced schedule, dense A, MH = 4): instead of whole 256-row tiles dealt round-robin (fc1: 688 tiles on 256 CUs = 2.69
// rounds, the third one 69 % full), every column of 256 outputs gets floor(256 / tiles_n) workgroups and each of them a contiguous
// span of ceil(M / 64) / that many 64-row units (+-1), cut into tiles of 64 * {2, 3, 4} rows of near-equal height (fc1: 704 rows =
// 256 + 256 + 192 per CU): all CUs finish together.  A tile of 64 * MHC rows runs the MHC-instantiation of the tile body.
// LNC (consumer of a folded LayerNorm, UdGemm.row_stats_in): A holds the RAW fp16 rows; the per-row (rstd, -mean rstd) of a tile are
// copied to a small LDS table by LDS-DMA together with the tile's first operand K-tile (at kernel start for the first tile, from the last
// K-tile of the previous tile otherwise: the same vmcnt(0) + barrier covers both), two tables alternating per tile; the epilogue applies
// them -- no register lives across the K loop or the epilogues for it, any number of tiles per workgroup.  (A first version reduced the producer's 16 partial pairs per row here, per tile: 32 live registers across the
// epilogue, 55-340 spilled registers in the Q|K / V^T instantiations, +14..16 us per launch -- slower than the LayerNorm kernel it
// replaced; the reduction is now ud_row_stats_finalize, one thread per row.)
// GRP: a grouped problem (UdGemm.groups: G independent GEMMs of equal shape whose A rows / output rows are stacked along M and whose
// weights sit gW apart) run as ONE tile list over G * M rows: the group of a tile is m0 / grp_rows (grp_rows % tile height == 0, so no
// tile straddles two groups) and only moves the W rows and the bias; with gA == 0 all groups read the same A.  The 128-row kernel runs
// such problems as blockIdx.z slices at ~480 TFLOP/s (K = 512: four short K loops per CU); here they share the persistent tile stream.
// W3 (round 4; dense A, tile list, K >= 128): the WEIGHT operand gets a 3-deep LDS ring and is fetched TWO K-tiles ahead, the activation
// operand keeps its 2-deep ring (192-row tiles: 2 x 24 KB + 3 x 32 KB = 144 KB).  In the step every layer's weights are cold -- last
// touched one step = ~3 GB of traffic ago, so each of the 8 XCDs pulls its W panels from HBM -- while the activations were written by the
// previous launch and mostly sit in the Infinity Cache; with everything one K-tile (~1.3 us) ahead the HBM round trip of the weight
// lines was exposed in every K-tile (tools/r4_insitu.py: fc2 95 us with warm operands, 107 us when only the weights rotate through 24
// layers, 114 us in the step).  The end-of-K-tile wait becomes a COUNTED vmcnt(4): A(kt+1) and W(kt+1) have landed, the four DMA
// instructions of W(kt+2) stay in flight across the barrier (loads retire in order among themselves; stores only make the count
// conservative).  The stream stays continuous across the workgroup's tiles: W runs two K-tiles ahead over the tile boundary as well.
// SPK (round 6; tile list with at most 128 tiles, K a multiple of 128): TWO workgroups per output tile, each multiplying one half of K --
// blocks 2t / 2t+1 = K halves 0 / 1 of tile t.  A one-round list that fills less than half of the CUs (the decoder's stage-0 3x3 convolutions:
// M = 11008, N = 512, K = 4608 -> 116 tiles of 192 x 256; they ran as 344 tiles of 128 x 128 at ~540 TFLOP/s) gets twice the workgroups and
// half the K loop; the partial tiles meet through the same fence-free exchange as the 128 x 128 kernel's split (system-scope write-through
// stores, vmcnt(0), one ticket per tile; the workgroup that draws the odd ticket adds its partner's partial and runs the epilogue; a + b is
// order-independent, so the bits do not depend on which half finishes).
template <int MH, int EPI, int AMODE, bool BAL = false, bool LNC = false, bool GRP = false, bool W3 = false, bool SPK = false>
__global__ __launch_bounds__(512) void gemm256_kernel(const UdGemm p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using C = BigCfg<MH>;
  static_assert(!W3 || (AMODE == UD_A_DENSE && !BAL && !GRP), "W3: dense A, tile list");
  static_assert(!SPK || (!BAL && !LNC && !GRP && !W3 && (EPI == UD_EPI_F16 || EPI == UD_EPI_F32)), "K split: plain tile list, fp16 / fp32 epilogues");
  constexpr int RING_BYTES = W3 ? 2 * C::A_BYTES + 3 * 32768 : 2 * C::STAGE;     // LDS behind the operand ring: LNC tables / ticket flag
  static_assert(!LNC || (AMODE == UD_A_DENSE && (EPI == UD_EPI_F16 || EPI == UD_EPI_QKV)), "LayerNorm-folded consumer: dense A, fp16 outputs");
  constexpr int BM = C::BM;
  constexpr int TM = 2 * MH;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv >> 2, wn = wv & 3;
  const int nk = SPK ? (p.K >> 7) : (p.K >> 6);             // SPK: this workgroup multiplies K-tiles [kbase, kbase + nk)
  const int kbase = SPK ? (int)(blockIdx.x & 1) * nk : 0;
  const int tiles_n = (p.N + 255) >> 8;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int nblk = tiles_m * tiles_n;
  const half_t* A = (const half_t*)p.A;
  const half_t* W = (const half_t*)p.W;

  // tile list entry -> (m0, n0): XCD-aware (contiguous range of the list per XCD; gridDim.x is a multiple of 8 or == nblk, so
  // t % 8 == blockIdx.x % 8 == the XCD) + group-M walk (8 row-tiles x all column tiles at a time share A and W panels in L2).
  // A list that fits one round (tiles <= 256 CUs: proj / fc2 / patch embed at bs = 8, most decoder GEMMs) is walked row-major instead
  // (groups of ONE row tile): an XCD's range of 29 tiles then covers 8-9 row tiles with all their column tiles, where the 8-row groups
  // (32 tiles) straddled the ranges and every XCD touched ~11 row tiles, re-fetching their A rows through the fabric.  Measured on the
  // proj / fc2 class (profiles/r05_tile_map_ab.txt): 188.2 -> 147.5 MB read per launch (modelled 141: A once + W per XCD + the fp32
  // residual), step +0.3 %.  Giving every XCD whole row tiles (8 x 32 slots, the surplus workgroups exit) reads 141.8 MB but loads two
  // XCDs with 32 tiles against 28 and cost 1 % of the step: not kept.
  auto decode = [&](int t, int& m0, int& n0) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = t & 7, idx = t >> 3;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int GM = nblk <= 256 ? 1 : 8;
    const int gsz = GM * tiles_n;
    const int grp = bid / gsz;
    const int first_m = grp * GM;
    const int gm = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int rem = bid - grp * gsz;
    m0 = (first_m + rem % gm) * BM;
    n0 = (rem / gm) << 8;
  };

  // ---------------- loader: thread owns chunks tid + 512*i -> rows (tid >> 3) + 64*i.  Operands are addressed through buffer
  // descriptors (buffer_load_dwordx4 ... lds): one 32-bit byte offset per chunk in a VGPR, the K-tile advance in an SGPR.
  const int lrow = tid >> 3;
  const int csrc = (tid & 7) ^ ((lrow >> 1) & 7);
  const ud_rsrc_t rA = ud_make_rsrc(A, 0x80000000u);
  const ud_rsrc_t rW = ud_make_rsrc(W, 0x80000000u);
  unsigned pa[C::A_LOADS];        // dense: byte offset of the lane's chunk in K-tile 0; conv: byte offset of the image
  unsigned cyx[C::A_LOADS];       // conv: (y << 16) | x of the row's pixel (y = 0x4000 for rows past the image: always padding)
  unsigned pb[4];
  const float inv_cc = (AMODE != UD_A_DENSE) ? 1.0f / (float)(p.Cin >> 3) : 0.0f;
  auto setupA = [&](int m0, int mh) {
    int grp = 0;
    if constexpr (GRP) grp = m0 / p.grp_rows;
#pragma unroll
    for (int i = 0; i < C::A_LOADS; ++i) {
      if (i >= mh) continue;
      int m = m0 + lrow + 64 * i;
      m = m < p.M ? m : p.M - 1;
      if constexpr (GRP) {
        if (p.gA == 0) m -= grp * p.grp_rows;                      // every group multiplies the same A rows
      }
      if constexpr (AMODE == UD_A_DENSE) {
        pa[i] = ((unsigned)m * (unsigned)p.lda + csrc * 8) * 2u;
      } else {
        const int img = m / p.rows_img;
        const int pp = m - img * p.rows_img;
        const int y = pp / p.Wimg;
        pa[i] = (unsigned)((long long)img * p.img_stride) * 2u;
        cyx[i] = pp < p.Himg * p.Wimg ? ((unsigned)y << 16) | (unsigned)(pp - y * p.Wimg) : 0x40000000u;
      }
    }
  };
  auto setupB = [&](int m0, int n0) {
    int grp = 0;
    if constexpr (GRP) grp = m0 / p.grp_rows;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int n = n0 + lrow + 64 * i;
      n = n < p.N ? n : p.N - 1;
      pb[i] = ((unsigned)n * (unsigned)p.ldw + csrc * 8) * 2u;
      if constexpr (GRP) pb[i] += (unsigned)grp * (unsigned)p.gW * 2u;
    }
  };
  auto setup = [&](int m0, int n0, int mh) {
    setupA(m0, mh);
    setupB(m0, n0);
  };
  // split-fp16 products by K concatenation (UdGemm.a_wrap / w_wrap): the K index wraps around once inside the narrower operand
  const int a_wt = p.a_wrap > 0 ? (AMODE == UD_A_DENSE ? p.a_wrap >> 6 : p.a_wrap) : 0x7fffffff;
  const int w_wt = p.w_wrap > 0 ? p.w_wrap >> 6 : 0x7fffffff;
  // LDS address of ring slot `stg` of either operand (W3: separate rings, A 2-deep then W 3-deep; else [A | W] per stage)
  auto a_stage = [&](int stg) -> char* { return smem + stg * (W3 ? C::A_BYTES : C::STAGE); };
  auto b_stage = [&](int stg) -> char* { return W3 ? smem + 2 * C::A_BYTES + stg * 32768 : smem + stg * C::STAGE + C::A_BYTES; };
  auto issueA = [&](int kt, int stg, int mh) {
    char* sb = a_stage(stg) + wv * 1024;
    kt += kbase;
    if constexpr (AMODE == UD_A_DENSE) {
      const int ka = kt >= a_wt ? kt - a_wt : kt;
#pragma unroll
      for (int i = 0; i < C::A_LOADS; ++i)
        if (i < mh) ud_bufl16(rA, pa[i], ka * 128, sb + i * 8192);
    } else {
      // implicit-GEMM gather: this lane's 16-byte chunk = 8 channels of tap (kc / (Cin/8)); one tap decode per K-tile;
      // padding taps use an offset beyond the descriptor's range and read as zeros
      static_assert(AMODE != UD_A_CONV3_REFLECT, "large-tile kernel: zero-padded convolutions only");
      const int kc = kt * 8 + csrc;
      const int tap = (int)(((float)kc + 0.5f) * inv_cc);
      int cch = (kc - tap * (p.Cin >> 3)) << 3;
      cch = cch >= a_wt ? cch - a_wt : cch;
      const int t3 = (tap * 11) >> 5;
      const int dy = t3 - 1, dx = tap - t3 * 3 - 1;
#pragma unroll
      for (int i = 0; i < C::A_LOADS; ++i) {
        if (i >= mh) continue;
        const int yy = (int)(cyx[i] >> 16) + dy, xx = (int)(cyx[i] & 0xffff) + dx;
        const bool ok = tap < 9 && (unsigned)yy < (unsigned)p.Himg && (unsigned)xx < (unsigned)p.Wimg;
        const unsigned off = ok ? pa[i] + (unsigned)((yy * p.Wimg + xx) * p.cstride + p.coff + cch) * 2u : 0xfffffff0u;
        ud_bufl16(rA, off, 0, sb + i * 8192);
      }
    }
  };
  auto issueB = [&](int kt, int stg) {
    char* sb = b_stage(stg) + wv * 1024;
    kt += kbase;
    const int kb = kt >= w_wt ? kt - w_wt : kt;
#pragma unroll
    for (int i = 0; i < 4; ++i) ud_bufl16(rW, pb[i], kb * 128, sb + i * 8192);
  };
  auto issue = [&](int kt, int stg, int mh) {
    issueA(kt, stg, mh);
    issueB(kt, stg);
  };

  // ---------------- fragment read offsets inside a stage: row-major [row][64 halves], chunk index swizzled by (row >> 1) & 7
  const int fswz = (lane & 15) >> 1;
  const int c0 = ((lane >> 4) ^ fswz) << 4;          // k-step 0
  const int c1 = ((4 + (lane >> 4)) ^ fswz) << 4;    // k-step 1
  const int b_off = (wn * 64 + (lane & 15)) * 128;     // inside a W stage

  f32x4 acc[TM][4];
  half8 a0[MH], a1[MH], b0[4], b1[4];
  int stg = 0;                                 // ring stage holding the K-tile about to be multiplied (W3: of the A ring)
  int stgb = 0;                                // W3: the same for the 3-deep W ring

  constexpr bool ACC_EPI = (EPI == UD_EPI_F32);        // `out (+)= ...`: old values are preloaded into the accumulators

  // ---------------- LNC: (rstd, -mean) of the rows of EVERY tile this workgroup will visit, parked in LDS once at kernel start
  float* const lds_rstd = (float*)(smem + RING_BYTES);        // [LNC_TILES][256] rstd, then [LNC_TILES][256] -mean * rstd
  float* const lds_nm = lds_rstd + UD_LNC_PLANE;
  int tl = 0;                                                    // local index of the current tile = its table

  int m0, n0, mhc = MH;
  int t = SPK ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;          // classic: index into the tile list; balanced: index of the tile inside this workgroup's row span
  int bal_k = 0, bal_tb = 0, bal_te = 0, bal_u0 = 0, bal_n0 = 0;
  if constexpr (BAL) {
    const int G = gridDim.x;                       // = cpc * tiles_n (launch256bal)
    const int cpc = G / tiles_n;                   // workgroups per column of 256 outputs
    const int U = (p.M + 63) >> 6;                 // 64-row units per column
    // blocks b, b + 8, ... share an XCD (and its L2): give each XCD a compact block of the (row span x column) grid, 4 x 2 XCDs when
    // the grid divides (fc1: 4 row spans x 8 columns per XCD: 5.6 MB of A rows + 4 MB of W panels per L2 -- with one column pair
    // per XCD instead every L2 streamed all of A, 2.3x the fabric traffic, and the balanced schedule gained only 3 %)
    int col, r;
    if ((G & 7) == 0 && (cpc & 3) == 0 && (tiles_n & 1) == 0) {
      const int x = blockIdx.x & 7, j = blockIdx.x >> 3;          // XCD, slot inside the XCD (G / 8 slots)
      const int cx = tiles_n >> 1, rx = cpc >> 2;                  // columns / row spans per XCD
      col = (x & 1) * cx + j % cx;
      r = (x >> 1) * rx + j / cx;
    } else if (tiles_n == 1 && (G & 7) == 0) {
      // one column of tiles (the decoder's 256-channel 3x3 convolutions): every XCD takes a contiguous run of row spans, so the image rows two
      // neighbouring spans share as 3x3 halo meet in one L2
      col = 0;
      r = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    } else {
      const int lc = blockIdx.x;
      col = lc / cpc;
      r = lc - col * cpc;
    }
    const int base = U / cpc, extra = U - base * cpc;
    const int un = base + (r < extra ? 1 : 0);
    if (un == 0) return;
    bal_u0 = r * base + (r < extra ? r : extra);
    bal_k = (un + 3) >> 2;
    bal_tb = un / bal_k;
    bal_te = un - bal_tb * bal_k;
    bal_n0 = col << 8;
    t = 0;
  }
  auto tile_at = [&](int i, int& tm0, int& tn0, int& tmh) {
    if constexpr (BAL) {
      tmh = bal_tb + (i < bal_te ? 1 : 0);
      tm0 = (bal_u0 + i * bal_tb + (i < bal_te ? i : bal_te)) << 6;
      tn0 = bal_n0;
    } else {
      decode(i, tm0, tn0);
      tmh = MH;
    }
  };
  const int tcount = BAL ? bal_k : nblk;
  const int tstep = BAL ? 1 : (SPK ? nblk : (int)gridDim.x);      // SPK: one tile (half) per workgroup
  tile_at(t, m0, n0, mhc);
  setup(m0, n0, mhc);
  issue(0, 0, mhc);
  if constexpr (W3) issueB(1, 1);                  // W runs two K-tiles ahead (K >= 128 is a launch condition)
  // fill a statistics table by LDS-DMA (4 bytes per lane, lane-linear destination = 64 consecutive rows of one plane): no VGPR holds the
  // values and nothing waits here -- they land with the tile's first operand K-tile.  (Loading them through registers before the
  // first barrier cost 10-15 us per launch.)
  const ud_rsrc_t rS = ud_make_rsrc(LNC ? p.row_stats_in : nullptr, (unsigned)p.M * 8u);
  auto stats_dma = [&](int tm0, int tmh, int buf) {
    if (wv < 4 && wv < tmh) {
      int m = tm0 + wv * 64 + lane;
      m = m < p.M ? m : p.M - 1;
      ud_bufl4(rS, (unsigned)m * 8u, 0, lds_rstd + buf * 256 + wv * 64);
      ud_bufl4(rS, (unsigned)m * 8u + 4u, 0, lds_nm + buf * 256 + wv * 64);
    }
  };
  if constexpr (LNC) stats_dma(m0, mhc, 0);

  int trace_tile = 0;
  bool first = true;
  for (; t < tcount; t += tstep, ++trace_tile) {
    UD_STAMP(0);
    const int tn = t + tstep;
    const bool has_next = tn < tcount;
    int m0n = 0, n0n = 0, mhn = MH;
    if (has_next) tile_at(tn, m0n, n0n, mhn);
    const bool swap = !(EPI == UD_EPI_QKV && n0 >= p.vsplit);
    // ---- the tile body, instantiated per tile height (MHC row tiles of 16 per wave and m-half)
    auto tile_body = [&](auto MHT) {
    constexpr int MHC = decltype(MHT)::value;
    constexpr int TMC = 2 * MHC;
    constexpr int BMC = 64 * MHC;
    const int a_off = (wm * (BMC / 2) + (lane & 15)) * 128;
    const int mbase = m0 + wm * (BMC / 2), nbase = n0 + wn * 64;

#pragma unroll
    for (int i = 0; i < TMC; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // `out += A W^T`: the old fp32 values become the accumulators' initial state (the summation order of every output element
    // is then independent of where its row sits in the tile, i.e. of the image's position in the batch: infer() stays
    // bit-for-bit batch-permutation equivariant; an in-loop variant that streamed them in during the first K-tiles hid the
    // 7 us load burst but made the result depend on the batch position at the 2e-4 level after fp16 re-rounding).
    // Issued before the wait for the first operand tile: both bursts are in flight together.
    bool counted = false;        // the first tile's operand wait may leave the residual loads in flight (full tiles: their number is exact)
    if constexpr (ACC_EPI) {
      if (p.accumulate && kbase == 0) {                  // (K split: the old values enter through the first half only)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        if (first && (m0 + BMC <= p.M) && (n0 + 256 <= p.N) && p.rows_in == 0) {
          // full tile: TMC * 4 unconditional 16-byte loads per lane, issued AFTER the operand DMA of K-tile 0 (and of W(1)): loads retire in
          // order, so `vmcnt(TMC * 4)` below means "the operands have landed" while the 45 MB residual burst of all workgroups is still
          // streaming in -- the first MFMA phases start as their own rows arrive (the compiler's waits before each accumulator's first use)
          // instead of behind the whole burst (6.3 us serial prologue with the matrix pipes idle, DESIGN_HISTORY 8.2; measured: proj 37.5 -> 36.6 us
          // alone, fc2 and the step inside the noise, profiles/r05_resid_counted_ab.txt).  Same values, same
          // summation order: the old value is still the accumulator's initial state.
          const float* ob = (const float*)p.out + (size_t)(mbase + (ln & 15)) * p.ldc + nbase + 4 * (ln >> 4);
#pragma unroll
          for (int i = 0; i < TMC; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = *(const f32x4*)(ob + (size_t)(i * 16) * p.ldc + j * 16);
          counted = true;
        } else {
          gemm_preload_acc<TMC, 4>(p, acc, mbase, nbase, ln, (const char*)p.out);
        }
      }
    }
    if (first) {
      if (counted) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(TMC * 4) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      first = false;
    }
    {  // first fragments of this tile (its K-tile 0 landed before the barrier that ended the previous tile's K loop)
      const char* sa0 = a_stage(stg);
      const char* sb0 = b_stage(W3 ? stgb : stg);
#pragma unroll
      for (int i = 0; i < MHC; ++i) a0[i] = *(const half8*)(sa0 + a_off + i * 2048 + c0);
#pragma unroll
      for (int j = 0; j < 4; ++j) b0[j] = *(const half8*)(sb0 + b_off + j * 2048 + c0);
    }

    // ---- one K-tile: SW = operand order, LAST = last K-tile of the output tile (prefetches the next tile's first K-tile
    //      instead of this tile's next one)
    // PH (W3 only): 0 = a K-tile with two more to come, 1 = the second-last one (its W prefetch belongs to the next tile), 2 = the last
    auto ktile = [&](auto SW, auto LASTT, int kt, auto PHT) {
      constexpr bool SWAP = decltype(SW)::value;
      constexpr bool LAST = decltype(LASTT)::value;
      constexpr int PH = decltype(PHT)::value;
      const int stgb1 = stgb == 2 ? 0 : stgb + 1, stgb2 = stgb == 0 ? 2 : stgb - 1;     // W3: slots of W(kt+1), W(kt+2)
      const char* sa = a_stage(stg);
      const char* sb = b_stage(W3 ? stgb : stg);
      const char* san = a_stage(stg ^ 1);
      const char* sbn = b_stage(W3 ? stgb1 : stg ^ 1);
#define UD_MFMA_HALF(ROW0, AF, BF)                                                                               \
  _Pragma("unroll") for (int i = 0; i < MHC; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) {               \
    if constexpr (SWAP) acc[ROW0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(BF[j], AF[i], acc[ROW0 + i][j], 0, 0, 0); \
    else acc[ROW0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(AF[i], BF[j], acc[ROW0 + i][j], 0, 0, 0);   \
  }
      // ---- phase (k0, m-half 0): DMA of the whole next K-tile goes out first
      if constexpr (W3) {
        // issue order matters for the counted wait below: everything that must have landed by the end of this K-tile first, W(kt+2) last
        if constexpr (PH == 2) {
          if (has_next) {
            if constexpr (LNC) stats_dma(m0n, mhn, (tl + 1) & 1);
            setupA(m0n, mhn);
            issueA(0, stg ^ 1, mhn);
            issueB(1, stgb2);                                          // pb already points at the next tile (set in the second-last K-tile)
          }
        } else if constexpr (PH == 1) {
          issueA(kt + 1, stg ^ 1, MHC);
          if (has_next) {
            setupB(m0n, n0n);
            issueB(0, stgb2);
          }
        } else {
          issueA(kt + 1, stg ^ 1, MHC);
          issueB(kt + 2, stgb2);
        }
      } else if constexpr (LAST) {
        if (has_next) {
          setup(m0n, n0n, mhn);
          issue(0, stg ^ 1, mhn);
          if constexpr (LNC) stats_dma(m0n, mhn, (tl + 1) & 1);      // the table epilogue tl - 1 read: every wave is past it
        }
      } else {
        issue(kt + 1, stg ^ 1, MHC);
      }
#pragma unroll
      for (int i = 0; i < MHC; ++i) a1[i] = *(const half8*)(sa + a_off + (MHC + i) * 2048 + c0);
      UD_MFMA_HALF(0, a0, b0)
      ud_interleave_reads<MHC, 4 * MHC>();
      __builtin_amdgcn_sched_barrier(0);
      // ---- phase (k0, m-half 1)
#pragma unroll
      for (int i = 0; i < MHC; ++i) a0[i] = *(const half8*)(sa + a_off + i * 2048 + c1);
#pragma unroll
      for (int j = 0; j < 4; ++j) b1[j] = *(const half8*)(sb + b_off + j * 2048 + c1);
      UD_MFMA_HALF(MHC, a1, b0)
      ud_interleave_reads<MHC + 4, 4 * MHC>();
      __builtin_amdgcn_sched_barrier(0);
      // ---- phase (k1, m-half 0)
#pragma unroll
      for (int i = 0; i < MHC; ++i) a1[i] = *(const half8*)(sa + a_off + (MHC + i) * 2048 + c1);
      UD_MFMA_HALF(0, a0, b1)
      ud_interleave_reads<MHC, 4 * MHC>();
      __builtin_amdgcn_sched_barrier(0);
      // ---- phase (k1, m-half 1): next K-tile must have landed for every wave before anyone reads it
      if constexpr (W3) {
        // A(kt+1) and W(kt+1) landed; the 4 DMA instructions of W(kt+2), issued last, may stay in flight (none were issued: wait for all)
        if (PH == 0 || has_next) asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if constexpr (!LAST) {                   // (the next TILE's first fragments are read after the epilogue: 32 registers less there)
#pragma unroll
        for (int i = 0; i < MHC; ++i) a0[i] = *(const half8*)(san + a_off + i * 2048 + c0);
#pragma unroll
        for (int j = 0; j < 4; ++j) b0[j] = *(const half8*)(sbn + b_off + j * 2048 + c0);
      }
      UD_MFMA_HALF(MHC, a1, b1)
      if constexpr (!LAST) ud_interleave_reads<MHC + 4, 4 * MHC>();
      __builtin_amdgcn_sched_barrier(0);
#undef UD_MFMA_HALF
      stg ^= 1;
      if constexpr (W3) stgb = stgb1;
    };

    auto kloop = [&](auto SW) {
      if constexpr (W3) {
        for (int kt = 0; kt < nk - 2; ++kt) ktile(SW, BoolTag<false>{}, kt, IntTag<0>{});
        ktile(SW, BoolTag<false>{}, nk - 2, IntTag<1>{});
        ktile(SW, BoolTag<true>{}, nk - 1, IntTag<2>{});
      } else {
        for (int kt = 0; kt < nk - 1; ++kt) ktile(SW, BoolTag<false>{}, kt, IntTag<0>{});
        ktile(SW, BoolTag<true>{}, nk - 1, IntTag<0>{});
      }
    };
    UD_STAMP(1);
    if constexpr (EPI == UD_EPI_QKV) {
      if (swap) kloop(BoolTag<true>{});
      else kloop(BoolTag<false>{});
    } else {
      kloop(BoolTag<true>{});
    }
    UD_STAMP(2);
    if constexpr (SPK) {
      // ---- join of the two K halves (see the template comment; same protocol as gemm_body's SPLIT)
      constexpr int NQ = TMC * 4;
      const int half = (int)(blockIdx.x & 1);
      f32x4* mine = (f32x4*)p.splitk_ws + ((size_t)(t * 2 + half) * 8 + wv) * (NQ * 64) + lane;
      const f32x4* other = (const f32x4*)p.splitk_ws + ((size_t)(t * 2 + (half ^ 1)) * 8 + wv) * (NQ * 64) + lane;
#pragma unroll
      for (int i = 0; i < TMC; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 v = acc[i][j];
          asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 2" ::"v"(mine + (i * 4 + j) * 64), "v"(v) : "memory");
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();            // every wave's partial is out (and nobody reads operand tiles any more)
      unsigned* tk = (unsigned*)(smem + RING_BYTES);
      if (tid == 0) *tk = atomicAdd((unsigned*)p.splitk_cnt + t, 1u);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const unsigned ticket = *tk;
      if ((ticket & 1u) == 0) return;          // first of the pair: the partner finishes the tile (tickets: parity, never reset)
#pragma unroll
      for (int c = 0; c < NQ / 8; ++c) {
        f32x4 o[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(o[q]) : "v"(other + (c * 8 + q) * 64) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7])::"memory");
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[(c * 8 + q) >> 2][(c * 8 + q) & 3] += o[q];     // a + b == b + a: independent of which half came last
      }
    }
    const float* const lnst = LNC ? lds_rstd + (tl & 1) * 256 + wm * (BMC / 2) : nullptr;   // rstd of the wave's rows

    // =================================== epilogue ===================================
    // Accumulator layout (SWAP): lane owns row mbase + 16 i + (lane & 15), columns nbase + 16 j + 4 q .. + 3, q = lane >> 4.
    // v_permlane16_swap of the packed dwords of tiles j / j+1 leaves lane q with 8 consecutive columns starting at
    // nbase + 16 (j + (q & 1)) + 8 (q >> 1): 16-byte stores, 64-byte row segments.  V^T tiles (!SWAP) own 4 consecutive rows
    // (tokens) per lane; the same exchange between row tiles i / i+1 gives 8 consecutive tokens starting at
    // (v_permlane32_swap between row tiles i / i+1 packs them in the V^T block order, see below).
    const float* biasp = p.bias;
    if constexpr (GRP) {
      if (p.bias) biasp = p.bias + (long long)(m0 / p.grp_rows) * p.gBias;
    }
    const bool full = (m0 + BMC <= p.M) && (n0 + 256 <= p.N) && p.rows_in == 0 && p.add == nullptr && p.bias != nullptr;
    bool fast = false;
    bool ticket_taken = false;                   // row_stats_final: the straight-line fp32 epilogue takes the row tile's ticket early
    unsigned ticket_val = 0;
    int eln = lane;
    asm volatile("" : "+v"(eln));              // epilogue addresses are derived here, per tile: nothing of them lives across the K loop
    const int frow = eln & 15, fq = eln >> 4;
    if constexpr (EPI == UD_EPI_F16 || EPI == UD_EPI_QKV) {
      if (swap) {
        fast = full && (p.ldc & 7) == 0;
        if (fast) {
          const int act = p.act;
          f32x4 bv[4], ws[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) bv[j] = *(const f32x4*)(biasp + nbase + j * 16 + 4 * fq);
          if constexpr (LNC) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ws[j] = *(const f32x4*)(p.wsum + nbase + j * 16 + 4 * fq);
          }
          half_t* o = (half_t*)p.out + (size_t)(mbase + frow) * p.ldc + nbase + 16 * (fq & 1) + 8 * (fq >> 1);
          auto body = [&](auto ACT) {
            constexpr int AC = decltype(ACT)::value;
#pragma unroll
            for (int i = 0; i < TMC; ++i) {
              unsigned w[4][2];
              float rs = 0.f, nmr = 0.f;
              if constexpr (LNC) {
                rs = lnst[i * 16 + frow];
                nmr = lnst[UD_LNC_PLANE + i * 16 + frow];
              }
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                f32x4 v;
                if constexpr (LNC) v = ud_ln_apply4(acc[i][j], rs, nmr, ws[j], bv[j]);
                else v = acc[i][j] + bv[j];
                if constexpr (AC == UD_ACT_GELU) {
                  const f32x2 g0 = ud_gelu_erf2((f32x2){v[0], v[1]}), g1 = ud_gelu_erf2((f32x2){v[2], v[3]});
                  w[j][0] = ud_pack2(g0[0], g0[1]);
                  w[j][1] = ud_pack2(g1[0], g1[1]);
                } else {
                  w[j][0] = ud_pack2(ud_act_t<AC>(v[0]), ud_act_t<AC>(v[1]));
                  w[j][1] = ud_pack2(ud_act_t<AC>(v[2]), ud_act_t<AC>(v[3]));
                }
              }
#pragma unroll
              for (int jp = 0; jp < 2; ++jp) {
                ud_pair16(w[2 * jp][0], w[2 * jp + 1][0]);
                ud_pair16(w[2 * jp][1], w[2 * jp + 1][1]);
                u32x4 s;
                s[0] = w[2 * jp][0]; s[1] = w[2 * jp][1]; s[2] = w[2 * jp + 1][0]; s[3] = w[2 * jp + 1][1];
                *(u32x4*)(o + (size_t)(i * 16) * p.ldc + jp * 32) = s;
              }
            }
          };
          if (act == UD_ACT_GELU) body(IntTag<UD_ACT_GELU>{});
          else if (act == UD_ACT_LRELU) body(IntTag<UD_ACT_LRELU>{});
          else body(IntTag<UD_ACT_NONE>{});
        }
      } else {
        if constexpr (EPI == UD_EPI_QKV) {
          fast = full && (p.tok_per_img & 15) == 0 && (p.kv_ld & 7) == 0;
          if (fast) {
            half_t* vt = (half_t*)p.out2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int n = nbase + j * 16 + frow;
              const float bvn = biasp[n];
              float wsn = 0.f;
              if constexpr (LNC) wsn = p.wsum[n];
              const int nv = n - p.vsplit;
              half_t* vrow = vt + (size_t)((nv >> 6) * 64 + (nv & 63)) * p.kv_ld;
#pragma unroll
              for (int ip = 0; ip < MHC; ++ip) {
                unsigned w0[2], w1[2];
                if constexpr (LNC) {                               // the lane's tokens: rows 16 (2 ip) + 4 fq + r and 16 more
                  const f32x4 r0 = *(const f32x4*)(lnst + (2 * ip) * 16 + 4 * fq), q0 = *(const f32x4*)(lnst + UD_LNC_PLANE + (2 * ip) * 16 + 4 * fq);
                  const f32x4 r1 = *(const f32x4*)(lnst + (2 * ip) * 16 + 4 * fq + 16), q1 = *(const f32x4*)(lnst + UD_LNC_PLANE + (2 * ip) * 16 + 4 * fq + 16);
                  w0[0] = ud_pack2(ud_ln_apply(acc[2 * ip][j][0], r0[0], q0[0], wsn, bvn), ud_ln_apply(acc[2 * ip][j][1], r0[1], q0[1], wsn, bvn));
                  w0[1] = ud_pack2(ud_ln_apply(acc[2 * ip][j][2], r0[2], q0[2], wsn, bvn), ud_ln_apply(acc[2 * ip][j][3], r0[3], q0[3], wsn, bvn));
                  w1[0] = ud_pack2(ud_ln_apply(acc[2 * ip + 1][j][0], r1[0], q1[0], wsn, bvn), ud_ln_apply(acc[2 * ip + 1][j][1], r1[1], q1[1], wsn, bvn));
                  w1[1] = ud_pack2(ud_ln_apply(acc[2 * ip + 1][j][2], r1[2], q1[2], wsn, bvn), ud_ln_apply(acc[2 * ip + 1][j][3], r1[3], q1[3], wsn, bvn));
                } else {
                w0[0] = ud_pack2(acc[2 * ip][j][0] + bvn, acc[2 * ip][j][1] + bvn);
                w0[1] = ud_pack2(acc[2 * ip][j][2] + bvn, acc[2 * ip][j][3] + bvn);
                w1[0] = ud_pack2(acc[2 * ip + 1][j][0] + bvn, acc[2 * ip + 1][j][1] + bvn);
                w1[1] = ud_pack2(acc[2 * ip + 1][j][2] + bvn, acc[2 * ip + 1][j][3] + bvn);
                }
                // half-wave exchange between row tiles 2ip / 2ip+1: lane q ends up with tokens {4 (q & 1) .. +3, 8 + 4 (q & 1) .. +3}
                // of row tile 2ip + (q >> 1), i.e. one 16-byte chunk of the [0, 2, 1, 3] block order of that 16-token group
                ud_pair32(w0[0], w1[0]);
                ud_pair32(w0[1], w1[1]);
                const int mb = mbase + 16 * (2 * ip + (fq >> 1));
                const int img = mb / p.tok_per_img;
                const int tk = mb - img * p.tok_per_img + 8 * (fq & 1);
                u32x4 s;
                s[0] = w0[0]; s[1] = w0[1]; s[2] = w1[0]; s[3] = w1[1];
                *(u32x4*)(vrow + (size_t)img * p.heads_v * 64 * p.kv_ld + tk) = s;
              }
            }
          }
        }
      }
    } else if constexpr (EPI == UD_EPI_F32) {
      fast = full && (p.ldc & 3) == 0 && (p.out2 == nullptr || (p.ldc2 & 7) == 0) && p.act == UD_ACT_NONE;   // activated fp32 outputs: generic path
      if (fast) {
        f32x4 bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = *(const f32x4*)(biasp + nbase + j * 16 + 4 * fq);
        float* o = (float*)p.out + (size_t)(mbase + frow) * p.ldc + nbase + 4 * fq;
        float* omax = p.max_out ? p.max_out + (size_t)(mbase + frow) * p.ldc + nbase + 4 * fq : nullptr;
        half_t* o2 = p.out2 ? (half_t*)p.out2 + (size_t)(mbase + frow) * p.ldc2 + nbase + 16 * (fq & 1) + 8 * (fq >> 1) : nullptr;
        const bool wr32 = p.accumulate != 2;   // accumulate == 2: the fp32 stream dies here, only the fp16 copy is consumed
        const bool lre = p.act2 == UD_ACT_LRELU;
        if constexpr (!BAL) {
          if (p.row_stats_final) {
            // in-kernel reduction of the row statistics: the partial sums go out FIRST, then the ticket is taken, and only then the big
            // stores of the row values are issued -- the drain before the ticket covers 16 bytes per row instead of the whole tile, and
            // the ticket's round trip and the other column tiles' arrival hide under the stores (the finalizer below saw ~5 us per
            // launch with the ticket after the stores).  Same values, same order as the store loop below: v = acc + bias.
#pragma unroll
            for (int i = 0; i < TMC; ++i) {
              float rs1 = 0.f, rs2 = 0.f;
#pragma unroll
              for (int j = 0; j < 4; ++j) ud_row_stats_acc(acc[i][j] + bv[j], rs1, rs2);
              ud_row_stats_store(p, rs1, rs2, mbase + i * 16 + frow, nbase, eln, true);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (tid == 0) ticket_val = atomicInc(p.row_stats_ticket + m0 / BMC, (unsigned)tiles_n - 1u);
            ticket_taken = true;
          }
        }
#pragma unroll
        for (int i = 0; i < TMC; ++i) {
          unsigned w[4][2];
          float rs1 = 0.f, rs2 = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 v = acc[i][j] + bv[j];
            if (wr32) *(f32x4*)(o + (size_t)(i * 16) * p.ldc + j * 16) = v;
            if (omax) {
              float* mp = omax + (size_t)(i * 16) * p.ldc + j * 16;
              f32x4 mv = v;
              if (!p.max_init) {
                const f32x4 mo = *(const f32x4*)mp;
#pragma unroll
                for (int r = 0; r < 4; ++r) mv[r] = fmaxf(mo[r], v[r]);
              }
              *(f32x4*)mp = mv;
            }
            ud_row_stats_acc(v, rs1, rs2);
            if (o2) {
              f32x4 a = v;
              if (lre) {
#pragma unroll
                for (int r = 0; r < 4; ++r) a[r] = ud_lrelu(v[r]);
              }
              w[j][0] = ud_pack2(a[0], a[1]);
              w[j][1] = ud_pack2(a[2], a[3]);
            }
          }
          if (o2) {
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
              ud_pair16(w[2 * jp][0], w[2 * jp + 1][0]);
              ud_pair16(w[2 * jp][1], w[2 * jp + 1][1]);
              u32x4 s;
              s[0] = w[2 * jp][0]; s[1] = w[2 * jp][1]; s[2] = w[2 * jp + 1][0]; s[3] = w[2 * jp + 1][1];
              *(u32x4*)(o2 + (size_t)(i * 16) * p.ldc2 + jp * 32) = s;
            }
          }
          if (p.row_stats_out && !ticket_taken) ud_row_stats_store(p, rs1, rs2, mbase + i * 16 + frow, nbase, eln, true);
        }
      }
    }
    if constexpr (EPI == UD_EPI_D2S) {
      // ConvTranspose(k = s) depth-to-space accumulate, straight-line: the wave's 64 columns lie inside ONE (a, c) sub-pixel block
      // when Co % 64 == 0, so the destination pixel depends on the row only (one (img, y, x) decode per row tile, not per element);
      // fp32 read-modify-write in 16-byte pieces, fp16 copy paired to 16-byte stores.
      fast = (m0 + BMC <= p.M) && (n0 + 256 <= p.N) && (p.d2s_Co & 63) == 0 && (p.ldc & 3) == 0 && (p.out2 == nullptr || (p.ldc2 & 7) == 0);
      if (fast) {
        const int k = p.d2s_k;
        const int ac = nbase / p.d2s_Co;
        const int o0 = nbase - ac * p.d2s_Co;
        const int sa = ac / k, sc = ac - sa * k;
        const int Wout = p.d2s_Win * k;
        f32x4 bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = *(const f32x4*)(biasp + o0 + j * 16 + 4 * fq);
        const bool lre = p.act2 == UD_ACT_LRELU;
        // Round 6: a two-stage software pipeline over the wave's row tiles.  This epilogue IS the kernel for the transposed convolutions (K = 512: 8
        // K-tiles per tile behind 0.5 MB of fp32 read-modify-write per tile), and the old values of row tile i + 1 -- 16 gathered 16-byte loads per
        // lane with UdGemm.up_src -- used to be requested only after the stores of row tile i (the compiler cannot move a load above a store that
        // may alias it): one memory round trip per row tile, 6-8 in a row per output tile.  Now the requests of i + 1 are in flight under the
        // arithmetic and the stores of i.  Padding rows read pixel 0 of their image (valid memory) and store nothing: no branch around the loads.
        auto pixel_of = [&](int i, int& img, int& Y, int& X, bool& ok) -> long long {
          const int m = mbase + i * 16 + frow;
          img = m / p.d2s_rows_in_img;
          int pp = m - img * p.d2s_rows_in_img;
          ok = pp < p.d2s_Hin * p.d2s_Win;                                 // rows past the image are padding tokens
          pp = ok ? pp : 0;
          const int y = pp / p.d2s_Win, x = pp - y * p.d2s_Win;
          Y = y * k + sa; X = x * k + sc;
          return (long long)img * p.d2s_out_img_pix + (long long)Y * Wout + X;
        };
        auto fetch_old = [&](int i, f32x4 (&v)[4]) {
          int img, Y, X; bool ok;
          const long long pix = pixel_of(i, img, Y, X, ok);
          if (p.up_src) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ud_up2_fetch(p, img, Y, X, o0 + j * 16 + 4 * fq);
          } else {
            const float* src = (const float*)p.out + pix * p.ldc + o0 + 4 * fq;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = *(const f32x4*)(src + j * 16);
          }
        };
        f32x4 nxt[4];
        fetch_old(0, nxt);
#pragma unroll
        for (int i = 0; i < TMC; ++i) {
          int img, Y, X; bool ok;
          const long long pix = pixel_of(i, img, Y, X, ok);
          float* dst = (float*)p.out + pix * p.ldc + o0 + 4 * fq;
          f32x4 v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = nxt[j];
          if (i + 1 < TMC) fetch_old(i + 1, nxt);
          unsigned w[4][2];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] += acc[i][j] + bv[j];
            if (ok) *(f32x4*)(dst + j * 16) = v[j];
            f32x4 a = v[j];
            if (lre) {
#pragma unroll
              for (int r = 0; r < 4; ++r) a[r] = ud_lrelu(v[j][r]);
            }
            w[j][0] = ud_pack2(a[0], a[1]);
            w[j][1] = ud_pack2(a[2], a[3]);
          }
          if (p.out2) {
            // after the exchange lane q holds columns 16 (jp*2 + (q & 1)) + 8 (q >> 1) .. + 7 of ITS OWN row (same row, same pixel)
            half_t* d2 = (half_t*)p.out2 + pix * p.ldc2 + o0 + 16 * (fq & 1) + 8 * (fq >> 1);
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
              ud_pair16(w[2 * jp][0], w[2 * jp + 1][0]);
              ud_pair16(w[2 * jp][1], w[2 * jp + 1][1]);
              u32x4 sv;
              sv[0] = w[2 * jp][0]; sv[1] = w[2 * jp][1]; sv[2] = w[2 * jp + 1][0]; sv[3] = w[2 * jp + 1][1];
              if (ok) *(u32x4*)(d2 + jp * 32) = sv;
            }
          }
        }
      }
    }
    if (!fast) {
      // generic path: edge tiles, row remaps, `add` operands, depth-to-space (8-byte fp16 stores, per-element bounds checks)
      constexpr bool PRE = ACC_EPI;            // the residual is already inside the accumulators (in-loop or preloaded)
      if constexpr (EPI == UD_EPI_QKV) {
        if (swap) gemm_epilogue<TMC, 4, EPI, true, PRE>(p, acc, mbase, nbase, eln, biasp, (char*)p.out, (char*)p.out2, nullptr, 0.f, 0.f, nullptr, lnst);
        else gemm_epilogue<TMC, 4, EPI, false, PRE>(p, acc, mbase, nbase, eln, biasp, (char*)p.out, (char*)p.out2, nullptr, 0.f, 0.f, nullptr, lnst);
      } else {
        gemm_epilogue<TMC, 4, EPI, true, PRE>(p, acc, mbase, nbase, eln, biasp, (char*)p.out, (char*)p.out2, nullptr, 0.f, 0.f, nullptr, lnst);
      }
    }
    if constexpr (EPI == UD_EPI_F32 && !BAL) {
      if (p.row_stats_final) {
        // ---- LayerNorm statistics of the rows of this row tile: the LAST of the tiles_n workgroups to get here reduces the partial sums
        // of all column tiles (ascending slab order: one summation order per row) -- no separate reduction launch (~7.5 us each, 47
        // per step).  A ticket counts the arrivals of one row tile and wraps to 0 with the last one (atomicInc with bound tiles_n - 1):
        // every completed launch leaves the set at zero, whatever tiles_n is and however often the plan is replayed.
        unsigned* flag = (unsigned*)(smem + RING_BYTES);
        if (ticket_taken) {                       // straight-line epilogue: partial sums and ticket went out ahead of the row stores
          if (tid == 0) *flag = ticket_val;
        } else {                                  // edge tiles: after the stores
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          if (tid == 0) *flag = atomicInc(p.row_stats_ticket + m0 / BMC, (unsigned)tiles_n - 1u);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned arrived = *flag;
        if (arrived == (unsigned)tiles_n - 1u) {
          const int slabs = p.N >> 6;
          if (tid < BMC && m0 + tid < p.M) {
            const float* src = p.row_stats_out + (size_t)(m0 + tid) * slabs * 2;
            f32x4 q[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              q[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
              if (2 * k < slabs) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(q[k]) : "v"(src + 4 * k) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7])::"memory");
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              s1 += q[k][0]; s2 += q[k][1];
              s1 += q[k][2]; s2 += q[k][3];
            }
            const float inv = 1.0f / (float)p.ln_D;
            const float mean = s1 * inv;
            const float var = fmaxf(__builtin_fmaf(-mean, mean, s2 * inv), 0.0f);
            f32x2 o;
            o[0] = rsqrtf(var + p.ln_eps);
            o[1] = -mean * o[0];
            *(f32x2*)(p.row_stats_final + 2 * (size_t)(m0 + tid)) = o;
          }
        }
      }
    }
    UD_STAMP(3);
#ifdef UD_TRACE_DRAIN
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    UD_STAMP(4);
#endif
    };
    if constexpr (BAL) {
      if (mhc == 4) tile_body(IntTag<4>{});
      else if (mhc == 3) tile_body(IntTag<3>{});
      else tile_body(IntTag<2>{});
    } else {
      tile_body(IntTag<MH>{});
    }
    UD_STAMP(5);
    m0 = m0n;
    n0 = n0n;
    mhc = mhn;
    ++tl;
  }
}

// K split of the 192-row tile list (gemm256_kernel SPK): 2 * tiles workgroups; scratch per (tile, half) = 192 x 256 fp32
constexpr size_t SPK_SLOT_BYTES = (size_t)192 * 256 * 4;
template <int EPI, int AMODE>
int launch256sk(const UdGemm& d, hipStream_t s) {
  const int tiles = ((d.N + 255) >> 8) * ((d.M + 191) / 192);
  const int lds = 2 * BigCfg<3>::STAGE + 64;
  static bool attr_set[UD_MAX_DEVICES];
  if (!ud_attr_once(attr_set)) {
    if (hipFuncSetAttribute((const void*)gemm256_kernel<3, EPI, AMODE, false, false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      ud_set_error("ud_gemm_f16: cannot reserve the LDS ring of the large-tile kernel (K split)");
      return UD_ERR_LAUNCH;
    }
  }
  hipLaunchKernelGGL((gemm256_kernel<3, EPI, AMODE, false, false, false, false, true>), dim3(2 * tiles), dim3(512), lds, s, d);
  UD_CHECK_LAUNCH("ud_gemm_f16 (large tile, K split) launch");
  return UD_OK;
}
// eligibility of the K split of the large-tile list: a one-round list on at most half of the CUs, long K, plain fp16 / fp32 epilogues,
// caller-provided scratch of 2 * tiles * SPK_SLOT_BYTES
inline bool big_split_ok(const UdGemm& d) {
#ifdef UD_AB_PREV
  return false;
#endif
  if (d.amode != UD_A_DENSE && d.amode != UD_A_CONV3_ZERO) return false;
  if (d.epi != UD_EPI_F16 && d.epi != UD_EPI_F32) return false;
  if (d.groups > 1 || d.row_stats_in || d.row_stats_out || d.max_out || d.a_wrap || d.w_wrap || (d.K & 127) || d.K < 2048) return false;
  const int tiles = ((d.N + 255) >> 8) * ((d.M + 191) / 192);
  if (tiles > 128 || tiles < 32 || d.M < 1024 || d.N < 192) return false;
  if (!d.splitk_ws || !d.splitk_cnt || (size_t)d.splitk_ws_bytes < 2 * (size_t)tiles * SPK_SLOT_BYTES) return false;
  return true;
}

// W3 form of the 192-row tile list (3-deep weight ring, see the kernel): dense problems with at least two K-tiles.
inline bool w3_enabled() { return true; }

template <int MH, int EPI, int AMODE, bool LNC = false, bool GRP = false>
int launch256(const UdGemm& d, hipStream_t s) {
  constexpr int BM = BigCfg<MH>::BM;
  const int tiles = ((d.N + 255) >> 8) * ((d.M + BM - 1) / BM);
  if constexpr (MH == 3 && AMODE == UD_A_DENSE && !GRP) {
    if (d.K >= 128 && d.tile_hint != 9 && w3_enabled()) {
      const int lds3 = 2 * BigCfg<MH>::A_BYTES + 3 * 32768 + (LNC ? LNC_LDS : 0) + (EPI == UD_EPI_F32 ? 64 : 0);
      static bool attr3_set[UD_MAX_DEVICES];
      if (!ud_attr_once(attr3_set)) {
        if (hipFuncSetAttribute((const void*)gemm256_kernel<MH, EPI, AMODE, false, LNC, GRP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds3) != hipSuccess) {
          ud_set_error("ud_gemm_f16: cannot reserve the LDS rings of the large-tile kernel (3-deep weight ring)");
          return UD_ERR_LAUNCH;
        }
      }
      hipLaunchKernelGGL((gemm256_kernel<MH, EPI, AMODE, false, LNC, GRP, true>), dim3(tiles < 256 ? tiles : 256), dim3(512), lds3, s, d);
      UD_CHECK_LAUNCH("ud_gemm_f16 (large tile, 3-deep weight ring) launch");
      return UD_OK;
    }
  }
  const int lds = 2 * BigCfg<MH>::STAGE + (LNC ? LNC_LDS : 0) + (EPI == UD_EPI_F32 ? 64 : 0);
  static bool attr_set[UD_MAX_DEVICES];
  if (!ud_attr_once(attr_set)) {
    if (hipFuncSetAttribute((const void*)gemm256_kernel<MH, EPI, AMODE, false, LNC, GRP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      ud_set_error("ud_gemm_f16: cannot reserve the LDS ring of the large-tile kernel");
      return UD_ERR_LAUNCH;
    }
  }
  hipLaunchKernelGGL((gemm256_kernel<MH, EPI, AMODE, false, LNC, GRP>), dim3(tiles < 256 ? tiles : 256), dim3(512), lds, s, d);
  UD_CHECK_LAUNCH("ud_gemm_f16 (large tile) launch");
  return UD_OK;
}

// Row-balanced schedule (gemm256_kernel<4, EPI, DENSE, true>): worth it when the classic tile list leaves the last round partly empty.
// Needs at least two 64-row units per workgroup (tile heights 128 / 192 / 256 only).
inline int bal_cpc(const UdGemm& d) {
  const int tiles_n = (d.N + 255) >> 8;
  if (tiles_n > 128 || (d.N & 255)) return 0;
  const int cpc = 256 / tiles_n;
  const int U = (d.M + 63) >> 6;
  return (U / cpc >= 2) ? cpc : 0;
}
// predicted time (us) of the balanced schedule, same units as pick_tiles: rows of the longest span at the 256-row tile's per-row
// rate, plus the fixed cost and ~2.5 us of un-hidden prologue / epilogue per extra tile
inline double bal_time(const UdGemm& d) {
  const int cpc = bal_cpc(d);
  if (!cpc) return 1e30;
  const int U = (d.M + 63) >> 6;
  const int un = (U + cpc - 1) / cpc;
  const int k = (un + 3) >> 2;
  return 8.0 + (un * (30.0 / 4.0) + (k - 1) * 2.5) * ((double)d.K / 1024.0);
}

template <int EPI, bool LNC = false, int AMODE = UD_A_DENSE>
int launch256bal(const UdGemm& d, hipStream_t s) {
  const int tiles_n = (d.N + 255) >> 8;
  const int cpc = bal_cpc(d);
  const int lds = 2 * BigCfg<4>::STAGE + (LNC ? LNC_LDS : 0);
  static bool attr_set[UD_MAX_DEVICES];
  if (!ud_attr_once(attr_set)) {
    if (hipFuncSetAttribute((const void*)gemm256_kernel<4, EPI, AMODE, true, LNC>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      ud_set_error("ud_gemm_f16: cannot reserve the LDS ring of the large-tile kernel");
      return UD_ERR_LAUNCH;
    }
  }
  hipLaunchKernelGGL((gemm256_kernel<4, EPI, AMODE, true, LNC>), dim3(cpc * tiles_n), dim3(512), lds, s, d);
  UD_CHECK_LAUNCH("ud_gemm_f16 (large tile, row-balanced) launch");
  return UD_OK;
}

// Tile-shape choice for dense GEMMs.  Cost model fitted on MI355X (tools/bench_gemm*.py): per-launch fixed cost + rounds x
// K-loop time per round.  The large-tile kernel (one workgroup per CU) loses to wave quantisation when the tile count is
// small or just above a multiple of 256 CUs; the 128x128 kernel runs two workgroups per CU, so its rounds quantise in halves.
// Returns 0 (128x128), 3 (192x256) or 4 (256x256).
inline int pick_tiles(const UdGemm& d) {
  if (d.amode == UD_A_CONV3_REFLECT || d.groups > 1 || d.M < 1024 || d.N < 192 || (d.K & 63)) return 0;
  if (d.amode == UD_A_CONV3_ZERO && d.epi != UD_EPI_F16 && d.epi != UD_EPI_F32) return 0;
  {  // the large-tile loader addresses its operands with 32-bit byte offsets inside 2 GB buffer descriptors
    const double a_bytes = d.amode == UD_A_DENSE ? 2.0 * d.M * d.lda : 2.0 * ((double)(d.M / d.rows_img) + 1.0) * (double)d.img_stride;
    if (a_bytes >= 2147483648.0 || 2.0 * d.N * d.ldw >= 2147483648.0) return 0;
  }
  if (d.epi == UD_EPI_QKV && (d.vsplit & 255)) return 0;
  if (d.tile_hint == 1 || (d.tile_hint >= 5 && d.tile_hint != 8 && d.tile_hint != 9 && d.tile_hint != 10 && (d.tile_hint < 11 || d.tile_hint > 14))) return 0;
  if (d.tile_hint >= 11 && d.tile_hint <= 14) return 3;         // ping-pong / duo forms refused (gemm_pp.hip): the 192-row list
  if (d.tile_hint == 2) return 4;
  if (d.tile_hint == 10) return big_split_ok(d) ? 10 : 0;       // 10: 192-row tile list with the two-way K split (when eligible)
  if (d.tile_hint == 3 || d.tile_hint == 9) return 3;      // 9: 192-row tile list with the 2-deep weight ring (A/B and tests of the 3-deep form)
  const bool bal_ok = (d.amode == UD_A_DENSE || (d.amode == UD_A_CONV3_ZERO && d.epi != UD_EPI_QKV)) &&
                      (d.epi == UD_EPI_F16 || d.epi == UD_EPI_F32 || d.epi == UD_EPI_QKV) && bal_cpc(d) > 0;
  if (d.tile_hint == 8 && bal_ok) return 8;
  const double kk = (double)d.K / 1024.0;
  const double tn = (double)((d.N + 255) / 256);
  const double t256 = 8.0 + ceil(tn * ((d.M + 255) / 256) / 256.0) * 30.0 * kk;
  const double t192 = 8.0 + ceil(tn * ((d.M + 191) / 192) / 256.0) * 23.5 * kk;
  // a span of one tile per workgroup cannot finish earlier than the tile list does (proj / fc2 at bs = 8: 192 rows either way, and the
  // MH = 4 instantiation measured 4 % slower there): the balanced schedule competes only when spans hold two or more tiles
  const double tbal = (bal_ok && ((d.M + 63) / 64 + bal_cpc(d) - 1) / bal_cpc(d) > 4) ? bal_time(d) : 1e30;
  const double small_tiles = (double)((d.N + 127) / 128) * ((d.M + 127) / 128);
  const double t_small = 6.0 + ceil(small_tiles / 256.0) * 0.5 * 21.5 * kk;
  double t_big = t256 <= t192 ? t256 : t192;
  int which = t256 <= t192 ? 4 : 3;
  if (tbal < 0.985 * t_big) { t_big = tbal; which = 8; }      // measured: fc1 94.5 vs 98.5 us (model 95.5 / 98), qkv 82 vs 74.5 (80.5 / 78.5)
  if (big_split_ok(d)) {                                       // half the K loop on twice the workgroups + ~6 us for the exchange of the partial tiles
    const double tsplit = 8.0 + 0.5 * 23.5 * kk + 6.0;
    if (tsplit < t_big && tsplit < 0.93 * t_small) return 10;
  }
  if (t_big >= 0.93 * t_small) return 0;
  return which;
}

// tiles one workgroup of the large-tile kernel walks for this problem (the folded-LayerNorm consumer keeps one statistics table per tile)
inline int big_tiles_per_wg(const UdGemm& d, int which) {
  if (which == 8) {
    const int cpc = bal_cpc(d);
    const int U = (d.M + 63) >> 6;
    return (((U + cpc - 1) / cpc) + 3) >> 2;
  }
  if (which == 10) return 1;
  const int bm = which == 3 ? 192 : 256;
  const int tiles = ((d.N + 255) >> 8) * ((d.M + bm - 1) / bm);
  return (tiles + 255) / 256;
}

template <int EPI, int AMODE = UD_A_DENSE>
int launch_big(const UdGemm& d, hipStream_t s, int which) {
  if constexpr (AMODE == UD_A_DENSE && (EPI == UD_EPI_F16 || EPI == UD_EPI_QKV)) {
    if (d.row_stats_in) {                 // consumer of a folded LayerNorm: separate instantiations (statistics table in LDS)
      if (which == 8) return launch256bal<EPI, true>(d, s);
      return which == 3 ? launch256<3, EPI, AMODE, true>(d, s) : launch256<4, EPI, AMODE, true>(d, s);
    }
  }
  if constexpr (AMODE == UD_A_DENSE && (EPI == UD_EPI_F16 || EPI == UD_EPI_F32 || EPI == UD_EPI_QKV)) {
    if (which == 8 && !d.row_stats_final) return launch256bal<EPI>(d, s);
  }
  if constexpr (AMODE == UD_A_CONV3_ZERO && (EPI == UD_EPI_F16 || EPI == UD_EPI_F32)) {
    if (which == 8 && !d.row_stats_final) return launch256bal<EPI, false, UD_A_CONV3_ZERO>(d, s);
  }
  if constexpr (EPI == UD_EPI_F16 || EPI == UD_EPI_F32) {
    if (which == 10) return launch256sk<EPI, AMODE>(d, s);
  }
  if (which == 8 || which == 10) which = 3;               // in-kernel statistics reduction: tickets are per row tile of ONE height (tile list only)
  return which == 3 ? launch256<3, EPI, AMODE>(d, s) : launch256<4, EPI, AMODE>(d, s);
}

// ================================================================================================================
// Halo-tile 3x3 convolution for narrow outputs (N <= 64: the two head convolutions, decoder.py:199-226).
// The implicit-GEMM kernels above re-gather every input pixel once per tap (9x) through L2 -> LDS; with N <= 64 there is so
// little MFMA work per gathered byte that they are bound by that traffic (measured 346 / 517 TFLOP/s).  Here a workgroup owns
// a 16 x 16 output tile: the 18 x 18 x 64-channel input halo is DMA'd into LDS once per 64-channel chunk and the 9 taps
// read it in place (shifted fragment addresses); only the per-tap [N][64] weight slab streams (double buffered).
// 4 waves, wave w = output rows 4w..4w+3 (one 16-pixel MFMA m-tile per row), v_mfma_f32_16x16x32_f16, swapped operands.
// ================================================================================================================
constexpr int HALO_BYTES = 41 * 1024;          // 328 pixel slots x 128 B (324 used)

// UPS: the convolution's input is the bilinear align_corners=True up-sampling of A [B, Hsrc, Wsrc, Cin] to (Himg, Wimg) (reference
// decoder.py:299-301,309-311 F.interpolate between to_*_lr and to_*_hr) and is NEVER materialised: every halo pixel is interpolated once
// from its 4 source pixels while the tile is staged (round 1 wrote the 518 x 518 x 64-channel map of both branches to HBM, 549 MB, and
// re-read it with halos, 697 MB: resize_ac_kernel 0.20 ms + this kernel 0.27 ms per step).
template <int NT, int EPI, bool REFLECT, bool UPS = false>
__global__ __launch_bounds__(256) void conv_tile_kernel(const UdGemm p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* wbuf = smem + HALO_BYTES;
  constexpr int WB = NT * 16 * 128;             // bytes per weight buffer
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_x = (p.Wimg + 15) >> 4;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int x0 = tx << 4, y0 = ty << 4;
  const int b = blockIdx.y, g = blockIdx.z;
  const half_t* in = (const half_t*)p.A + (long long)g * p.gA + (long long)b * p.img_stride + p.coff;
  const half_t* W = (const half_t*)p.W + (long long)g * p.gW;
  const float* bias = p.bias + (long long)g * p.gBias;
  const float ups_sy = UPS && p.Himg > 1 ? (float)(p.Hsrc - 1) / (float)(p.Himg - 1) : 0.f;
  const float ups_sx = UPS && p.Wimg > 1 ? (float)(p.Wsrc - 1) / (float)(p.Wimg - 1) : 0.f;

  f32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto halo_coord = [&](int i, int& hp, int& cpos, int& iy, int& ix, bool& ok) {
    const int idx = i * 64 + lane;
    hp = idx >> 3;
    cpos = idx & 7;
    const int hy = hp / 18, hx = hp - hy * 18;
    iy = y0 - 1 + hy;
    ix = x0 - 1 + hx;
    ok = hp < 324;
    if constexpr (REFLECT) {
      iy = iy < 0 ? -iy : (iy >= p.Himg ? 2 * p.Himg - 2 - iy : iy);
      ix = ix < 0 ? -ix : (ix >= p.Wimg ? 2 * p.Wimg - 2 - ix : ix);
      ok = ok && iy >= 0 && ix >= 0;         // tile overhang beyond the reflected border: unused outputs
    } else {
      ok = ok && (unsigned)iy < (unsigned)p.Himg && (unsigned)ix < (unsigned)p.Wimg;
    }
  };
  auto issue_halo = [&](int cc) {
    if constexpr (UPS) {
      // (1) the source pixels this halo needs form a small patch (<= 13 x 13 for the decoder's 4/7 scale): DMA it to LDS once;
      // (2) every (halo pixel, 8-channel chunk) is interpolated from 4 LDS reads with packed fp16 math (v_pk_fma_f16: the operands and the
      //     result are fp16 anyway) and stored where the plain loader's DMA would have put it.
      // Reading the 4 source pixels straight from global memory per item exposed an L2 round trip per item (1.05 ms per step).
      const int ylo = y0 == 0 ? 0 : y0 - 1, yhi = y0 + 16 < p.Himg ? y0 + 16 : p.Himg - 1;
      const int xlo = x0 == 0 ? 0 : x0 - 1, xhi = x0 + 16 < p.Wimg ? x0 + 16 : p.Wimg - 1;
      const int syA = (int)(ups_sy * (float)ylo), sxA = (int)(ups_sx * (float)xlo);
      int syB = (int)(ups_sy * (float)yhi) + 1, sxB = (int)(ups_sx * (float)xhi) + 1;
      syB = syB < p.Hsrc ? syB : p.Hsrc - 1;
      sxB = sxB < p.Wsrc ? sxB : p.Wsrc - 1;
      const int PR = syB - syA + 1, PC = sxB - sxA + 1;          // <= UPS_PATCH each (checked on the host)
      char* patch = wbuf + 2 * WB;
      const int npiece = (PR * PC * 8 + 63) >> 6;
      for (int i = wv; i < npiece; i += 4) {
        const int idx = i * 64 + lane;
        int pp = idx >> 3;
        pp = pp < PR * PC ? pp : PR * PC - 1;
        const int pr = pp / PC, pc = pp - pr * PC;
        ud_glds16(in + ((long long)(syA + pr) * p.Wsrc + sxA + pc) * p.cstride + cc * 64 + ((idx & 7) << 3), patch + i * 1024);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 11; ++j) {
        const int i = wv + 4 * j;
        if (i < 41) {
          int hp, cpos, iy, ix;
          bool ok;
          halo_coord(i, hp, cpos, iy, ix, ok);
          // a tile overhanging the bottom / right border reflects halo rows far ABOVE the patch (2 Himg - 2 - iy < ylo): those feed unused
          // outputs only, but their source pixels are not in the staged patch -- write zeros instead of reading LDS out of range
          ok = ok && iy >= ylo && iy <= yhi && ix >= xlo && ix <= xhi;
          half8 o8;
#pragma unroll
          for (int e = 0; e < 8; ++e) o8[e] = (half_t)0.f;
          if (ok) {
            const float fy = ups_sy * (float)iy, fx = ups_sx * (float)ix;
            const int sy0 = (int)fy, sx0 = (int)fx;
            const int sy1 = sy0 + (sy0 < p.Hsrc - 1), sx1 = sx0 + (sx0 < p.Wsrc - 1);
            const half_t ly = (half_t)(fy - (float)sy0), lx = (half_t)(fx - (float)sx0);
            const int cs = (cpos ^ ((hp >> 1) & 7)) << 4;
            const half8 h00 = *(const half8*)(patch + ((sy0 - syA) * PC + (sx0 - sxA)) * 128 + cs);
            const half8 h01 = *(const half8*)(patch + ((sy0 - syA) * PC + (sx1 - sxA)) * 128 + cs);
            const half8 h10 = *(const half8*)(patch + ((sy1 - syA) * PC + (sx0 - sxA)) * 128 + cs);
            const half8 h11 = *(const half8*)(patch + ((sy1 - syA) * PC + (sx1 - sxA)) * 128 + cs);
            const half8 top = h00 + lx * (h01 - h00);
            const half8 bot = h10 + lx * (h11 - h10);
            o8 = top + ly * (bot - top);
          }
          *(half8*)(halo + i * 1024 + lane * 16) = o8;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 11; ++j) {
        const int i = wv + 4 * j;                  // wave-instruction index (uniform)
        if (i < 41) {
          int hp, cpos, iy, ix;
          bool ok;
          halo_coord(i, hp, cpos, iy, ix, ok);
          const half_t* src = ok ? in + ((long long)iy * p.Wimg + ix) * p.cstride + cc * 64 + ((cpos ^ ((hp >> 1) & 7)) << 3)
                                 : (const half_t*)p.zeros;
          ud_glds16(src, halo + i * 1024);
        }
      }
    }
  };
  auto issue_w = [&](int tap, int cc, int buf) {
#pragma unroll
    for (int j = 0; j < NT / 2; ++j) {
      const int i = wv * (NT / 2) + j;
      const int idx = i * 64 + lane;
      const int n = idx >> 3, cpos = idx & 7;
      ud_glds16(W + (size_t)n * p.ldw + tap * p.Cin + cc * 64 + ((cpos ^ ((n >> 1) & 7)) << 3), wbuf + buf * WB + i * 1024);
    }
  };

  const int px = lane & 15, fq = lane >> 4;
  const int nchunks = p.Cin >> 6;
  for (int cc = 0; cc < nchunks; ++cc) {
    __syncthreads();                              // previous chunk fully consumed
    issue_halo(cc);
    issue_w(0, cc, 0);
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tap < 8) issue_w(tap + 1, cc, (tap + 1) & 1);
      const int t3 = (tap * 11) >> 5;
      const int dy = t3, dx = tap - t3 * 3;       // halo coordinates already include the -1 offset
      const char* wb = wbuf + (tap & 1) * WB;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        half8 af[4], bf[NT];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int hp = (wv * 4 + i + dy) * 18 + px + dx;
          af[i] = *(const half8*)(halo + hp * 128 + (((ks * 4 + fq) ^ ((hp >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int n = j * 16 + px;
          bf[j] = *(const half8*)(wb + n * 128 + (((ks * 4 + fq) ^ ((n >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: lane owns pixel (y0 + 4*wv + i, x0 + px), channels j*16 + 4*fq .. +3
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int y = y0 + wv * 4 + i, x = x0 + px;
    const bool ok = y < p.Himg && x < p.Wimg;
    const size_t row = ((size_t)b * p.Himg + y) * p.Wimg + x;
    if constexpr (EPI == UD_EPI_HEAD) {
      const float* w2 = p.w2 + (long long)g * p.gW2;
      float part = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int nb = j * 16 + 4 * fq;
        const f32x4 bv = *(const f32x4*)(bias + nb);
        const f32x4 wv2 = *(const f32x4*)(w2 + nb);
#pragma unroll
        for (int r = 0; r < 4; ++r) part += ud_lrelu(acc[i][j][r] + bv[r]) * wv2[r];
      }
      part += __shfl_xor(part, 16, 64);
      part += __shfl_xor(part, 32, 64);
      if (ok && fq == 0) {
        float yv = part + (g == 0 ? p.b2 : p.b2_g1);
        yv = fminf(fmaxf(yv, -8.0f), 8.0f);
        ((float*)p.out)[(long long)g * p.gOut + row] = __expf(yv + (g == 0 ? p.post_add : p.post_add_g1));
      }
    } else {
      if (ok) {
        const bool lre = p.act == UD_ACT_LRELU;
        half_t* o = (half_t*)p.out + (long long)g * p.gOut + row * p.ldc;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int nb = j * 16 + 4 * fq;
          const f32x4 bv = *(const f32x4*)(bias + nb);
          half4 h;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float t = acc[i][j][r] + bv[r];
            h[r] = (half_t)(lre ? ud_lrelu(t) : t);       // branch-free: act is NONE or LeakyReLU here (conv_tile_ok)
          }
          *(half4*)(o + nb) = h;
        }
      }
    }
  }
}


// ================================================================================================================
// Head convolution with its weights in REGISTERS (round 4): the ViT-L head (N = 32, Cin = 64, 3x3 reflect over the fused align_corners
// up-sampling, LeakyReLU + 1x1 + clip / exp epilogue) has 36 KB of weights per branch -- 9 taps x [32][64] fp16 -- and the kernel above
// streams them through LDS per tile and tap: 9 x (wait for the slab's DMA, barrier, 16 MFMAs per wave), ~12 us per 16 x 16 tile against
// ~1.2 us of MFMA work.  Here a PERSISTENT workgroup loads the branch's weights once into registers as MFMA fragments (36 x half8 per lane,
// 144 VGPRs; two workgroups per CU = two waves per SIMD, 256 registers each) and walks its tiles: source patch by LDS-DMA (issued one tile
// ahead, behind the barrier that frees the patch buffer), interpolation into the halo, then 144 MFMAs per wave with NO barrier and no DMA
// wait between taps, epilogue.  Two barriers per tile instead of eleven; LDS 63 KB (halo + patch).
// ================================================================================================================
__global__ __launch_bounds__(256, 2) void conv_head_regw_kernel(const UdGemm p, const int tiles_x, const int tiles_img, const int tiles_g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* patch = smem + HALO_BYTES;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = blockIdx.y;
  const int px = lane & 15, fq = lane >> 4;
  const half_t* W = (const half_t*)p.W + (long long)g * p.gW;
  const float* bias = p.bias + (long long)g * p.gBias;
  const float* w2 = p.w2 + (long long)g * p.gW2;
  const float ups_sy = p.Himg > 1 ? (float)(p.Hsrc - 1) / (float)(p.Himg - 1) : 0.f;
  const float ups_sx = p.Wimg > 1 ? (float)(p.Wsrc - 1) / (float)(p.Wimg - 1) : 0.f;

  // ---- the branch's weights as B fragments: lane (px, fq) holds W[n = 16 j + px][tap * 64 + (4 ks + fq) * 8 .. + 8]
  half8 bfr[9][2][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j) bfr[tap][ks][j] = *(const half8*)(W + (size_t)(j * 16 + px) * p.ldw + tap * 64 + (ks * 4 + fq) * 8);
  // tile geometry is workgroup-uniform: kept in SGPRs (readfirstlane), the register file belongs to the weight fragments
  struct Geo { int b, x0, y0, ylo, yhi, xlo, xhi, syA, sxA, PR, PC; };
  auto geo = [&](int t) {
    Geo q;
    q.b = __builtin_amdgcn_readfirstlane(t / tiles_img);
    const int r = t - q.b * tiles_img;
    const int ty = __builtin_amdgcn_readfirstlane(r / tiles_x), tx = r - ty * tiles_x;
    q.x0 = tx << 4; q.y0 = ty << 4;
    q.ylo = q.y0 == 0 ? 0 : q.y0 - 1; q.yhi = q.y0 + 16 < p.Himg ? q.y0 + 16 : p.Himg - 1;
    q.xlo = q.x0 == 0 ? 0 : q.x0 - 1; q.xhi = q.x0 + 16 < p.Wimg ? q.x0 + 16 : p.Wimg - 1;
    q.syA = (int)(ups_sy * (float)q.ylo); q.sxA = (int)(ups_sx * (float)q.xlo);
    int syB = (int)(ups_sy * (float)q.yhi) + 1, sxB = (int)(ups_sx * (float)q.xhi) + 1;
    syB = syB < p.Hsrc ? syB : p.Hsrc - 1;
    sxB = sxB < p.Wsrc ? sxB : p.Wsrc - 1;
    q.syA = __builtin_amdgcn_readfirstlane(q.syA); q.sxA = __builtin_amdgcn_readfirstlane(q.sxA);
    q.PR = __builtin_amdgcn_readfirstlane(syB - q.syA + 1); q.PC = __builtin_amdgcn_readfirstlane(sxB - q.sxA + 1);
    return q;
  };
  auto issue_patch = [&](const Geo& q) {
    const half_t* in = (const half_t*)p.A + (long long)g * p.gA + (long long)q.b * p.img_stride + p.coff;
    const int npiece = (q.PR * q.PC * 8 + 63) >> 6;
    for (int i = wv; i < npiece; i += 4) {
      const int idx = i * 64 + lane;
      int pp = idx >> 3;
      pp = pp < q.PR * q.PC ? pp : q.PR * q.PC - 1;
      const int pr = pp / q.PC, pc = pp - pr * q.PC;
      ud_glds16(in + ((long long)(q.syA + pr) * p.Wsrc + q.sxA + pc) * p.cstride + ((idx & 7) << 3), patch + i * 1024);
    }
  };

  int t = __builtin_amdgcn_readfirstlane(blockIdx.x);
  if (t >= tiles_g) return;
  Geo cur = geo(t);
  issue_patch(cur);
  for (; t < tiles_g; t += gridDim.x) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of the patch landed (and its stores of the previous tile left)
    __syncthreads();                                       // ... everybody's; and every wave is past the halo reads of the previous tile
    // ---- interpolate the 18 x 18 halo from the staged patch (same arithmetic as conv_tile_kernel<.., UPS>: same bits)
#pragma unroll 1
    for (int jj = 0; jj < 11; ++jj) {
      const int i = wv + 4 * jj;
      if (i < 41) {
        const int idx = i * 64 + lane;
        const int hp = idx >> 3, cpos = idx & 7;
        const int hy = hp / 18, hx = hp - hy * 18;
        int iy = cur.y0 - 1 + hy, ix = cur.x0 - 1 + hx;
        iy = iy < 0 ? -iy : (iy >= p.Himg ? 2 * p.Himg - 2 - iy : iy);
        ix = ix < 0 ? -ix : (ix >= p.Wimg ? 2 * p.Wimg - 2 - ix : ix);
        const bool ok = hp < 324 && iy >= cur.ylo && iy <= cur.yhi && ix >= cur.xlo && ix <= cur.xhi;
        half8 o8;
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] = (half_t)0.f;
        if (ok) {
          const float fy = ups_sy * (float)iy, fx = ups_sx * (float)ix;
          const int sy0 = (int)fy, sx0 = (int)fx;
          const int sy1 = sy0 + (sy0 < p.Hsrc - 1), sx1 = sx0 + (sx0 < p.Wsrc - 1);
          const half_t ly = (half_t)(fy - (float)sy0), lx = (half_t)(fx - (float)sx0);
          const int cs = (cpos ^ ((hp >> 1) & 7)) << 4;
          const half8 h00 = *(const half8*)(patch + ((sy0 - cur.syA) * cur.PC + (sx0 - cur.sxA)) * 128 + cs);
          const half8 h01 = *(const half8*)(patch + ((sy0 - cur.syA) * cur.PC + (sx1 - cur.sxA)) * 128 + cs);
          const half8 h10 = *(const half8*)(patch + ((sy1 - cur.syA) * cur.PC + (sx0 - cur.sxA)) * 128 + cs);
          const half8 h11 = *(const half8*)(patch + ((sy1 - cur.syA) * cur.PC + (sx1 - cur.sxA)) * 128 + cs);
          const half8 top = h00 + lx * (h01 - h00);
          const half8 bot = h10 + lx * (h11 - h10);
          o8 = top + ly * (bot - top);
        }
        *(half8*)(halo + i * 1024 + lane * 16) = o8;
      }
    }
    __syncthreads();                                       // halo complete; the patch buffer is free
    const int tn = t + (int)gridDim.x;
    Geo nxt = cur;
    if (tn < tiles_g) {
      nxt = geo(tn);
      issue_patch(nxt);                                    // lands under the MFMAs below
    }
    // ---- 9 taps x 2 k-steps x (4 rows x 2 column tiles) MFMAs, fragments of A straight from the halo, weights from registers
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        half8 af[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int hp = (wv * 4 + i + dy) * 18 + px + dx;
          af[i] = *(const half8*)(halo + hp * 128 + (((ks * 4 + fq) ^ ((hp >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bfr[tap][ks][j], af[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);                 // fragment reads stay one k-step ahead at most: 144 registers hold weights
      }
    }
    // ---- epilogue (UD_EPI_HEAD): lane owns pixel (y0 + 4 wv + i, x0 + px), channels 16 j + 4 fq .. + 3
    f32x4 bv[2], wv2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bv[j] = *(const f32x4*)(bias + j * 16 + 4 * fq);
      wv2[j] = *(const f32x4*)(w2 + j * 16 + 4 * fq);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = cur.y0 + wv * 4 + i, x = cur.x0 + px;
      const bool ok = y < p.Himg && x < p.Wimg;
      const size_t row = ((size_t)cur.b * p.Himg + y) * p.Wimg + x;
      float part = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) part += ud_lrelu(acc[i][j][r] + bv[j][r]) * wv2[j][r];
      part += __shfl_xor(part, 16, 64);
      part += __shfl_xor(part, 32, 64);
      if (ok && fq == 0) {
        float yv = part + (g == 0 ? p.b2 : p.b2_g1);
        yv = fminf(fmaxf(yv, -8.0f), 8.0f);
        ((float*)p.out)[(long long)g * p.gOut + row] = __expf(yv + (g == 0 ? p.post_add : p.post_add_g1));
      }
    }
    cur = nxt;
  }
}

constexpr int UPS_PATCH = 13;                  // source patch side of the fused up-sampling loader (halo of 18 at a scale <= 0.6, + 2)
template <int NT, int EPI, bool REFLECT, bool UPS = false>
int launch_conv_tile(const UdGemm& d, hipStream_t s) {
  const int lds = HALO_BYTES + 2 * NT * 16 * 128 + (UPS ? (UPS_PATCH * UPS_PATCH * 128 + 1023) / 1024 * 1024 : 0);
  if (UPS) {
    static bool attr_set[UD_MAX_DEVICES];
    if (!ud_attr_once(attr_set)) (void)hipFuncSetAttribute((const void*)conv_tile_kernel<NT, EPI, REFLECT, UPS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  const int B = d.M / d.rows_img;
  dim3 grid(((d.Wimg + 15) >> 4) * ((d.Himg + 15) >> 4), B, d.groups > 0 ? d.groups : 1);
  hipLaunchKernelGGL((conv_tile_kernel<NT, EPI, REFLECT, UPS>), grid, dim3(256), lds, s, d);
  UD_CHECK_LAUNCH("ud_gemm_f16 (halo-tile conv) launch");
  return UD_OK;
}

inline bool head_regw_enabled() { return true; }

// eligibility: dense images (rows_img == H*W), N in {32, 64}, Cin multiple of 64, weights laid out [N][tap*Cin + ci]
inline bool conv_tile_ok(const UdGemm& d) {
  return d.amode != UD_A_DENSE && d.a_wrap == 0 && (d.N == 32 || d.N == 64) && (d.Cin & 63) == 0 && d.rows_img == d.Himg * d.Wimg &&
         d.M % d.rows_img == 0 && d.bias != nullptr && d.tile_hint != 1 && d.Himg >= 2 && d.Wimg >= 2 &&
         (d.epi == UD_EPI_HEAD || (d.epi == UD_EPI_F16 && d.act != UD_ACT_GELU && d.rows_in == 0 && d.add == nullptr && (d.ldc & 3) == 0));
}

// Variant of the dense 128 x 128 kernel (N > 64; F16 / F32 / QKV epilogues): 0 = plain 2-stage ring, 1 = 4-stage pipelined ring,
// 2 = ring + two-way K split across CUs.  With at most one workgroup per CU anyway (tile count <= CU count: small batches) the
// plain kernel, tuned for two co-resident workgroups hiding each other's stalls, leaves the CU idle through every DMA / LDS round
// trip.  tile_hint 5 keeps the plain kernel, 6 forces the ring without the split, 7 the split (when the scratch is there).
inline int ring_variant(const UdGemm& d) {
  const int tiles = ((d.N + 127) / 128) * ((d.M + 127) / 128);
  const int nkt = d.K >> 6;
  if (d.amode != UD_A_DENSE || d.N <= 64 || (d.epi != UD_EPI_F16 && d.epi != UD_EPI_F32 && d.epi != UD_EPI_QKV)) return 0;
  if (!(d.groups <= 1 && nkt >= 4 && d.tile_hint != 5 && ((tiles <= 256 && nkt >= 8) || d.tile_hint >= 6) && !(ud_debug_flags_host() & 16))) return 0;
  if (d.epi != UD_EPI_QKV && d.splitk_ws && d.splitk_cnt && tiles <= 128 && (nkt & 1) == 0 && (nkt >= 16 || (d.tile_hint == 7 && nkt >= 8)) &&
      d.tile_hint != 6 && !(ud_debug_flags_host() & 32))
    return 2;
  return 1;
}

template <class C, int EPI, int AMODE>
int launch(const UdGemm& d, hipStream_t s) {
  const int tiles_n = (d.N + C::BN - 1) / C::BN;
  const int tiles_m = (d.M + C::BM - 1) / C::BM;
  if constexpr (C::BN == 128 && AMODE == UD_A_DENSE && (EPI == UD_EPI_F16 || EPI == UD_EPI_F32 || EPI == UD_EPI_QKV)) {
    const int variant = ring_variant(d);
    if (variant) {
      const int lds4 = 4 * C::STAGE_BYTES;
      if constexpr (EPI == UD_EPI_F16 || EPI == UD_EPI_F32) {
        if (variant == 2) {              // two-way K split across CUs: twice the workgroups, i.e. twice the DMA bytes in flight
          static bool attrs_set[UD_MAX_DEVICES];
          if (!ud_attr_once(attrs_set)) {
            if (hipFuncSetAttribute((const void*)gemm_kernel<C, EPI, AMODE, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds4) != hipSuccess) {
              ud_set_error("ud_gemm_f16: cannot reserve the 4-stage LDS ring");
              return UD_ERR_LAUNCH;
            }
          }
          hipLaunchKernelGGL((gemm_kernel<C, EPI, AMODE, 4, true>), dim3(2 * tiles_m * tiles_n), dim3(256), lds4, s, d);
          UD_CHECK_LAUNCH("ud_gemm_f16 (4-stage ring, K split) launch");
          return UD_OK;
        }
      }
      static bool attr4_set[UD_MAX_DEVICES];
      if (!ud_attr_once(attr4_set)) {
        if (hipFuncSetAttribute((const void*)gemm_kernel<C, EPI, AMODE, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds4) != hipSuccess) {
          ud_set_error("ud_gemm_f16: cannot reserve the 4-stage LDS ring");
          return UD_ERR_LAUNCH;
        }
      }
      hipLaunchKernelGGL((gemm_kernel<C, EPI, AMODE, 4>), dim3(tiles_m * tiles_n), dim3(256), lds4, s, d);
      UD_CHECK_LAUNCH("ud_gemm_f16 (4-stage ring) launch");
      return UD_OK;
    }
  }
  const int lds = 2 * C::STAGE_BYTES;
  static bool attr_set[UD_MAX_DEVICES];
  if (!ud_attr_once(attr_set)) (void)hipFuncSetAttribute((const void*)gemm_kernel<C, EPI, AMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  dim3 grid(tiles_m * tiles_n, 1, d.groups > 0 ? d.groups : 1);
  hipLaunchKernelGGL((gemm_kernel<C, EPI, AMODE>), grid, dim3(256), lds, s, d);
  UD_CHECK_LAUNCH("ud_gemm_f16 launch");
  return UD_OK;
}

template <int EPI, int AMODE>
int dispatch_bn(const UdGemm& d, hipStream_t s) {
  if (d.N > 64) return launch<Cfg<128, 64, 64>, EPI, AMODE>(d, s);
  if (d.N > 32) return launch<Cfg<64, 64, 32>, EPI, AMODE>(d, s);
  return launch<Cfg<32, 32, 32>, EPI, AMODE>(d, s);
}

// A grouped dense problem as ONE launch of the 256 x 256 kernel (gemm256_kernel GRP): groups stacked along M (A rows shared or stacked,
// outputs stacked), rows per group a multiple of the tile height.  Returns true and the merged descriptor when that form applies and
// the tile-shape model prefers the large kernel for the merged problem.
inline bool grouped_as_big(const UdGemm& d, UdGemm& m) {
  if (d.groups <= 1 || d.amode != UD_A_DENSE || (d.M & 255) || d.tile_hint == 1 || d.a_wrap || d.w_wrap || d.row_stats_in || d.row_stats_out || d.max_out ||
      d.rows_in || d.add)
    return false;
  // fp16 epilogues only: the fp32 (accumulate) epilogues of the x4 launches are traffic-bound (fp32 read + write of the stream) and ran
  // SLOWER as 344 large tiles with the residual preload exposed per tile (dh.out 68 -> 80 us, dh.fc2 147 -> 164 us, adapters 89 -> 86)
  // than as blockIdx.z slices of the 128-row kernel; dh.fc1 198 -> 135 us, dh.kv 87 -> 66 us, dh.q 50 -> 44 us (same-run table,
  // profiles/r03_ops_per_launch.tsv against the r3c5 run)
  // round 6 (tools/r6_dec_ab.py, same process, interleaved): the non-accumulating fp32 launch (the four input adapters, K = 1024) 80 -> 70 us as one
  // large-tile list; the accumulating ones unchanged or slower again (dh.out 58 / 58 us, dh.fc2 150 / 158 us)
#ifdef UD_AB_PREV
  if (d.epi != UD_EPI_F16 && d.epi != UD_EPI_QKV) return false;
#endif
  if (d.epi != UD_EPI_F16 && d.epi != UD_EPI_QKV && !(d.epi == UD_EPI_F32 && (d.tile_hint == 2 || (d.accumulate == 0 && d.K >= 1024)))) return false;
  if (!(d.gA == 0 || d.gA == (long long)d.M * d.lda) || d.gOut != (long long)d.M * d.ldc) return false;
  if (d.bias && d.gBias < 0) return false;
  if (d.out2) {
    if (d.epi == UD_EPI_QKV) {
      if (d.tok_per_img <= 0 || d.M % d.tok_per_img || d.gOut2 != (long long)(d.M / d.tok_per_img) * d.heads_v * 64 * d.kv_ld) return false;
    } else if (d.gOut2 != (long long)d.M * d.ldc2) {
      return false;
    }
  }
  m = d;
  m.grp_rows = d.M;
  m.M = d.M * d.groups;
  m.groups = 1;
  if (2.0 * (double)d.groups * d.N * d.ldw >= 2147483648.0 || 2.0 * (double)m.M * d.lda >= 2147483648.0) return false;
  m.tile_hint = 0;
  const int bt = pick_tiles(m);
  return bt == 4 || bt == 3 || bt == 8;         // the merged problem is large enough for the large-tile kernel (it then runs 256-row tiles)
}

}  // namespace

// ping-pong form of the 192-row tile for the fp32 residual-accumulate class (gemm_pp.hip): one-round tile lists (tile_hint 11 forces it
// where eligible, 0 picks it when the cost model would take the 192-row list and the list fits one round)
bool ud_gemm_pp_ok(const UdGemm& d);
int ud_gemm_pp_launch(const UdGemm& d, hipStream_t s);
bool ud_gemm_duo_ok(const UdGemm& d);
int ud_gemm_duo_launch(const UdGemm& d, hipStream_t s, int prio_mode);
namespace {
inline bool pp_pick(const UdGemm& d) {
#ifdef UD_AB_PREV      // A/B builds only (tools/r6/sessions.sh: ab/libprev.so = this tree with the round-6 schedules switched off, same ABI)
  return false;
#endif
  if (!ud_gemm_pp_ok(d)) return false;
  if (d.tile_hint == 11) return true;
  if (d.tile_hint != 0) return false;
  const int tiles = (d.N >> 8) * ((d.M + 191) / 192);
  return tiles <= 256 && tiles >= 128 && pick_tiles(d) == 3;
}
}  // namespace

extern "C" int ud_gemm_f16(const UdGemm* desc, void* stream) {
  UdGemm dcopy = *desc;
  if ((ud_debug_flags_host() & 4) && (dcopy.epi == UD_EPI_F16 || dcopy.epi == UD_EPI_QKV)) dcopy.ldc2 |= (1 << 30);
  if (ud_debug_flags_host() & 8) dcopy.tile_hint = 1;                                  // bisect: 128x128 tiles only                               // bisect: no LDS-staged stores
  const UdGemm& d = dcopy;
  hipStream_t s = (hipStream_t)stream;
  if (!d.A || !d.W || !d.out || d.M <= 0 || d.N <= 0 || d.K <= 0 || (d.K & 63) || (d.N & 3)) {
    ud_set_error("ud_gemm_f16: bad argument (need K % 64 == 0, N % 4 == 0)");
    return UD_ERR_BAD_ARG;
  }
  if (d.a_wrap < 0 || d.w_wrap < 0 || (d.amode == UD_A_DENSE ? (d.a_wrap & 63) : (d.a_wrap & 7)) || (d.w_wrap & 63) ||
      (d.a_wrap && d.w_wrap) || (d.w_wrap && d.amode != UD_A_DENSE) || (d.a_wrap && d.amode == UD_A_DENSE && 2 * d.a_wrap < d.K) ||
      (d.w_wrap && 2 * d.w_wrap < d.K) || (d.a_wrap && d.amode != UD_A_DENSE && 2 * d.a_wrap < d.Cin) ||
      ((d.a_wrap || d.w_wrap) && (d.amode == UD_A_CONV3_REFLECT_UP || d.groups > 1 && d.amode != UD_A_DENSE))) {
    ud_set_error("ud_gemm_f16: bad a_wrap / w_wrap (one operand may wrap, once: 2 * wrap >= K resp. Cin; dense: multiple of 64, conv: of 8)");
    return UD_ERR_BAD_ARG;
  }
  if (d.row_stats_in) {
    const int bt = d.amode == UD_A_DENSE ? pick_tiles(d) : 0;
    if (d.amode != UD_A_DENSE || (d.epi != UD_EPI_F16 && d.epi != UD_EPI_QKV) || !d.wsum || d.add || d.rows_in || (d.N & 15) || bt == 0) {
      ud_set_error("ud_gemm_f16: LayerNorm-folded consumer (row_stats_in) needs dense A, an fp16 epilogue without add / row remap, wsum, N % 16 == 0 "
                   "and a problem the large-tile kernel takes (ud_gemm_pick >= 3)");
      return UD_ERR_UNSUPPORTED;
    }
  }
  if (d.max_out && (d.epi != UD_EPI_F32 || d.rows_in || d.accumulate == 2)) {
    ud_set_error("ud_gemm_f16: max_out needs the fp32 epilogue without a row remap");
    return UD_ERR_UNSUPPORTED;
  }
  if (d.row_stats_out) {
    const int bt = d.amode == UD_A_DENSE || d.amode == UD_A_CONV3_ZERO ? pick_tiles(d) : 0;
    if (d.epi != UD_EPI_F32 || (d.N & 63) || (bt == 0 && (d.N <= 64 || d.groups > 1))) {
      ud_set_error("ud_gemm_f16: row_stats_out needs the fp32 epilogue, N % 64 == 0 and 64-column wave tiles (N > 64, no groups)");
      return UD_ERR_UNSUPPORTED;
    }
    if (d.row_stats_final && (bt == 0 || !d.row_stats_ticket || d.ln_D <= 0 || d.N > 1024 || (d.N & 127) || d.groups > 1)) {
      ud_set_error("ud_gemm_f16: row_stats_final (in-kernel reduction of the row statistics) needs the large-tile kernel, a ticket buffer, ln_D, "
                   "N <= 1024 and N % 128 == 0 (the finalizer reads the 64-column slabs in pairs)");
      return UD_ERR_UNSUPPORTED;
    }
  } else if (d.row_stats_final) {
    ud_set_error("ud_gemm_f16: row_stats_final without row_stats_out");
    return UD_ERR_BAD_ARG;
  }
  if (d.amode == UD_A_CONV3_REFLECT_UP && d.epi != UD_EPI_HEAD) {
    ud_set_error("ud_gemm_f16: CONV3_REFLECT_UP is implemented for the HEAD epilogue only");
    return UD_ERR_UNSUPPORTED;
  }
  if (d.amode != UD_A_DENSE && (!d.zeros || (d.Cin & 7) || d.K < 9 * d.Cin || d.rows_img < d.Himg * d.Wimg)) {
    ud_set_error("ud_gemm_f16: bad conv geometry");
    return UD_ERR_BAD_ARG;
  }
  if (d.epi == UD_EPI_QKV) {
    if (d.amode != UD_A_DENSE || !d.out2 || (d.vsplit % 128) || (d.tok_per_img & 3) || (d.kv_ld & 3) || d.N <= 64) {
      ud_set_error("ud_gemm_f16: bad QKV epilogue geometry");
      return UD_ERR_BAD_ARG;
    }
    {
      UdGemm mg;
      if (grouped_as_big(d, mg)) return launch256<4, UD_EPI_QKV, UD_A_DENSE, false, true>(mg, s);
    }
    if (const int bt = pick_tiles(d)) return launch_big<UD_EPI_QKV>(d, s, bt);
    return launch<Cfg<128, 64, 64>, UD_EPI_QKV, UD_A_DENSE>(d, s);
  }
  if (d.up_src && d.epi != UD_EPI_D2S) {
    ud_set_error("ud_gemm_f16: up_src belongs to the D2S epilogue");
    return UD_ERR_BAD_ARG;
  }
  if (d.epi == UD_EPI_D2S) {
    if (d.amode != UD_A_DENSE || (d.d2s_Co & 3)) {
      ud_set_error("ud_gemm_f16: bad D2S epilogue geometry");
      return UD_ERR_BAD_ARG;
    }
    if (d.up_src && (d.up_H < 1 || d.up_W < 1 || 2 * d.up_H != d.d2s_Hin * d.d2s_k || 2 * d.up_W != d.d2s_Win * d.d2s_k || (d.up_ld & 3) || d.up_ld < d.d2s_Co ||
                     d.up_img_rows < d.up_H * d.up_W || d.groups > 1)) {
      ud_set_error("ud_gemm_f16: up_src (fused x2 up-sampling under the D2S accumulate) needs 2 * up_H == d2s_Hin * d2s_k (same for W), up_ld % 4 == 0, "
                   "up_ld >= d2s_Co, up_img_rows >= up_H * up_W");
      return UD_ERR_BAD_ARG;
    }
    if (const int bt = pick_tiles(d)) return launch_big<UD_EPI_D2S>(d, s, bt);
    return dispatch_bn<UD_EPI_D2S, UD_A_DENSE>(d, s);
  }
  if (d.epi == UD_EPI_HEAD) {
    if ((d.amode != UD_A_CONV3_REFLECT && d.amode != UD_A_CONV3_REFLECT_UP) || d.N != 32 || !d.w2 || !d.bias) {
      ud_set_error("ud_gemm_f16: HEAD epilogue needs reflect conv, N == 32");
      return UD_ERR_BAD_ARG;
    }
    if (d.amode == UD_A_CONV3_REFLECT_UP) {
      // the loader stages a source patch of at most UPS_PATCH^2 pixels per 16 x 16 tile: 17 * scale + 3 <= UPS_PATCH
      const bool fits = 17.0 * (d.Hsrc - 1) / (d.Himg > 1 ? d.Himg - 1 : 1) + 3.0 <= UPS_PATCH && 17.0 * (d.Wsrc - 1) / (d.Wimg > 1 ? d.Wimg - 1 : 1) + 3.0 <= UPS_PATCH;
      if (!conv_tile_ok(d) || d.Hsrc < 1 || d.Wsrc < 1 || (d.cstride & 7) || !fits) {
        ud_set_error("ud_gemm_f16: CONV3_REFLECT_UP needs Cin % 64 == 0, dense images, Hsrc / Wsrc >= 1 and an up-sampling factor >= ~1.7");
        return UD_ERR_BAD_ARG;
      }
      if (d.Cin == 64 && head_regw_enabled()) {
        // weights in registers, persistent workgroups (conv_head_regw_kernel); UD_HEAD_REGW=0 keeps the LDS-streamed form (A/B)
        const int lds = HALO_BYTES + (UPS_PATCH * UPS_PATCH * 128 + 1023) / 1024 * 1024;
        static bool attr_set[UD_MAX_DEVICES];
        if (!ud_attr_once(attr_set)) (void)hipFuncSetAttribute((const void*)conv_head_regw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        const int B = d.M / d.rows_img;
        const int tiles_x = (d.Wimg + 15) >> 4, tiles_img = tiles_x * ((d.Himg + 15) >> 4), tiles_g = tiles_img * B;
        const int G = d.groups > 0 ? d.groups : 1;
        int wgs = (512 + G - 1) / G;                             // two workgroups per CU over all branches
        wgs = wgs < tiles_g ? wgs : tiles_g;
        hipLaunchKernelGGL(conv_head_regw_kernel, dim3(wgs, G), dim3(256), lds, s, d, tiles_x, tiles_img, tiles_g);
        UD_CHECK_LAUNCH("ud_gemm_f16 (head conv, register-resident weights) launch");
        return UD_OK;
      }
      return launch_conv_tile<2, UD_EPI_HEAD, true, true>(d, s);
    }
    if (conv_tile_ok(d)) return launch_conv_tile<2, UD_EPI_HEAD, true>(d, s);
    return launch<Cfg<32, 32, 32>, UD_EPI_HEAD, UD_A_CONV3_REFLECT>(d, s);
  }
  if (d.epi == UD_EPI_F16) {
    if (conv_tile_ok(d)) {
      if (d.amode == UD_A_CONV3_REFLECT) return d.N == 64 ? launch_conv_tile<4, UD_EPI_F16, true>(d, s) : launch_conv_tile<2, UD_EPI_F16, true>(d, s);
      return d.N == 64 ? launch_conv_tile<4, UD_EPI_F16, false>(d, s) : launch_conv_tile<2, UD_EPI_F16, false>(d, s);
    }
    {
      UdGemm mg;
      if (grouped_as_big(d, mg)) return launch256<4, UD_EPI_F16, UD_A_DENSE, false, true>(mg, s);
    }
    if (const int bt = pick_tiles(d))
      return d.amode == UD_A_DENSE ? launch_big<UD_EPI_F16>(d, s, bt) : launch_big<UD_EPI_F16, UD_A_CONV3_ZERO>(d, s, bt);
    if (d.amode == UD_A_DENSE) return dispatch_bn<UD_EPI_F16, UD_A_DENSE>(d, s);
    if (d.amode == UD_A_CONV3_ZERO) return dispatch_bn<UD_EPI_F16, UD_A_CONV3_ZERO>(d, s);
    return dispatch_bn<UD_EPI_F16, UD_A_CONV3_REFLECT>(d, s);
  }
  if (d.epi == UD_EPI_F32) {
    {
      UdGemm mg;
      if (grouped_as_big(d, mg)) return launch256<4, UD_EPI_F32, UD_A_DENSE, false, true>(mg, s);
    }
    if (d.tile_hint >= 12 && d.tile_hint <= 14 && ud_gemm_duo_ok(d)) return ud_gemm_duo_launch(d, s, d.tile_hint == 12 ? 1 : d.tile_hint == 13 ? 0 : 2);
    if (pp_pick(d)) return ud_gemm_pp_launch(d, s);
    if (const int bt = pick_tiles(d))
      return d.amode == UD_A_DENSE ? launch_big<UD_EPI_F32>(d, s, bt) : launch_big<UD_EPI_F32, UD_A_CONV3_ZERO>(d, s, bt);
    if (d.amode == UD_A_DENSE) return dispatch_bn<UD_EPI_F32, UD_A_DENSE>(d, s);
    if (d.amode == UD_A_CONV3_ZERO) return dispatch_bn<UD_EPI_F32, UD_A_CONV3_ZERO>(d, s);
  }
  ud_set_error("ud_gemm_f16: unsupported epi/amode combination");
  return UD_ERR_UNSUPPORTED;
}

#ifdef UD_TRACE
extern "C" int ud_trace_set(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(ud_trace_ptr), &buf, sizeof(buf)) == hipSuccess ? UD_OK : UD_ERR_LAUNCH;
}
#endif

// Which kernel ud_gemm_f16 would launch for this descriptor (for profiling labels): 0/1/2 = 128-row kernels with BN 128/64/32,
// 3 = 192x256 tiles, 4 = 256x256 tiles, 5 = halo-tile conv, 6 / 7 = 128x128 pipelined ring without / with the K split, 8 = row-balanced,
// 10 = 192x256 tiles with the two-way K split, 11 = 192x256 tiles in the ping-pong form (gemm_pp.hip);
// + 16 when the folded-LayerNorm consumer instantiation runs (row_stats_in), + 32 for a grouped problem run as one large-tile launch.
extern "C" int ud_gemm_pick(const UdGemm* desc) {
  const UdGemm& d = *desc;
  {
    UdGemm mg;
    if ((d.epi == UD_EPI_F16 || d.epi == UD_EPI_F32 || d.epi == UD_EPI_QKV) && grouped_as_big(d, mg)) return 4 + 32;
  }
  if (conv_tile_ok(d) && (d.epi == UD_EPI_HEAD || d.epi == UD_EPI_F16)) return 5;
  if (d.epi == UD_EPI_F32 && d.tile_hint >= 12 && d.tile_hint <= 14 && ud_gemm_duo_ok(d)) return 12;
  if (d.epi == UD_EPI_F32 && pp_pick(d)) return 11;
  if (d.epi != UD_EPI_HEAD) {
    int bt = pick_tiles(d);
    if (bt == 8 && d.row_stats_final) bt = 3;          // launch_big: the in-kernel statistics reduction runs on the 192-row tile list
    if (bt) return bt + (d.row_stats_in ? 16 : 0);
  }
  if (d.N > 64 && d.epi != UD_EPI_D2S) {
    const int v = ring_variant(d);
    if (v) return 5 + v;
  }
  return d.N > 64 ? 0 : (d.N > 32 ? 1 : 2);
}
