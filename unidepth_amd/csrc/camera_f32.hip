// fp32 island for the camera head (see UdLinearF32 in include/unidepth_hip.h for why): 4 tokens per image, ~9 M weights,
// latency-bound; plain fp32 FMAs are the right tool (no MFMA: at M = 4*B rows the matrix pipe would idle anyway).
#include "ud_common.h"
#include <mutex>
#include <type_traits>

namespace {

// out[m, n] for m < 32 rows (a batch chunk), 8 columns per workgroup: thread (m = tid & 31, c = tid >> 5) owns ONE output element
// and walks K sequentially -- no cross-lane reduction, a fixed summation order.  Both operands are staged through LDS in
// K-chunks of 256 floats by global_load_lds (one 1 KB row piece per wave-instruction, rows padded by 16 B so the 16 lanes of a
// ds_read_b128 group hit 64 distinct banks), double buffered; the W fragment of a column is a broadcast read.
// The first version (one wave per column, K split over the lanes, every wave re-reading the whole activation block through
// L1) ran 13-38 us per layer at M = 32 -- the 20-launch camera chain cost 0.43 ms per infer(); it was bound by the
// 32 x K x 4 B activation re-read per wave, not by the 1-4 MB of weights.
constexpr int LF_KC = 256;                    // floats per K-chunk
constexpr int LF_ROWB = LF_KC * 4 + 16;       // bytes per staged row
constexpr int LF_STAGE = 40 * LF_ROWB;        // 32 x rows + 8 W rows

__global__ __launch_bounds__(256) void linear_f32_kernel(const UdLinearF32 p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];      // 2 * LF_STAGE (83 KB: above the 64 KB static limit)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = tid & 31, c = tid >> 5;
  const int n0 = blockIdx.x * 8;
  const int m0 = blockIdx.y * 32;
  const int rows = (p.M - m0) < 32 ? (p.M - m0) : 32;
  const int nch = (p.K + LF_KC - 1) / LF_KC;

  // wave w stages x rows 8w .. 8w+7 and W rows 2w, 2w+1 (rows / columns past the end re-read the last valid one; K tail lanes
  // read offset 0 and are never multiplied)
  auto issue = [&](int ch, int stage) {
    char* sb = smem + stage * LF_STAGE;
    const int k = ch * LF_KC + lane * 4;
    const bool kok = k < p.K;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int r = wv * 8 + i;
      r = r < rows ? r : rows - 1;
      ud_glds16(p.x + (size_t)(m0 + r) * p.ldx + (kok ? k : 0), sb + (wv * 8 + i) * LF_ROWB);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int n = n0 + wv * 2 + i;
      n = n < p.N ? n : p.N - 1;
      ud_glds16(p.W + (size_t)n * p.ldw + (kok ? k : 0), sb + (32 + wv * 2 + i) * LF_ROWB);
    }
  };

  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;     // 4 independent FMA chains (k mod 4), summed in a fixed order at the end
  issue(0, 0);
  for (int ch = 0; ch < nch; ++ch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                          // chunk ch landed for every wave; everyone is done reading the other stage
    if (ch + 1 < nch) issue(ch + 1, (ch + 1) & 1);
    const char* xs = smem + (ch & 1) * LF_STAGE + m * LF_ROWB;
    const char* ws = smem + (ch & 1) * LF_STAGE + (32 + c) * LF_ROWB;
    const int kn = (p.K - ch * LF_KC) < LF_KC ? (p.K - ch * LF_KC) : LF_KC;      // multiple of 4
    if (kn == LF_KC) {
#pragma unroll 16
      for (int k4 = 0; k4 < LF_KC / 4; ++k4) {
        const f32x4 xv = *(const f32x4*)(xs + k4 * 16);
        const f32x4 wq = *(const f32x4*)(ws + k4 * 16);
        a0 = fmaf(xv[0], wq[0], a0); a1 = fmaf(xv[1], wq[1], a1); a2 = fmaf(xv[2], wq[2], a2); a3 = fmaf(xv[3], wq[3], a3);
      }
    } else {
      for (int k4 = 0; k4 < kn / 4; ++k4) {
        const f32x4 xv = *(const f32x4*)(xs + k4 * 16);
        const f32x4 wq = *(const f32x4*)(ws + k4 * 16);
        a0 = fmaf(xv[0], wq[0], a0); a1 = fmaf(xv[1], wq[1], a1); a2 = fmaf(xv[2], wq[2], a2); a3 = fmaf(xv[3], wq[3], a3);
      }
    }
  }
  const float acc = (a0 + a1) + (a2 + a3);
  const int n = n0 + c;
  if (m < rows && n < p.N) {
    const int mm = m0 + m;
    float y = acc;
    if (p.bias) y += p.bias[n];
    if (p.add) y += p.add[(size_t)(mm % p.add_mod) * p.ldadd + n];
    if (p.act == UD_ACT_GELU) y = 0.5f * y * (1.0f + erff(y * 0.70710678118654752440f));
    float* o = p.out + (size_t)mm * p.ldc + n;
    *o = p.accumulate ? *o + y : y;
  }
}

// one wave per (image, head): lane = channel d of the head (hd <= 64), T <= 8 tokens
__global__ __launch_bounds__(64) void attention_small_kernel(const float* q, const float* kv, float* out, int B, int T, int H, int C, float scale) {
  const int lane = threadIdx.x;
  const int h = blockIdx.x % H;
  const int b = blockIdx.x / H;
  const int hd = C / H;
  const bool on = lane < hd;
  float qv[8], kk[8], vv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const bool ok = on && i < T;
    const size_t row = (size_t)(b * T + (i < T ? i : 0));
    qv[i] = ok ? q[row * C + h * hd + lane] : 0.f;
    kk[i] = ok ? kv[row * 2 * C + h * hd + lane] : 0.f;
    vv[i] = ok ? kv[row * 2 * C + C + h * hd + lane] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i >= T) break;
    float s[8];
    float mx = -1e30f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      s[j] = j < T ? ud_wave_sum(qv[i] * kk[j]) * scale : -1e30f;
      mx = fmaxf(mx, s[j]);
    }
    float den = 0.f, o = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float pj = j < T ? expf(s[j] - mx) : 0.f;
      den += pj;
      o = fmaf(pj, vv[j], o);
    }
    if (on) out[(size_t)(b * T + i) * C + h * hd + lane] = o / den;
  }
}


// ================================================================================================================
// The whole camera head as one persistent grid (UdCameraHead in include/unidepth_hip.h; reference decoder.py:34-45,94-108).
// 256 threads: lane & 31 = K slice (float4 groups q = 32 i + slice of a row: 512 B contiguous per row and instruction), (lane >> 5) + 2 wave =
// row group rq: a thread owns rows rq + 8 j (j < 4) of a 32-row block, so one weight fragment read from LDS feeds four rows.  A workgroup
// owns ceil(N / G) output columns of a phase; their weights (a contiguous slab of W) were copied into LDS by global_load_lds during the
// previous phase.  Per output: four rows x 4 FMAs per fragment in k order inside the slice, then a fixed reduction tree over the 32 slices -- a
// summation order that does not depend on the row's position, the batch size or the grid size.
// ================================================================================================================
constexpr int CH_SLAB = 32 * 1024;             // bytes of one weight slab
constexpr int CH_LDS = 96 * 1024;              // two slabs; more than half a CU's LDS, so a CU holds one of these workgroups
constexpr unsigned CH_SPIN_LIMIT = 1u << 22;   // polls of the barrier counter (s_sleep + one uncached load each: several seconds) before a workgroup gives up

constexpr int CH_SYS = 17;                     // buffer cache policy sc0 | sc1: system scope (write-through / L2-bypassing)
__device__ __forceinline__ f32x4 ch_ld4(ud_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, CH_SYS));
}
__device__ __forceinline__ float ch_ld1(ud_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, CH_SYS));
}
__device__ __forceinline__ void ch_st1(ud_rsrc_t r, unsigned byte_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)byte_off, 0, CH_SYS);
}
__device__ __forceinline__ void ch_sum32x4(float (&v)[4]) {     // four independent sums: the shuffles of a step are in flight together
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) {
    float t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = __shfl_xor(v[j], o, 64);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += t[j];
  }
}

__global__ __launch_bounds__(256) void camera_head_kernel(const UdCameraHead p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];      // [2][CH_SLAB]
  __shared__ unsigned s_dead;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x, g = blockIdx.x;
  const int ks = lane & 31;
  const int rq = (lane >> 5) + 2 * wv;
  unsigned barriers = 0;
  const unsigned spin_limit = p.spin_limit ? p.spin_limit : CH_SPIN_LIMIT;
  if (tid == 0) s_dead = 0;

  // the weight slab of phase `ip` -> LDS buffer `buf`: rows [n0, n0 + ncol) of W are contiguous, copied in 1 KB wave pieces
  auto slab_issue = [&](int ip, int buf) {
    if (ip >= p.n_phases) return;
    const UdCamPhase& P = p.ph[ip];
    if (P.kind != 0) return;
    const int cpw = (P.N + G - 1) / G;
    const int n0 = g * cpw;
    int ncol = P.N - n0;
    ncol = ncol > cpw ? cpw : ncol;
    if (ncol <= 0) return;
    const int bytes = ncol * P.K * 4;
    const char* src = (const char*)(P.W + (size_t)n0 * P.K);
    char* dst = smem + buf * CH_SLAB;
    for (int i = wv; i * 1024 < bytes; i += 4) {
      int off = i * 1024 + lane * 16;
      off = off < bytes ? off : bytes - 16;                     // tail lanes re-read the last fragment (their LDS bytes are never used)
      ud_glds16(src + off, dst + i * 1024);
    }
  };

  auto linear = [&](const UdCamPhase& P, const char* slab) {
    const int cpw = (P.N + G - 1) / G;
    const int n0 = g * cpw;
    int ncol = P.N - n0;
    ncol = ncol > cpw ? cpw : ncol;
    if (ncol <= 0) return;
    const int K = P.K;
    const int GI = K >> 7;                                      // float4 groups per row and thread
    const int nchunk = (GI + 3) >> 2;                           // x lives in registers four groups at a time
    const ud_rsrc_t rx = ud_make_rsrc(P.x, (unsigned)P.M * (unsigned)P.ldx * 4u);
    const ud_rsrc_t ro = ud_make_rsrc(P.out, (unsigned)P.M * (unsigned)P.ldc * 4u);
    const float inv_k = 1.0f / (float)K;
    for (int rb = 0; rb * 32 < P.M; ++rb) {
      unsigned xoff[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int m = rb * 32 + rq + 8 * j;
        m = m < P.M ? m : P.M - 1;
        xoff[j] = ((unsigned)m * (unsigned)P.ldx + (unsigned)ks * 4u) * 4u;
      }
      f32x4 xr[4][4];
      auto load_x = [&](int ch) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int gi = ch * 4 + i;
#pragma unroll
          for (int j = 0; j < 4; ++j) xr[j][i] = gi < GI ? ch_ld4(rx, xoff[j] + (unsigned)gi * 512u) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      };
      if (nchunk == 1) {
        load_x(0);
        if (P.ln) {                                             // two-pass statistics, like layernorm_kernel: mean, then sum (x - mean)^2
          float st[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            st[j] = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) st[j] += (xr[j][i][0] + xr[j][i][1]) + (xr[j][i][2] + xr[j][i][3]);
          }
          ch_sum32x4(st);
          float mean[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            mean[j] = st[j] * inv_k;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              if (i < GI) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const float d = xr[j][i][e] - mean[j];
                  q = fmaf(d, d, q);
                }
              }
            }
            st[j] = q;
          }
          ch_sum32x4(st);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float rstd = rsqrtf(st[j] * inv_k + p.eps);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int e = 0; e < 4; ++e) xr[j][i][e] = (xr[j][i][e] - mean[j]) * rstd;
          }
        }
      }
      for (int cb = 0; cb * 4 < ncol; ++cb) {
        float acc[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[c][j] = 0.f;
        const char* wcol[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          int col = cb * 4 + c;
          col = col < ncol ? col : ncol - 1;
          wcol[c] = slab + ((size_t)col * K + (size_t)ks * 4) * 4;
        }
        // the 16 sums of a row group are reduce-scattered over the K slices (below): lane ks ends up with sum number ks >> 1 = 4 c + j, and the
        // even lane finishes output (column cb * 4 + c, row rq + 8 j).  What its epilogue reads (bias, positional add, the old value of an
        // accumulate) is requested NOW and lands under the FMAs instead of one memory latency after them
        const int ec = ks >> 3, ej = (ks >> 1) & 3;
        const int ecol = cb * 4 + ec;
        const int em = rb * 32 + rq + 8 * ej;
        const bool emine = !(ks & 1) && ecol < ncol && em < P.M;
        const int en = n0 + ecol;
        const unsigned eoff = ((unsigned)em * (unsigned)P.ldc + (unsigned)en) * 4u;
        float e_bias = 0.f, e_add = 0.f, e_old = 0.f;
        if (emine) {
          if (P.bias) e_bias = P.bias[en];
          if (P.add && en < P.add_cols) e_add = P.add[(size_t)(em % P.add_mod) * P.ldadd + en];
          if (P.accumulate) e_old = ch_ld1(ro, eoff);
        }
        for (int ch = 0; ch < nchunk; ++ch) {
          if (nchunk > 1) load_x(ch);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (ch * 4 + i < GI) {
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const f32x4 wq = *(const f32x4*)(wcol[c] + (ch * 4 + i) * 512);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  float a = acc[c][j];
                  a = fmaf(xr[j][i][0], wq[0], a);
                  a = fmaf(xr[j][i][1], wq[1], a);
                  a = fmaf(xr[j][i][2], wq[2], a);
                  a = fmaf(xr[j][i][3], wq[3], a);
                  acc[c][j] = a;
                }
              }
            }
          }
        }
        // 16 sums per row group, 32 partials each: halve the set of sums a lane carries at every exchange (xor 16, 8, 4, 2: 8 + 4 + 2 + 1
        // shuffles, all of a step in flight together), then one plain exchange (xor 1).  A fixed tree: the result depends on nothing but k.
        float r8[8], r4[4], r2[2];
        {
          const bool up = ks & 16;
#pragma unroll
          for (int v = 0; v < 8; ++v) {
            const float keep = up ? acc[(v + 8) >> 2][(v + 8) & 3] : acc[v >> 2][v & 3];
            const float send = up ? acc[v >> 2][v & 3] : acc[(v + 8) >> 2][(v + 8) & 3];
            r8[v] = keep + __shfl_xor(send, 16, 64);
          }
        }
        {
          const bool up = ks & 8;
#pragma unroll
          for (int v = 0; v < 4; ++v) r4[v] = (up ? r8[v + 4] : r8[v]) + __shfl_xor(up ? r8[v] : r8[v + 4], 8, 64);
        }
        {
          const bool up = ks & 4;
#pragma unroll
          for (int v = 0; v < 2; ++v) r2[v] = (up ? r4[v + 2] : r4[v]) + __shfl_xor(up ? r4[v] : r4[v + 2], 4, 64);
        }
        float mine;
        {
          const bool up = ks & 2;
          mine = (up ? r2[1] : r2[0]) + __shfl_xor(up ? r2[0] : r2[1], 2, 64);
        }
        mine += __shfl_xor(mine, 1, 64);
        if (emine) {
          float y = (mine + e_bias) + e_add;
          if (P.act == UD_ACT_GELU) y = 0.5f * y * (1.0f + erff(y * 0.70710678118654752440f));
          ch_st1(ro, eoff, y + e_old);
        }
      }
    }
  };

  // one wave per (image, head): lane = channel of the head, T <= TT tokens (the arithmetic of attention_small_kernel)
  auto attention = [&](const UdCamPhase& P, auto TTAG) {
    constexpr int TT = decltype(TTAG)::value;
    const int T = p.T, H = p.H, C = p.C, hd = C / H;
    const int B = P.M / T;
    const ud_rsrc_t rx = ud_make_rsrc(P.x, (unsigned)P.M * (unsigned)P.ldx * 4u);
    const ud_rsrc_t ro = ud_make_rsrc(P.out, (unsigned)P.M * (unsigned)P.ldc * 4u);
    const bool on = lane < hd;
    for (int pair = g * 4 + wv; pair < B * H; pair += 4 * G) {
      const int h = pair % H, b = pair / H;
      float qv[TT], kk[TT], vv[TT];
#pragma unroll
      for (int i = 0; i < TT; ++i) {
        const bool ok = on && i < T;
        const unsigned row = (unsigned)(b * T + (i < T ? i : 0));
        const unsigned o = (row * (unsigned)P.ldx + (unsigned)(h * hd + (on ? lane : 0))) * 4u;
        const float q = ch_ld1(rx, o), k = ch_ld1(rx, o + (unsigned)C * 4u), v = ch_ld1(rx, o + (unsigned)C * 8u);
        qv[i] = ok ? q : 0.f; kk[i] = ok ? k : 0.f; vv[i] = ok ? v : 0.f;
      }
      // all score partials first, then ONE butterfly over the 64 channels with every exchange of a step in flight together (one dependent
      // 6-step chain per score cost ~5 us per phase)
      float sc[TT][TT];
#pragma unroll
      for (int i = 0; i < TT; ++i)
#pragma unroll
        for (int jj = 0; jj < TT; ++jj) sc[i][jj] = qv[i] * kk[jj];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        float tt[TT][TT];
#pragma unroll
        for (int i = 0; i < TT; ++i)
#pragma unroll
          for (int jj = 0; jj < TT; ++jj) tt[i][jj] = __shfl_xor(sc[i][jj], o, 64);
#pragma unroll
        for (int i = 0; i < TT; ++i)
#pragma unroll
          for (int jj = 0; jj < TT; ++jj) sc[i][jj] += tt[i][jj];
      }
#pragma unroll
      for (int i = 0; i < TT; ++i) {
        if (i < T) {
          float mx = -1e30f;
#pragma unroll
          for (int jj = 0; jj < TT; ++jj) {
            sc[i][jj] = jj < T ? sc[i][jj] * p.scale : -1e30f;
            mx = fmaxf(mx, sc[i][jj]);
          }
          float den = 0.f, o = 0.f;
#pragma unroll
          for (int jj = 0; jj < TT; ++jj) {
            const float pj = jj < T ? expf(sc[i][jj] - mx) : 0.f;
            den += pj;
            o = fmaf(pj, vv[jj], o);
          }
          if (on) ch_st1(ro, ((unsigned)(b * T + i) * (unsigned)P.ldc + (unsigned)(h * hd + lane)) * 4u, o / den);
        }
      }
    }
  };

#ifdef UD_CAM_TRACE
  // tools/trace_camera_head.py: 100 MHz stamps of workgroup 0 (and the last one) per phase -- start, work done (stores issued), stores
  // acknowledged, barrier passed -- behind the 16 counter words
  unsigned long long* const trace = (unsigned long long*)(p.sync_ws + 16) + (g == 0 ? 0 : UD_CAM_MAX_PHASES * 4);
#define CH_STAMP(k) do { if (tid == 0 && (g == 0 || g == G - 1)) trace[ip * 4 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CH_STAMP(k) do { } while (0)
#endif
  slab_issue(0, 0);
  for (int ip = 0; ip < p.n_phases; ++ip) {
    const UdCamPhase& P = p.ph[ip];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this phase's slab has landed (issued one phase ago)
    __syncthreads();                                            // ... for every wave; everyone is done with the other buffer
    CH_STAMP(0);
    slab_issue(ip + 1, (ip + 1) & 1);
    if (P.kind == 1) {
      if (p.T <= 4) attention(P, std::integral_constant<int, 4>{});
      else attention(P, std::integral_constant<int, 8>{});
    }
    else linear(P, smem + (ip & 1) * CH_SLAB);
    CH_STAMP(1);
#ifdef UD_CAM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CH_STAMP(2);
#endif
    if (P.sync) {
      // grid barrier: this workgroup's write-through stores are acknowledged (vmcnt), then one arrival per workgroup on an agent-scope
      // counter; the loads behind the barrier bypass the (per-XCD, mutually non-coherent) L2s
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      ++barriers;
      if (tid == 0 && !s_dead) {
        __hip_atomic_fetch_add(p.sync_ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = barriers * (unsigned)G;
        unsigned spins = 0;
        while (__hip_atomic_load(p.sync_ws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(4);
          if (++spins > spin_limit) {                           // the grid is not co-resident (never on a whole MI355X): give up, flag it
            __hip_atomic_store(p.sync_ws + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p.fail_host) __hip_atomic_store(p.fail_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            s_dead = 1;
            break;
          }
        }
      }
      __syncthreads();
    }
    CH_STAMP(3);
  }
  // the last workgroup out re-arms the counters for the next launch (everybody is past every barrier by then)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // a timed-out barrier (this launch or an earlier one the caller has not acknowledged) makes the result LOUDLY wrong: the last phase's
  // output (the camera parameters every ray and depth value derives from) becomes NaN.  Every workgroup that ends with the flag set writes
  // the whole (tiny) output after its own last store; the workgroup that set the flag ends after setting it, so whichever normal store comes
  // last, a NaN store follows it.
  if (__hip_atomic_load(p.sync_ws + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
    const UdCamPhase& L = p.ph[p.n_phases - 1];
    if (L.kind == 0) {
      const ud_rsrc_t ro = ud_make_rsrc(L.out, (unsigned)L.M * (unsigned)L.ldc * 4u);
      for (int i = tid; i < L.M * L.N; i += 256)
        ch_st1(ro, ((unsigned)(i / L.N) * (unsigned)L.ldc + (unsigned)(i % L.N)) * 4u, __builtin_nanf(""));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(p.sync_ws + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (unsigned)G - 1u) {
      __hip_atomic_store(p.sync_ws, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p.sync_ws + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace

extern "C" int ud_linear_f32(const UdLinearF32* desc, void* stream) {
  const UdLinearF32& d = *desc;
  if (!d.x || !d.W || !d.out || d.M <= 0 || d.N <= 0 || d.K <= 0 || (d.K & 3) || (d.ldx & 3) || (d.ldw & 3) || (d.add && d.add_mod <= 0)) {
    ud_set_error("ud_linear_f32: bad argument (K, ldx, ldw % 4 == 0)");
    return UD_ERR_BAD_ARG;
  }
  static bool attr_set[UD_MAX_DEVICES];
  if (!ud_attr_once(attr_set)) {
    if (hipFuncSetAttribute((const void*)linear_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LF_STAGE) != hipSuccess) {
      ud_set_error("ud_linear_f32: cannot reserve the LDS staging ring");
      return UD_ERR_LAUNCH;
    }
  }
  hipLaunchKernelGGL(linear_f32_kernel, dim3((d.N + 7) / 8, (d.M + 31) / 32), dim3(256), 2 * LF_STAGE, (hipStream_t)stream, d);
  UD_CHECK_LAUNCH("ud_linear_f32 launch");
  return UD_OK;
}

extern "C" int ud_attention_small_f32(const float* q, const float* kv, float* out, int B, int T, int H, int C, float scale, void* stream) {
  if (!q || !kv || !out || B <= 0 || T <= 0 || T > 8 || H <= 0 || C % H || C / H > 64) {
    ud_set_error("ud_attention_small_f32: bad argument (T <= 8)");
    return UD_ERR_BAD_ARG;
  }
  hipLaunchKernelGGL(attention_small_kernel, dim3(B * H), dim3(64), 0, (hipStream_t)stream, q, kv, out, B, T, H, C, scale);
  UD_CHECK_LAUNCH("ud_attention_small_f32 launch");
  return UD_OK;
}

namespace {
// the grid must be co-resident (grid barriers): at most one workgroup per CU
int camera_head_grid(const UdCameraHead& d) {
  int G = d.workgroups > 0 ? d.workgroups : 128;
  static int cus[UD_MAX_DEVICES];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= UD_MAX_DEVICES) dev = 0;
  if (!cus[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;     // no device (host-side checks): an MI355X
    cus[dev] = n;
  }
  return G < cus[dev] ? G : cus[dev];
}

int camera_head_check(const UdCameraHead* desc, int G) {
  if (!desc || !desc->sync_ws || desc->n_phases <= 0 || desc->n_phases > UD_CAM_MAX_PHASES) {
    ud_set_error("ud_camera_head_f32: bad argument (sync_ws, 1 <= n_phases <= UD_CAM_MAX_PHASES)");
    return UD_ERR_BAD_ARG;
  }
  const UdCameraHead& d = *desc;
  for (int i = 0; i < d.n_phases; ++i) {
    const UdCamPhase& P = d.ph[i];
    if (!P.x || !P.out || P.M <= 0) {
      ud_set_error("ud_camera_head_f32: phase without x / out / rows");
      return UD_ERR_BAD_ARG;
    }
    if (P.kind == 1) {
      if (d.T <= 0 || d.T > 8 || d.H <= 0 || d.C <= 0 || d.C % d.H || d.C / d.H > 64 || P.M % d.T || P.ldx < 3 * d.C || P.ldc < d.C) {
        ud_set_error("ud_camera_head_f32: attention phase needs T <= 8, C / H <= 64, rows % T == 0, packed [q | k | v] rows");
        return UD_ERR_UNSUPPORTED;
      }
      continue;
    }
    const int cpw = (P.N + G - 1) / G;
    if (P.kind != 0 || !P.W || P.N <= 0 || P.K <= 0 || (P.K & 127) || (P.ldx & 3) || P.ldx < P.K || P.ldc < P.N || (P.add && (P.add_mod <= 0 || P.ldadd < P.add_cols)) ||
        ((long long)cpw * P.K * 4 + 1023) / 1024 * 1024 > CH_SLAB || (P.ln && P.K > 512) || (P.act != UD_ACT_NONE && P.act != UD_ACT_GELU)) {
      ud_set_error("ud_camera_head_f32: linear phase outside the kernel's limits (K % 128 == 0, ceil(N / workgroups) * K * 4 <= 32 KB, LayerNorm K <= 512, "
                   "ldx % 4 == 0, act NONE / GELU)");
      return UD_ERR_UNSUPPORTED;
    }
  }
  return UD_OK;
}
}  // namespace

extern "C" int ud_camera_head_supported(const UdCameraHead* desc) {
  return desc ? camera_head_check(desc, camera_head_grid(*desc)) : UD_ERR_BAD_ARG;
}

extern "C" int ud_camera_head_f32(const UdCameraHead* desc, void* stream) {
  if (!desc) {
    ud_set_error("ud_camera_head_f32: null descriptor");
    return UD_ERR_BAD_ARG;
  }
  const int G = camera_head_grid(*desc);
  if (const int rc = camera_head_check(desc, G)) return rc;
  const UdCameraHead& d = *desc;
  static bool attr_set[UD_MAX_DEVICES];
  if (!ud_attr_once(attr_set)) {
    if (hipFuncSetAttribute((const void*)camera_head_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS) != hipSuccess) {
      ud_set_error("ud_camera_head_f32: cannot reserve the weight slabs in LDS");
      return UD_ERR_LAUNCH;
    }
  }
  // one spinning grid at a time per device: a launch waits (on its own stream) for the previous camera-head launch of this process, wherever
  // that was enqueued.  Two of these grids on different streams (pipelined requests) could otherwise each be PARTLY resident and starve each
  // other at their barriers.  Two host calls per launch, nothing on the GPU when the streams are the same.
#ifdef UD_AB_PREV       // A/B builds only (tools/r6/sessions.sh): the launch without the ordering events
  hipLaunchKernelGGL(camera_head_kernel, dim3(G), dim3(256), CH_LDS, (hipStream_t)stream, d);
  UD_CHECK_LAUNCH("ud_camera_head_f32 launch");
  return UD_OK;
#endif
  static std::mutex mu;
  static hipEvent_t last[UD_MAX_DEVICES];
  static bool have[UD_MAX_DEVICES];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= UD_MAX_DEVICES) dev = 0;
  std::lock_guard<std::mutex> lk(mu);
  if (have[dev]) {
    if (hipStreamWaitEvent((hipStream_t)stream, last[dev], 0) != hipSuccess) {
      ud_set_error("ud_camera_head_f32: cannot order the launch behind the previous camera-head launch");
      return UD_ERR_LAUNCH;
    }
  } else if (hipEventCreateWithFlags(&last[dev], hipEventDisableTiming) != hipSuccess) {
    ud_set_error("ud_camera_head_f32: cannot create the ordering event");
    return UD_ERR_LAUNCH;
  }
  have[dev] = true;
  hipLaunchKernelGGL(camera_head_kernel, dim3(G), dim3(256), CH_LDS, (hipStream_t)stream, d);
  UD_CHECK_LAUNCH("ud_camera_head_f32 launch");
  if (hipEventRecord(last[dev], (hipStream_t)stream) != hipSuccess) {
    ud_set_error("ud_camera_head_f32: cannot record the ordering event");
    return UD_ERR_LAUNCH;
  }
  return UD_OK;
}
