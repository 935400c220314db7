// Decoder-side kernels of the UniDepthV1 path that are not GEMMs / flash attention / LayerNorm (reference
// unidepth/models/unidepthv1/decoder.py, unidepthv1.py:28-98,288-373, utils/geometric.py, utils/sht.py:833, layers/nystrom_attention.py).
// One C-ABI entry, ud_v1_op(), dispatches on UdV1Op.kind (include/unidepth_hip.h documents every kind and cites what it replaces).
#include "ud_common.h"
#include <type_traits>

namespace {

// ------------------------------------------------------------------------------------------------ antialiased bilinear resize
// F.interpolate(mode="bilinear", align_corners=False, antialias=True) on NHWC fp32 (C % 4 == 0, 4 channels per thread): separable
// triangle filter of support max(scale, 1) per axis, taps normalised to sum 1 (ATen's _upsample_bilinear2d_aa; for up-sampling this
// is plain bilinear).  Thread = (output pixel, 4 channels); the (<= ~6 x 6) taps are walked directly.
__device__ __forceinline__ void aa_taps(int o, float scale, int n_in, int& lo, int& cnt, float& center, float& inv) {
  const float support = scale >= 1.0f ? scale : 1.0f;
  inv = scale >= 1.0f ? 1.0f / scale : 1.0f;
  center = scale * ((float)o + 0.5f);
  lo = (int)(center - support + 0.5f);
  lo = lo < 0 ? 0 : lo;
  int hi = (int)(center + support + 0.5f);
  hi = hi > n_in ? n_in : hi;
  cnt = hi - lo;
}
__device__ __forceinline__ float aa_w(int j, int lo, float center, float inv) {
  const float w = 1.0f - fabsf(((float)(j + lo) - center + 0.5f) * inv);
  return w < 0.f ? 0.f : w;
}

__global__ __launch_bounds__(256) void resize_aa_kernel(const float* in, float* out, int B, int Hi, int Wi, int Ho, int Wo, int C, int ldi, int ldo,
                                                        int y0, int x0, int Hc, int Wc) {
  // output = resize(in[:, y0:y0+Hc, x0:x0+Wc]) to (Ho, Wo): the crop window folds _postprocess' pad removal into the second resize
  const int CG = C >> 2;
  const long long total = (long long)B * Ho * Wo * CG;
  const float sy = (float)Hc / (float)Ho, sx = (float)Wc / (float)Wo;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int cg = (int)(idx % CG);
    const long long pix = idx / CG;
    const int ox = (int)(pix % Wo);
    const int oy = (int)((pix / Wo) % Ho);
    const int b = (int)(pix / ((long long)Wo * Ho));
    int ylo, yn, xlo, xn;
    float yc, yi, xc, xi;
    aa_taps(oy, sy, Hc, ylo, yn, yc, yi);
    aa_taps(ox, sx, Wc, xlo, xn, xc, xi);
    float wys = 0.f, wxs = 0.f;
    for (int j = 0; j < yn; ++j) wys += aa_w(j, ylo, yc, yi);
    for (int j = 0; j < xn; ++j) wxs += aa_w(j, xlo, xc, xi);
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* src = in + ((size_t)b * Hi * Wi) * ldi + cg * 4;
    for (int jy = 0; jy < yn; ++jy) {
      const float wy = aa_w(jy, ylo, yc, yi) / wys;
      f32x4 row = (f32x4){0.f, 0.f, 0.f, 0.f};
      const float* r = src + ((size_t)(y0 + ylo + jy) * Wi + x0 + xlo) * ldi;
      for (int jx = 0; jx < xn; ++jx) row += (aa_w(jx, xlo, xc, xi) / wxs) * *(const f32x4*)(r + (size_t)jx * ldi);
      acc += wy * row;
    }
    *(f32x4*)(out + (size_t)pix * ldo + cg * 4) = acc;
  }
}

// ------------------------------------------------------------------------------------------------ spherical-harmonics ray embedding
// Antialiased down-sampling of the planar ray map [nb,3,Hn,Wn] to (h, w) (flat_interpolate, decoder.py:205-219),
// F.normalize, the 81 real spherical harmonics of degree <= 8 (utils/sht.py:833 rsh_cart_8 is a generated closed-form table; here the
// standard recurrences: Q_m^m = (2m-1)!!, Q_{m+1}^m = (2m+1) z Q_m^m, (l-m) Q_l^m = (2l-1) z Q_{l-1}^m - (l+m-1) Q_{l-2}^m,
// (x+iy)^m = A_m + i B_m, Y_l^{+-m} = (-1)^m sqrt2 K_l^m Q_l^m {A_m, B_m}), then LayerNorm statistics over the 81 values (the MLP's
// norm, affine folded into proj1) -> fp16 row of 128 (columns 81.. stay zero).
template <int I, int N, typename F>
__device__ __forceinline__ void sh_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sh_static_for<I + 1, N>(f);
  }
}

// Normalisation K_l^m of Y_l^{+-m} (times sqrt2 (-1)^m for m > 0), row l at offset l (l + 1) / 2: evaluated in double, rounded once.
__constant__ float SH_K[45] = {
    2.820947918e-01f, 4.886025119e-01f, -4.886025119e-01f, 6.307831305e-01f, -3.641828102e-01f, 1.820914051e-01f,
    7.463526652e-01f, -3.046971996e-01f, 9.635371475e-02f, -3.933623933e-02f, 8.462843753e-01f, -2.676186174e-01f,
    6.307831305e-02f, -1.685838828e-02f, 5.960340338e-03f, 9.356025796e-01f, -2.415715473e-01f, 4.565273129e-02f,
    -9.318824751e-03f, 2.196468058e-03f, -6.945841871e-04f, 1.017107236e+00f, -2.219509952e-01f, 3.509353370e-02f,
    -5.848922283e-03f, 1.067862224e-03f, -2.276689911e-04f, 6.572237664e-05f, 1.092548431e+00f, -2.064722459e-01f,
    2.809731381e-02f, -3.973560225e-03f, 5.990367431e-04f, -9.983945719e-05f, 1.958012848e-05f, -5.233009454e-06f,
    1.163106623e+00f, -1.938511038e-01f, 2.316963852e-02f, -2.851985351e-03f, 3.681897256e-04f, -5.105872827e-05f,
    7.878532816e-06f, -1.438416714e-06f, 3.596041786e-07f};

// Two phases per wave of 64 tokens.  (1) the antialiased average of the ray map over a token's footprint is a 64-lane reduction (up to
// 32 x 32 taps at 1/16 resolution): the wave walks its 64 tokens one after the other and lane j keeps token j's direction.  (2) every lane
// evaluates all 81 harmonics of ITS token with the recurrences fully unrolled (~400 flops, no divergence) and LayerNorms them in
// registers.  The first version ran one token per wave with lane i evaluating harmonic i through loops whose trip counts depend on the
// lane: 534 us for the 307200 tokens of the 1/4 level at bs 16, all of it divergent scalar-style work.
__global__ __launch_bounds__(256) void sh_embed_kernel(const float* rays, half_t* out, int nb, int Hn, int Wn, int h, int w, int ldo, int rows_per_img, float eps, int tpw) {
  const int lane = threadIdx.x & 63;
  const int ntw = tpw < 0 ? 64 : tpw;                 // tokens per wave (<= 64); tpw < 0: 64, every lane averaging its own footprint
  const long long tok0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * ntw;
  const int hw = h * w;
  const long long ntok = (long long)nb * hw;
  if (tok0 >= ntok) return;
  const size_t HW = (size_t)Hn * Wn;
  float x = 0.f, y = 0.f, z = 1.f;
  const int cnt = ntok - tok0 < ntw ? (int)(ntok - tok0) : ntw;
  if (tpw < 0) {
    // small footprints (<= ~8 x 8 taps, the 1/4 level): lane = token from the start -- 64 tokens x (one 64-lane pass + four wave sums) cost
    // five times the 192 loads + FMAs a lane spends on its own footprint, and neighbouring lanes' footprints overlap in L1
    const long long tok = tok0 + lane;
    if (tok < ntok) {
      const int img = (int)(tok / hw), t = (int)(tok - (long long)img * hw);
      const int ty = t / w, tx = t - ty * w;
      int ylo, yn, xlo, xn;
      float yc, yi, xc, xi;
      aa_taps(ty, (float)Hn / (float)h, Hn, ylo, yn, yc, yi);
      aa_taps(tx, (float)Wn / (float)w, Wn, xlo, xn, xc, xi);
      const float* r = rays + (size_t)img * 3 * HW;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, ws = 0.f;
      for (int iy = 0; iy < yn; ++iy) {
        const float wy = aa_w(iy, ylo, yc, yi);
        const float* rr = r + (size_t)(ylo + iy) * Wn + xlo;
        for (int ix = 0; ix < xn; ++ix) {
          const float wgt = wy * aa_w(ix, xlo, xc, xi);
          a0 += wgt * rr[ix]; a1 += wgt * rr[ix + HW]; a2 += wgt * rr[ix + 2 * HW];
          ws += wgt;
        }
      }
      x = a0 / ws; y = a1 / ws; z = a2 / ws;
    }
  } else
  for (int j = 0; j < cnt; ++j) {
    const long long tok = tok0 + j;
    const int img = (int)(tok / hw), t = (int)(tok - (long long)img * hw);
    const int ty = t / w, tx = t - ty * w;
    int ylo, yn, xlo, xn;
    float yc, yi, xc, xi;
    aa_taps(ty, (float)Hn / (float)h, Hn, ylo, yn, yc, yi);
    aa_taps(tx, (float)Wn / (float)w, Wn, xlo, xn, xc, xi);
    const float* r = rays + (size_t)img * 3 * HW;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, ws = 0.f;
    for (int i = lane; i < yn * xn; i += 64) {
      const int iy = i / xn, ix = i - iy * xn;
      const float wgt = aa_w(iy, ylo, yc, yi) * aa_w(ix, xlo, xc, xi);
      const size_t off = (size_t)(ylo + iy) * Wn + xlo + ix;
      a0 += wgt * r[off]; a1 += wgt * r[off + HW]; a2 += wgt * r[off + 2 * HW];
      ws += wgt;
    }
    a0 = ud_wave_sum(a0); a1 = ud_wave_sum(a1); a2 = ud_wave_sum(a2); ws = ud_wave_sum(ws);
    if (lane == j) { x = a0 / ws; y = a1 / ws; z = a2 / ws; }
  }
  if (lane >= cnt) return;
  const float inv = 1.0f / fmaxf(sqrtf(x * x + y * y + z * z), 1e-12f);
  x *= inv; y *= inv; z *= inv;
  float A[9], Bv[9];                                  // (x + i y)^m = A_m + i B_m
  A[0] = 1.f; Bv[0] = 0.f;
#pragma unroll
  for (int m = 1; m <= 8; ++m) {
    A[m] = A[m - 1] * x - Bv[m - 1] * y;
    Bv[m] = A[m - 1] * y + Bv[m - 1] * x;
  }
  float v[81];
  float qmm = 1.f;
  sh_static_for<0, 9>([&](auto Mc) {                  // compile-time (l, m): every v[] index is a constant, the table stays in registers
    constexpr int m = decltype(Mc)::value;
    if (m > 0) qmm *= (float)(2 * m - 1);
    float q2 = 0.f, q1 = qmm;                         // Q_{l-2}^m, Q_{l-1}^m
    sh_static_for<m, 9>([&](auto Lc) {
      constexpr int l = decltype(Lc)::value;
      float q;
      if constexpr (l == m) q = qmm;
      else if constexpr (l == m + 1) q = (float)(2 * m + 1) * z * q1;
      else q = ((float)(2 * l - 1) * z * q1 - (float)(l + m - 1) * q2) / (float)(l - m);
      if constexpr (l > m) { q2 = q1; q1 = q; }
      const float kq = SH_K[l * (l + 1) / 2 + m] * q;
      if constexpr (m == 0) v[l * (l + 1)] = kq;
      else {
        v[l * (l + 1) + m] = kq * A[m];
        v[l * (l + 1) - m] = kq * Bv[m];
      }
    });
  });
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 81; ++i) sum += v[i];
  const float mean = sum / 81.0f;
  float var = 0.f;
#pragma unroll
  for (int i = 0; i < 81; ++i) { v[i] -= mean; var = fmaf(v[i], v[i], var); }
  const float rstd = rsqrtf(var / 81.0f + eps);
  const long long tok = tok0 + lane;
  const int img = (int)(tok / hw), t = (int)(tok - (long long)img * hw);
  half_t* row = out + ((size_t)img * rows_per_img + t) * ldo;
  if ((ldo & 7) == 0) {                               // 16-byte stores: ten chunks of 8 + the 81st value
#pragma unroll
    for (int c = 0; c < 10; ++c) {
      half8 hv;
#pragma unroll
      for (int e = 0; e < 8; ++e) hv[e] = (half_t)(v[c * 8 + e] * rstd);
      *(half8*)(row + c * 8) = hv;
    }
    row[80] = (half_t)(v[80] * rstd);
  } else {
#pragma unroll
    for (int i = 0; i < 81; ++i) row[i] = (half_t)(v[i] * rstd);
  }
}

// ------------------------------------------------------------------------------------------------ row softmax (fp32 scores -> fp16 probabilities)
// out[r, :N] = softmax(scale * in[r, :N]); columns N..ldo-1 are written as zeros (K padding of the following P V GEMM).  One wave per row.
// Rows that fit the wave's registers (N <= 256 NV, N and the strides multiples of 4): one 16-byte load per 4 scores, ONE exp per score, 8-byte
// fp16 stores -- the generic kernel below reads every row three times with 4-byte loads and exponentiates twice (226 us for the 16 x 1200
// x 4800 scores of aggregate_16 at bs 16: 368 MB in, 184 MB out at 2.4 TB/s).
template <int NV>
__global__ __launch_bounds__(256) void softmax_rows_reg_kernel(const float* in, half_t* out, long long rows, int N, int ldi, int ldo, float scale) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* src = in + (size_t)r * ldi;
  f32x4 v[NV];
  float m = -3.0e38f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = (k * 64 + lane) * 4;
    if (c < N) {
      v[k] = *(const f32x4*)(src + c);
      m = fmaxf(fmaxf(m, fmaxf(v[k][0], v[k][1])), fmaxf(v[k][2], v[k][3]));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  const float sc = scale * 1.4426950408889634f;
  const float msc = m * sc;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = (k * 64 + lane) * 4;
    if (c < N) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[k][e] = __builtin_amdgcn_exp2f(fmaf(v[k][e], sc, -msc)); s += v[k][e]; }
    }
  }
  s = 1.0f / ud_wave_sum(s);
  half_t* dst = out + (size_t)r * ldo;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = (k * 64 + lane) * 4;
    if (c < ldo) {
      half4 hv;
#pragma unroll
      for (int e = 0; e < 4; ++e) hv[e] = c < N ? (half_t)(v[k][e] * s) : (half_t)0.f;
      *(half4*)(dst + c) = hv;
    }
  }
  for (int c = (NV * 64 + lane) * 4; c < ldo; c += 256) *(half4*)(dst + c) = (half4){(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};    // K padding beyond the registers' span
}

__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* in, void* out, long long rows, int N, int ldi, int ldo, float scale, int out_f32) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* src = in + (size_t)r * ldi;
  float m = -3.0e38f;
  for (int c = lane; c < N; c += 64) m = fmaxf(m, src[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float s = 0.f;
  const float sc = scale * 1.4426950408889634f;
  for (int c = lane; c < N; c += 64) s += __builtin_amdgcn_exp2f((src[c] - m) * sc);
  s = 1.0f / ud_wave_sum(s);
  if (out_f32) {
    float* dst = (float*)out + (size_t)r * ldo;
    for (int c = lane; c < ldo; c += 64) dst[c] = c < N ? __builtin_amdgcn_exp2f((src[c] - m) * sc) * s : 0.f;
  } else {
    half_t* dst = (half_t*)out + (size_t)r * ldo;
    for (int c = lane; c < ldo; c += 64) dst[c] = c < N ? (half_t)(__builtin_amdgcn_exp2f((src[c] - m) * sc) * s) : (half_t)0.f;
  }
}

// ------------------------------------------------------------------------------------------------ few-query attention, one head of width D (camera head aggregate)
// q fp32 [B*T, D] (T <= 8 queries per image), kv fp16 [B*Nk, 2D] = [K | V], out fp32 [B*T, D].  Block = (query, image).
__global__ __launch_bounds__(256) void attention_fewq_kernel(const float* q, const half_t* kv, float* out, int T, int Nk, int D, float scale) {
  extern __shared__ float sm[];                         // Nk scores + D query values + 8 reduction slots
  float* sc = sm;
  float* qs = sm + Nk;
  float* red = qs + D;
  const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float* qv = q + ((size_t)b * T + t) * D;
  for (int d = tid; d < D; d += 256) qs[d] = qv[d] * scale;
  __syncthreads();
  const half_t* kb = kv + (size_t)b * Nk * 2 * D;
  float mx = -3.0e38f;
  for (int k = tid; k < Nk; k += 256) {
    const half8* kr = (const half8*)(kb + (size_t)k * 2 * D);
    float s = 0.f;
    for (int d8 = 0; d8 < (D >> 3); ++d8) {
      const half8 h = kr[d8];
#pragma unroll
      for (int e = 0; e < 8; ++e) s += (float)h[e] * qs[d8 * 8 + e];
    }
    sc[k] = s;
    mx = fmaxf(mx, s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int k = tid; k < Nk; k += 256) {
    const float p = __expf(sc[k] - mx);
    sc[k] = p;
    sum += p;
  }
  sum = ud_wave_sum(sum);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
  __syncthreads();
  const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));
  for (int d = tid; d < D; d += 256) {
    float acc = 0.f;
    const half_t* vcol = kb + D + d;
    for (int k = 0; k < Nk; ++k) acc += sc[k] * (float)vcol[(size_t)k * 2 * D];
    out[((size_t)b * T + t) * D + d] = acc * inv;
  }
}

// The same attention split over chunks of 64 keys (grid = chunks x images, all T queries per block): the one-block-per-query kernel
// above walks its 1200 keys with one key row per THREAD (2 KB apart: uncoalesced) and runs 64 blocks -- 842 us at bs = 16 for 39 MB
// of K|V.  Here a wave reads one key row per instruction (coalesced), every block leaves an un-normalised partial
// (max, sum, sum p V) in `part`, and fewq_merge_kernel combines the partials in ascending chunk order (deterministic).
constexpr int FQ_CHUNK = 64;
__global__ __launch_bounds__(256) void fewq_partial_kernel(const float* q, const half_t* kv, float* part, int T, int Nk, int D, float scale) {
  extern __shared__ float sm[];                         // T*D scaled queries | T x 64 scores / probabilities
  float* qs = sm;
  float* sc = sm + T * D;
  const int c = blockIdx.x, b = blockIdx.y, NC = gridDim.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < T * D; i += 256) qs[i] = q[(size_t)b * T * D + i] * scale;
  __syncthreads();
  const half_t* kb = kv + (size_t)b * Nk * 2 * D;
  for (int i = 0; i < 16; ++i) {
    const int kl = wv * 16 + i, k = c * FQ_CHUNK + kl;
    float s[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) s[t] = 0.f;
    if (k < Nk) {
      for (int d0 = lane * 8; d0 < D; d0 += 512) {
        const half8 h = *(const half8*)(kb + (size_t)k * 2 * D + d0);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          if (t < T) {
#pragma unroll
            for (int e = 0; e < 8; ++e) s[t] = fmaf((float)h[e], qs[t * D + d0 + e], s[t]);
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t < T) {
        const float v = ud_wave_sum(s[t]);
        if (lane == 0) sc[t * FQ_CHUNK + kl] = k < Nk ? v : -__builtin_inff();
      }
    }
  }
  __syncthreads();
  float* pb = part + ((size_t)b * NC + c) * T * (D + 2);
  for (int t = wv; t < T; t += 4) {
    const float v = sc[t * FQ_CHUNK + lane];
    float m = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    const float pexp = v == -__builtin_inff() ? 0.f : __expf(v - m);
    const float l = ud_wave_sum(pexp);
    sc[t * FQ_CHUNK + lane] = pexp;
    if (lane == 0) { pb[(size_t)t * (D + 2) + D] = m; pb[(size_t)t * (D + 2) + D + 1] = l; }
  }
  __syncthreads();
  const int kn = min(FQ_CHUNK, Nk - c * FQ_CHUNK);
  for (int d = tid; d < D; d += 256) {
    float acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = 0.f;
    const half_t* vcol = kb + (size_t)(c * FQ_CHUNK) * 2 * D + D + d;
    for (int k = 0; k < kn; ++k) {
      const float v = (float)vcol[(size_t)k * 2 * D];
#pragma unroll
      for (int t = 0; t < 8; ++t)
        if (t < T) acc[t] = fmaf(sc[t * FQ_CHUNK + k], v, acc[t]);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t)
      if (t < T) pb[(size_t)t * (D + 2) + d] = acc[t];
  }
}

__global__ __launch_bounds__(256) void fewq_merge_kernel(const float* part, float* out, int T, int NC, int D) {
  const int t = blockIdx.x, b = blockIdx.y;
  const float* pb = part + (size_t)b * NC * T * (D + 2) + (size_t)t * (D + 2);
  const size_t cs = (size_t)T * (D + 2);
  float M = -__builtin_inff();
  for (int c = 0; c < NC; ++c) M = fmaxf(M, pb[c * cs + D]);
  float L = 0.f;
  for (int c = 0; c < NC; ++c) L = fmaf(pb[c * cs + D + 1], __expf(pb[c * cs + D] - M), L);
  const float inv = 1.0f / L;
  for (int d = threadIdx.x; d < D; d += 256) {
    float acc = 0.f;
    for (int c = 0; c < NC; ++c) acc = fmaf(pb[c * cs + d], __expf(pb[c * cs + D] - M), acc);
    out[((size_t)b * T + t) * D + d] = acc * inv;
  }
}

// ------------------------------------------------------------------------------------------------ layers_8 / layers_4 attention
// What the reference's NystromBlock computes (layers/nystrom_attention.py:59-62,81 -> xformers NystromAttention, oracle/stubs/xformers): the module
// receives q, k, v as [b, n, h, d], reads `seq_len = k.size(-2)` = h, finds num_landmarks (128) >= seq_len and takes its small-sequence branch, a
// plain softmax attention over the LAST TWO axes -- every token's h head-vectors attend to each other:
//     att[b, n, i, j] = softmax_j((q[b, n, i, :] / sqrt(d)) . k[b, n, j, :]),   out[b, n, i, :] = sum_j att[b, n, i, j] v[b, n, j, :].
// No landmarks, no pseudo-inverse, nothing crosses tokens.  One wave per token, lane = channel inside a head (d = 64); the NH x NH scores are wave
// reductions, everything fp32, fp16 result (the A operand of the output projection).  HBM-bound: (3 NH 64) fp32 in + (NH 64) fp16 out per token.
template <int NH>
__global__ __launch_bounds__(256) void head_mix_kernel(const float* q, const float* kv, half_t* out, int M, int ldq, int ldkv, int ldo, float scale) {
  const int lane = threadIdx.x & 63;
  const int nw = gridDim.x * 4;
  for (int tok = blockIdx.x * 4 + (threadIdx.x >> 6); tok < M; tok += nw) {
    const float* qp = q + (size_t)tok * ldq + lane;
    const float* kp = kv + (size_t)tok * ldkv + lane;
    float qv[NH], kk[NH], vv[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      qv[i] = qp[i * 64] * scale;                     // xformers core.py scaled_query_key_softmax: q / sqrt(d) before the product
      kk[i] = kp[i * 64];
      vv[i] = kp[(NH + i) * 64];
    }
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      float sc[NH], mx = -__builtin_inff();
#pragma unroll
      for (int j = 0; j < NH; ++j) {
        sc[j] = ud_wave_sum(qv[i] * kk[j]);
        mx = fmaxf(mx, sc[j]);
      }
      float den = 0.f, acc = 0.f;
#pragma unroll
      for (int j = 0; j < NH; ++j) {
        const float e = __expf(sc[j] - mx);
        den += e;
        acc = fmaf(e, vv[j], acc);
      }
      out[(size_t)tok * ldo + i * 64 + lane] = (half_t)(acc / den);
    }
  }
}

// ------------------------------------------------------------------------------------------------ glue
__global__ __launch_bounds__(256) void add_kernel(float* dst, const float* a, const float* b, long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) ((f32x4*)dst)[i] = ((const f32x4*)a)[i] + ((const f32x4*)b)[i];
}
// dst[(img * rows_per_img + row_off + t) * ld + d] = src[(img * T + t) * D + d]; cast: 0 fp32 -> fp32, 1 fp32 -> fp16, 2 fp16 -> fp16 transposed pack of T^T
__global__ __launch_bounds__(256) void copy_rows_kernel(void* dst, const float* src, int n_img, int T, int rows_per_img, int row_off, int D, int ld, int to_f16) {
  const long long total = (long long)n_img * T * D;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int d = (int)(idx % D);
    const long long r = idx / D;
    const int t = (int)(r % T), img = (int)(r / T);
    const size_t o = ((size_t)img * rows_per_img + row_off + t) * ld + d;
    if (to_f16 == 2) {              // two fp16 terms side by side (ld >= 2 D): hi at column d, lo = fp16(x - hi) at column D + d
      const float x = src[idx];
      const half_t hi = (half_t)x;
      ((half_t*)dst)[o] = hi;
      ((half_t*)dst)[o + D] = (half_t)(x - (float)hi);
    } else if (to_f16) ((half_t*)dst)[o] = (half_t)src[idx];
    else ((float*)dst)[o] = src[idx];
  }
}
// nn.UpsamplingBilinear2d (align_corners=True; layers/upsample.py:34) of an fp32 NHWC map, written as the TWO-TERM fp16 A operand of the 3x3
// convolution behind it: out[pix][0 .. C) = hi = fp16(v), out[pix][C .. 2C) = lo = fp16(v - hi) -- with weights [W_hi | W_hi | W_lo] per tap and
// the channel index wrapping after 2 C (UdGemm.a_wrap) the product is A_hi W_hi + A_lo W_hi + A_hi W_lo: the activation's fp16 rounding
// (2^-11 relative, the largest single error source of the V1 depth stack: DESIGN 10.3) is gone, interpolation stays fp32 throughout.
// Same source-coordinate arithmetic as resize_ac_kernel (pointwise.hip).  4 channels per thread.
__global__ __launch_bounds__(256) void resize_ac_split_kernel(const float* in, half_t* out, int B, int Hin, int Win, int Hout, int Wout, int C) {
  const int CG = C >> 2;
  const float sy = Hout > 1 ? (float)(Hin - 1) / (float)(Hout - 1) : 0.f;
  const float sx = Wout > 1 ? (float)(Win - 1) / (float)(Wout - 1) : 0.f;
  const long long total = (long long)B * Hout * Wout * CG;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int cg = (int)(idx % CG);
    const long long pix = idx / CG;
    const int ox = (int)(pix % Wout);
    const long long r = pix / Wout;
    const int oy = (int)(r % Hout), b = (int)(r / Hout);
    const float fy = sy * (float)oy, fx = sx * (float)ox;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hin - 1), x1 = x0 + (x0 < Win - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float* ip = in + (size_t)b * Hin * Win * C + cg * 4;
    const f32x4 v00 = *(const f32x4*)(ip + ((size_t)y0 * Win + x0) * C), v01 = *(const f32x4*)(ip + ((size_t)y0 * Win + x1) * C);
    const f32x4 v10 = *(const f32x4*)(ip + ((size_t)y1 * Win + x0) * C), v11 = *(const f32x4*)(ip + ((size_t)y1 * Win + x1) * C);
    half4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float top = (1.0f - lx) * v00[e] + lx * v01[e];
      const float bot = (1.0f - lx) * v10[e] + lx * v11[e];
      const float v = (1.0f - ly) * top + ly * bot;
      hi[e] = (half_t)v;
      lo[e] = (half_t)(v - (float)hi[e]);
    }
    half_t* op = out + (size_t)pix * 2 * C + cg * 4;
    *(half4*)op = hi;
    *(half4*)(op + C) = lo;
  }
}
// camera tail of V1 (decoder.py:85-99 CameraHead, :347-353 run_camera; unidepthv1.py:88-92 _postprocess): raw [B*4] ->
// K33 at network resolution, its closed-form inverse, and the post-processed matrix ((fx, fy) / ratio, (cx - pad_l, cy - pad_t) / ratio).
__global__ void camera_v1_kernel(const float* raw, float* K33, float* Kinv33, float* Kpost33, int B, int Hn, int Wn, float ratio, int pad_l, int pad_t) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const float* o = raw + (size_t)b * 4;
  const float half = 0.5f * (float)(Hn > Wn ? Hn : Wn);
  const float fx = expf(o[0]) * half, fy = expf(o[1]) * half;
  const float cx = (1.0f / (1.0f + expf(-o[2]))) * (float)Wn, cy = (1.0f / (1.0f + expf(-o[3]))) * (float)Hn;
  float* K = K33 + b * 9;
  K[0] = fx; K[1] = 0.f; K[2] = cx; K[3] = 0.f; K[4] = fy; K[5] = cy; K[6] = 0.f; K[7] = 0.f; K[8] = 1.f;
  float* Ki = Kinv33 + b * 9;
  Ki[0] = 1.0f / fx; Ki[1] = 0.f; Ki[2] = -cx / fx; Ki[3] = 0.f; Ki[4] = 1.0f / fy; Ki[5] = -cy / fy; Ki[6] = 0.f; Ki[7] = 0.f; Ki[8] = 1.f;
  float* Kp = Kpost33 + b * 9;
  Kp[0] = fx / ratio; Kp[1] = 0.f; Kp[2] = (cx - (float)pad_l) / ratio; Kp[3] = 0.f; Kp[4] = fy / ratio; Kp[5] = (cy - (float)pad_t) / ratio;
  Kp[6] = 0.f; Kp[7] = 0.f; Kp[8] = 1.f;
}

// final assembly (unidepthv1.py:353-371): depth z [B,H,W] (column 0 of an [.., ldz] map) + intrinsics K33 -> points [B,3,H,W], depth [B,1,H,W]:
// ray through the pixel centre, theta = atan2(x, z), phi = acos(y) (utils/geometric.py:45-51), x = z tan(theta), y = z / tan(phi) / cos(theta) (:55-73)
__global__ __launch_bounds__(256) void v1_points_kernel(const float* zmap, const float* K33, float* points, float* depth, int B, int H, int W, int ldz, int nK) {
  const long long total = (long long)B * H * W;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int u = (int)(idx % W), v = (int)((idx / W) % H), b = (int)(idx / ((long long)W * H));
    const float* K = K33 + (nK == 1 ? 0 : b) * 9;
    float x = ((float)u + 0.5f - K[2]) / K[0], y = ((float)v + 0.5f - K[5]) / K[4], zz = 1.0f;
    const float inv = 1.0f / fmaxf(sqrtf(x * x + y * y + 1.0f), 1e-12f);
    x *= inv; y *= inv; zz *= inv;
    const float theta = atan2f(x, zz), phi = acosf(y);
    const float z = zmap[(size_t)idx * ldz];
    const size_t HW = (size_t)H * W, po = (size_t)v * W + u;
    points[((size_t)b * 3 + 0) * HW + po] = z * tanf(theta);
    points[((size_t)b * 3 + 1) * HW + po] = z / tanf(phi) / cosf(theta);
    points[((size_t)b * 3 + 2) * HW + po] = z;
    depth[(size_t)b * HW + po] = z;
  }
}

// mean of three [B,H,W,4]-strided maps' column 0 -> [B,H,W,4] column 0 (unidepthv1.py:66-77: the multi-scale predictions are averaged after resizing)
__global__ __launch_bounds__(256) void mean3_kernel(const float* a, const float* b, const float* c, float* out, long long n, int ld) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    out[i * ld] = (a[i * ld] + b[i * ld] + c[i * ld]) * (1.0f / 3.0f);
}

// network image of V1 (unidepthv1.py:305-321): /255 if uint8 (the caller decides the float cases), ImageNet normalisation, antialiased bilinear
// resize to (h, w), zero padding to (Hn, Wn) -> fp32 NCHW.  Thread = one output value.
__global__ __launch_bounds__(256) void preprocess_v1_kernel(const void* rgb, float* out, int B, int H, int W, int h, int w, int Hn, int Wn, int pad_l, int pad_t,
                                                            int is_u8, int div255, int normalize, f32x4 mean, f32x4 istd) {
  const long long total = (long long)B * 3 * Hn * Wn;
  const float sy = (float)H / (float)h, sx = (float)W / (float)w;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int ox = (int)(idx % Wn), oy = (int)((idx / Wn) % Hn), c = (int)((idx / ((long long)Wn * Hn)) % 3), b = (int)(idx / ((long long)3 * Wn * Hn));
    const int yy = oy - pad_t, xx = ox - pad_l;
    float v = 0.f;
    if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) {
      int ylo, yn, xlo, xn;
      float yc, yi, xc, xi;
      aa_taps(yy, sy, H, ylo, yn, yc, yi);
      aa_taps(xx, sx, W, xlo, xn, xc, xi);
      float wys = 0.f, wxs = 0.f;
      for (int j = 0; j < yn; ++j) wys += aa_w(j, ylo, yc, yi);
      for (int j = 0; j < xn; ++j) wxs += aa_w(j, xlo, xc, xi);
      const size_t base = ((size_t)b * 3 + c) * H * W;
      for (int jy = 0; jy < yn; ++jy) {
        float row = 0.f;
        for (int jx = 0; jx < xn; ++jx) {
          const size_t o = base + (size_t)(ylo + jy) * W + xlo + jx;
          float p = is_u8 ? (float)((const unsigned char*)rgb)[o] : ((const float*)rgb)[o];
          if (div255) p *= (1.0f / 255.0f);
          if (normalize) p = (p - mean[c]) * istd[c];
          row += (aa_w(jx, xlo, xc, xi) / wxs) * p;
        }
        v += (aa_w(jy, ylo, yc, yi) / wys) * row;
      }
    }
    out[idx] = v;
  }
}

inline unsigned grid1(long long total, int cap = 256 * 32) {
  long long g = (total + 255) / 256;
  if (g > cap) g = cap;
  return (unsigned)(g < 1 ? 1 : g);
}


// 3x3 convolution to ONE output channel, zero padding, then exp(clamp(., -10, 10)): the multi-scale outputs out8 / out4 / out2 of UniDepthV1
// (unidepthv1/decoder.py:185-187 nn.Conv2d(C, 1, 3, padding=1), :250-298).  With one output channel there is nothing for a matrix tile to do (the
// MFMA form ran three launches at 14 TFLOP/s-equivalent on 32-column tiles, behind a pass that split the fp32 maps into fp16 [hi | lo] pairs):
// this is a stencil over the fp32 map itself -- exact fp32 products, HBM-bound (every input pixel is read ~1.3 times).
// A workgroup owns 8 x 32 output pixels (one per thread); per chunk of 32 channels the 10 x 34 halo tile is staged in LDS (128 B per pixel,
// rows padded by 16 B so the b128 reads of 8 neighbouring pixels cover the 32 banks) beside the chunk's 9 x 32 weights.
constexpr int OC_TH = 8, OC_TW = 32, OC_CC = 32;
constexpr int OC_HW = OC_TW + 2;
constexpr int OC_HP = (OC_TH + 2) * OC_HW;
constexpr int OC_LD = OC_CC + 4;
__global__ __launch_bounds__(256) void out_conv3_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ out, int H, int W, int C, int ldo,
                                                         float bias, int tiles_x, int tiles_y) {
  __shared__ __attribute__((aligned(16))) float xs[OC_HP * OC_LD];
  __shared__ __attribute__((aligned(16))) float ws[9 * OC_CC];
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int b = t / tiles_y;
  const int y0 = ty * OC_TH, x0 = tx * OC_TW;
  const int py = tid >> 5, px = tid & 31;
  const float* xb = x + (size_t)b * H * W * C;
  float acc = 0.f;
  for (int c0 = 0; c0 < C; c0 += OC_CC) {
    __syncthreads();                                           // the previous chunk has been consumed
    for (int idx = tid; idx < OC_HP * 8; idx += 256) {
      const int pix = idx >> 3, piece = idx & 7;
      const int hy = pix / OC_HW, hx = pix - hy * OC_HW;
      const int gy = y0 + hy - 1, gx = x0 + hx - 1;
      const int c = c0 + piece * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (gy >= 0 && gy < H && gx >= 0 && gx < W && c < C) v = *(const f32x4*)(xb + ((size_t)gy * W + gx) * C + c);
      *(f32x4*)(xs + pix * OC_LD + piece * 4) = v;
    }
    if (tid < 72) {
      const int tap = tid >> 3, piece = tid & 7;
      const int c = c0 + piece * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (c < C) v = *(const f32x4*)(w + (size_t)tap * C + c);
      *(f32x4*)(ws + tap * OC_CC + piece * 4) = v;
    }
    __syncthreads();
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const float* xp = xs + ((py + dy) * OC_HW + px + dx) * OC_LD;
        const float* wp = ws + (dy * 3 + dx) * OC_CC;
#pragma unroll
        for (int c4 = 0; c4 < OC_CC / 4; ++c4) {
          const f32x4 xv = *(const f32x4*)(xp + c4 * 4);
          const f32x4 wv = *(const f32x4*)(wp + c4 * 4);
          acc = fmaf(xv[0], wv[0], acc); acc = fmaf(xv[1], wv[1], acc); acc = fmaf(xv[2], wv[2], acc); acc = fmaf(xv[3], wv[3], acc);
        }
      }
  }
  const int gy = y0 + py, gx = x0 + px;
  if (gy < H && gx < W) out[((size_t)b * H * W + (size_t)gy * W + gx) * ldo] = ud_clampexp(acc + bias);
}

}  // namespace

// UniDepthV1 on a DINOv2 backbone: what the decoder consumes of block i is max over the blocks of its level of (patch tokens + class token)
// (unidepthv1.py:324-328 adds the class token to every block's patch tokens, decoder.py:366-373 max_stack) and the raw class tokens of the last
// four blocks.  x fp32 [B*Np, D], row 0 of an image = class token, rows 1..hw = patches; smax fp32 [B*hw, D]; cls fp32 [B, D] or NULL.
__global__ __launch_bounds__(256) void vit_tap_kernel(const float* __restrict__ x, float* __restrict__ smax, float* __restrict__ cls, int Np, int hw, int D4, int init) {
  const int b = blockIdx.y;
  const f32x4* xb = (const f32x4*)x + (size_t)b * Np * D4;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < (long long)hw * D4; idx += (long long)gridDim.x * 256) {
    const int t = (int)(idx / D4), c = (int)(idx - (long long)t * D4);
    const f32x4 v = xb[(size_t)(1 + t) * D4 + c] + xb[c];
    f32x4* dst = (f32x4*)smax + ((size_t)b * hw + t) * D4 + c;
    if (init) {
      *dst = v;
    } else {
      const f32x4 o = *dst;
      f32x4 m;
#pragma unroll
      for (int r = 0; r < 4; ++r) m[r] = fmaxf(o[r], v[r]);
      *dst = m;
    }
    if (cls && t == 0) ((f32x4*)cls)[(size_t)b * D4 + c] = xb[c];
  }
}

extern "C" int ud_v1_op(const UdV1Op* desc, void* stream) {
  const UdV1Op& d = *desc;
  hipStream_t s = (hipStream_t)stream;
  const int* i = d.i;
  switch (d.kind) {
    case UD_V1_RESIZE_AA: {      // a = in fp32 NHWC, out fp32 NHWC; i = B, Hi, Wi, Ho, Wo, C, ldi, ldo, y0, x0, Hc, Wc
      if (!d.a || !d.out || i[0] <= 0 || (i[5] & 3) || (i[6] & 3) || (i[7] & 3) || i[10] <= 0 || i[11] <= 0 || i[8] + i[10] > i[1] || i[9] + i[11] > i[2]) break;
      hipLaunchKernelGGL(resize_aa_kernel, dim3(grid1((long long)i[0] * i[3] * i[4] * (i[5] >> 2))), dim3(256), 0, s, (const float*)d.a, (float*)d.out, i[0], i[1], i[2],
                         i[3], i[4], i[5], i[6], i[7], i[8], i[9], i[10], i[11]);
      UD_CHECK_LAUNCH("ud_v1_op(resize_aa) launch");
      return UD_OK;
    }
    case UD_V1_SH_EMBED: {       // a = rays fp32 [nb,3,Hn,Wn], out fp16 [nb*rows_per_img, ldo]; i = nb, Hn, Wn, h, w, ldo, rows_per_img; f[0] = eps
      if (!d.a || !d.out || i[0] <= 0 || i[5] < 81 || i[3] <= 0 || i[4] <= 0) break;
      const long long ntok = (long long)i[0] * i[3] * i[4];
      // tokens per wave: 64 where there are enough tokens to fill the chip with such waves (phase 2 at full lane use), fewer at the coarse
      // levels, whose footprints are large (phase 1 dominates) and whose token counts are small (measured with 64 everywhere at bs 16:
      // 470 us at the 1/16 level -- 300 waves walking 64 x 1024 taps each -- against 71 us for one token per wave)
      int tpw = 64;
      while (tpw > 1 && ntok / tpw < 4096) tpw >>= 1;
      const bool own = (long long)i[1] * i[2] <= 16LL * i[3] * i[4];      // footprint <= ~8 x 8 taps: every lane averages its own
      const int ntw = own ? 64 : tpw;
      if (own) tpw = -1;
      hipLaunchKernelGGL(sh_embed_kernel, dim3((unsigned)((ntok + 4 * ntw - 1) / (4 * ntw))), dim3(256), 0, s, (const float*)d.a, (half_t*)d.out, i[0], i[1], i[2], i[3], i[4], i[5],
                         i[6], d.f[0], tpw);
      UD_CHECK_LAUNCH("ud_v1_op(sh_embed) launch");
      return UD_OK;
    }
    case UD_V1_SOFTMAX: {        // a = scores fp32, out = fp16 (i[4] = 0) or fp32 (1); i = rows_lo, N, ldi, ldo, out_f32, rows_hi; f[0] = scale
      const long long rows = ((long long)i[5] << 31) + i[0];
      if (!d.a || !d.out || rows <= 0 || i[1] <= 0 || i[3] < i[1]) break;
      if (!i[4] && !(i[1] & 3) && !(i[2] & 3) && !(i[3] & 3) && i[1] > 256 && i[1] <= 5120 && !((size_t)d.a & 15) && !((size_t)d.out & 7)) {
        if (i[1] <= 1280) hipLaunchKernelGGL((softmax_rows_reg_kernel<5>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, (const float*)d.a, (half_t*)d.out, rows, i[1], i[2], i[3], d.f[0]);
        else hipLaunchKernelGGL((softmax_rows_reg_kernel<20>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, (const float*)d.a, (half_t*)d.out, rows, i[1], i[2], i[3], d.f[0]);
        UD_CHECK_LAUNCH("ud_v1_op(softmax, register rows) launch");
        return UD_OK;
      }
      hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, (const float*)d.a, d.out, rows, i[1], i[2], i[3], d.f[0], i[4]);
      UD_CHECK_LAUNCH("ud_v1_op(softmax) launch");
      return UD_OK;
    }
    case UD_V1_ATTN_FEWQ: {      // a = q fp32 [B*T, D], b = kv fp16 [B*Nk, 2D], out fp32 [B*T, D]; i = B, T, Nk, D; f[0] = scale
      if (!d.a || !d.b || !d.out || i[0] <= 0 || i[1] <= 0 || i[1] > 8 || i[2] <= 0 || (i[3] & 7)) break;
      if (d.c) {                 // c = scratch fp32 [B * ceil(Nk / 64) * T * (D + 2)]: key-chunked partials + merge
        const int nc = (i[2] + FQ_CHUNK - 1) / FQ_CHUNK;
        const int lds2 = (i[1] * i[3] + i[1] * FQ_CHUNK) * 4;
        if (lds2 > 64 * 1024) break;
        hipLaunchKernelGGL(fewq_partial_kernel, dim3(nc, i[0]), dim3(256), lds2, s, (const float*)d.a, (const half_t*)d.b, (float*)d.c, i[1], i[2], i[3], d.f[0]);
        hipLaunchKernelGGL(fewq_merge_kernel, dim3(i[1], i[0]), dim3(256), 0, s, (const float*)d.c, (float*)d.out, i[1], nc, i[3]);
        UD_CHECK_LAUNCH("ud_v1_op(attn_fewq, chunked) launch");
        return UD_OK;
      }
      const int lds = (i[2] + i[3] + 8) * 4;
      if (lds > 64 * 1024) break;
      hipLaunchKernelGGL(attention_fewq_kernel, dim3(i[1], i[0]), dim3(256), lds, s, (const float*)d.a, (const half_t*)d.b, (float*)d.out, i[1], i[2], i[3], d.f[0]);
      UD_CHECK_LAUNCH("ud_v1_op(attn_fewq) launch");
      return UD_OK;
    }
    case UD_V1_HEAD_MIX: {       // a = q fp32 [M, ldq], b = [K | V] fp32 [M, ldkv], out fp16 [M, ldo]; i = M, heads (2 / 4 / 8 of width 64), ldq, ldkv, ldo; f[0] = scale
      if (!d.a || !d.b || !d.out || i[0] <= 0 || (i[1] != 2 && i[1] != 4 && i[1] != 8) || i[2] < i[1] * 64 || i[3] < 2 * i[1] * 64 || i[4] < i[1] * 64) break;
      const dim3 grid((unsigned)((i[0] + 3) / 4 < 8192 ? (i[0] + 3) / 4 : 8192));
      if (i[1] == 2) hipLaunchKernelGGL(head_mix_kernel<2>, grid, dim3(256), 0, s, (const float*)d.a, (const float*)d.b, (half_t*)d.out, i[0], i[2], i[3], i[4], d.f[0]);
      else if (i[1] == 4) hipLaunchKernelGGL(head_mix_kernel<4>, grid, dim3(256), 0, s, (const float*)d.a, (const float*)d.b, (half_t*)d.out, i[0], i[2], i[3], i[4], d.f[0]);
      else hipLaunchKernelGGL(head_mix_kernel<8>, grid, dim3(256), 0, s, (const float*)d.a, (const float*)d.b, (half_t*)d.out, i[0], i[2], i[3], i[4], d.f[0]);
      UD_CHECK_LAUNCH("ud_v1_op(head_mix) launch");
      return UD_OK;
    }
    case UD_V1_ADD: {            // out = a + b, fp32, i[0] + (i[1] << 31) elements (% 4 == 0)
      const long long n = ((long long)i[1] << 31) + i[0];
      if (!d.a || !d.b || !d.out || n <= 0 || (n & 3)) break;
      hipLaunchKernelGGL(add_kernel, dim3(grid1(n / 4)), dim3(256), 0, s, (float*)d.out, (const float*)d.a, (const float*)d.b, n / 4);
      UD_CHECK_LAUNCH("ud_v1_op(add) launch");
      return UD_OK;
    }
    case UD_V1_VIT_TAP: {        // a = x fp32 [B*Np, D]; out = smax fp32 [B*hw, D]; out2 = cls fp32 [B, D] or NULL; i = B, Np, hw, D, init
      if (!d.a || !d.out || i[0] <= 0 || i[2] <= 0 || i[1] <= i[2] || (i[3] & 3)) break;
      const long long n = (long long)i[2] * (i[3] / 4);
      hipLaunchKernelGGL(vit_tap_kernel, dim3((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048), i[0]), dim3(256), 0, s, (const float*)d.a, (float*)d.out, (float*)d.out2,
                         i[1], i[2], i[3] / 4, i[4]);
      UD_CHECK_LAUNCH("ud_v1_op(vit_tap) launch");
      return UD_OK;
    }
    case UD_V1_RESIZE_AC_SPLIT: {   // a = fp32 NHWC [B, Hin, Win, C]; out = fp16 [B, Hout, Wout, 2 C] (hi | lo); i = B, Hin, Win, Hout, Wout, C
      if (!d.a || !d.out || i[0] <= 0 || i[1] <= 0 || i[2] <= 0 || i[3] <= 0 || i[4] <= 0 || i[5] <= 0 || (i[5] & 3)) break;
      hipLaunchKernelGGL(resize_ac_split_kernel, dim3(grid1((long long)i[0] * i[3] * i[4] * (i[5] >> 2))), dim3(256), 0, s, (const float*)d.a, (half_t*)d.out, i[0], i[1], i[2],
                         i[3], i[4], i[5]);
      UD_CHECK_LAUNCH("ud_v1_op(resize_ac_split) launch");
      return UD_OK;
    }
    case UD_V1_OUT_CONV3: {      // a = x fp32 NHWC [B, H, W, C]; b = w fp32 [9, C] (tap = ky * 3 + kx); out fp32 [B*H*W, ldo] (column 0 written); i = B, H, W, C, ldo; f[0] = bias
      if (!d.a || !d.b || !d.out || i[0] <= 0 || i[1] <= 0 || i[2] <= 0 || i[3] <= 0 || (i[3] & 3) || i[4] <= 0) break;
      const int tiles_x = (i[2] + OC_TW - 1) / OC_TW, tiles_y = (i[1] + OC_TH - 1) / OC_TH;
      if ((long long)i[0] * tiles_x * tiles_y > 0x7fffffffLL) break;
      hipLaunchKernelGGL(out_conv3_kernel, dim3(i[0] * tiles_x * tiles_y), dim3(256), 0, s, (const float*)d.a, (const float*)d.b, (float*)d.out, i[1], i[2], i[3], i[4],
                         d.f[0], tiles_x, tiles_y);
      UD_CHECK_LAUNCH("ud_v1_op(out_conv3) launch");
      return UD_OK;
    }
    case UD_V1_COPY_ROWS: {      // a = src fp32 [n_img*T, D]; out rows (img*rows_per_img + row_off + t), stride ld; i = n_img, T, rows_per_img, row_off, D, ld, to_f16
      if (!d.a || !d.out || i[0] <= 0 || i[1] <= 0 || i[4] <= 0 || (i[6] == 2 && i[5] < 2 * i[4])) break;
      hipLaunchKernelGGL(copy_rows_kernel, dim3(grid1((long long)i[0] * i[1] * i[4])), dim3(256), 0, s, d.out, (const float*)d.a, i[0], i[1], i[2], i[3], i[4], i[5], i[6]);
      UD_CHECK_LAUNCH("ud_v1_op(copy_rows) launch");
      return UD_OK;
    }
    case UD_V1_CAMERA: {         // a = raw fp32 [B*4]; out = K33, out2 = Kinv33, c = Kpost33 (written); i = B, Hn, Wn, pad_l, pad_t; f[0] = ratio
      if (!d.a || !d.out || !d.out2 || !d.c || i[0] <= 0 || !(d.f[0] > 0.f)) break;
      hipLaunchKernelGGL(camera_v1_kernel, dim3((i[0] + 63) / 64), dim3(64), 0, s, (const float*)d.a, (float*)d.out, (float*)d.out2, (float*)d.c, i[0], i[1], i[2], d.f[0], i[3], i[4]);
      UD_CHECK_LAUNCH("ud_v1_op(camera) launch");
      return UD_OK;
    }
    case UD_V1_POINTS: {         // a = z map fp32 (stride ldz), b = K33 fp32 [nK, 9]; out = points [B,3,H,W], out2 = depth [B,1,H,W]; i = B, H, W, ldz, nK
      if (!d.a || !d.b || !d.out || !d.out2 || i[0] <= 0) break;
      hipLaunchKernelGGL(v1_points_kernel, dim3(grid1((long long)i[0] * i[1] * i[2])), dim3(256), 0, s, (const float*)d.a, (const float*)d.b, (float*)d.out, (float*)d.out2, i[0], i[1], i[2], i[3], i[4]);
      UD_CHECK_LAUNCH("ud_v1_op(points) launch");
      return UD_OK;
    }
    case UD_V1_MEAN3: {          // a, b, c fp32 strided maps -> out; i = n_lo, ld, n_hi
      const long long n = ((long long)i[2] << 31) + i[0];
      if (!d.a || !d.b || !d.c || !d.out || n <= 0) break;
      hipLaunchKernelGGL(mean3_kernel, dim3(grid1(n)), dim3(256), 0, s, (const float*)d.a, (const float*)d.b, (const float*)d.c, (float*)d.out, n, i[1]);
      UD_CHECK_LAUNCH("ud_v1_op(mean3) launch");
      return UD_OK;
    }
    case UD_V1_PREPROCESS: {     // a = rgb (u8 or fp32 NCHW), out fp32 NCHW [B,3,Hn,Wn]; i = B, H, W, h, w, Hn, Wn, pad_l, pad_t, is_u8, div255, normalize; f = unused
      if (!d.a || !d.out || i[0] <= 0 || i[3] <= 0 || i[4] <= 0 || i[3] + i[8] > i[5] || i[4] + i[7] > i[6]) break;
      const f32x4 mean = (f32x4){0.485f, 0.456f, 0.406f, 0.f};
      const f32x4 istd = (f32x4){1.0f / 0.229f, 1.0f / 0.224f, 1.0f / 0.225f, 0.f};
      hipLaunchKernelGGL(preprocess_v1_kernel, dim3(grid1((long long)i[0] * 3 * i[5] * i[6])), dim3(256), 0, s, d.a, (float*)d.out, i[0], i[1], i[2], i[3], i[4], i[5], i[6],
                         i[7], i[8], i[9], i[10], i[11], mean, istd);
      UD_CHECK_LAUNCH("ud_v1_op(preprocess) launch");
      return UD_OK;
    }
    default:
      ud_set_error("ud_v1_op: unknown kind");
      return UD_ERR_UNSUPPORTED;
  }
  ud_set_error("ud_v1_op: bad argument");
  return UD_ERR_BAD_ARG;
}
