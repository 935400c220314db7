// fp32 island for the camera head (see UdLinearF32 in include/unidepth_hip.h for why): 4 tokens per image, ~9 M weights,
// latency-bound; plain fp32 FMAs are the right tool (no MFMA: at M = 4*B rows the matrix pipe would idle anyway).
#include "ud_common.h"

namespace {

// one wave per output column n; the 64 lanes split K (16 B per lane per step, coalesced, branch-free), every lane carries
// partial sums for up to 32 rows; wave reductions at the end.  W is streamed exactly once, x (<= 256 KB) comes from L1/L2.
// (Measured alternatives on MI355X -- K split over several waves per column, reduce-scatter shuffles -- were slower: every
// extra wave re-reads the activation slice and the kernel is bound by that L2 traffic, not by the shuffles.)
__global__ __launch_bounds__(256) void linear_f32_kernel(const UdLinearF32 p) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (n >= p.N) return;
  const int m0 = blockIdx.y * 32;
  const int rows = (p.M - m0) < 32 ? (p.M - m0) : 32;
  float acc[32];
#pragma unroll
  for (int m = 0; m < 32; ++m) acc[m] = 0.f;
  const float* wr = p.W + (size_t)n * p.ldw;
  const float* xb = p.x + (size_t)m0 * p.ldx;
  for (int k = lane * 4; k < p.K; k += 256) {
    const f32x4 wv = *(const f32x4*)(wr + k);
    f32x4 xv[32];
#pragma unroll
    for (int m = 0; m < 32; ++m) {          // rows beyond the batch re-read the last valid row (results unused)
      const int mr = m < rows ? m : rows - 1;
      xv[m] = *(const f32x4*)(xb + (size_t)mr * p.ldx + k);
    }
#pragma unroll
    for (int m = 0; m < 32; ++m)
      acc[m] = fmaf(xv[m][0], wv[0], fmaf(xv[m][1], wv[1], fmaf(xv[m][2], wv[2], fmaf(xv[m][3], wv[3], acc[m]))));
  }
  float mine = 0.f;   // lane m keeps the reduced value of row m
#pragma unroll
  for (int m = 0; m < 32; ++m) {
    const float s = ud_wave_sum(acc[m]);
    if (lane == m) mine = s;
  }
  if (lane < rows) {
    const int m = m0 + lane;
    float y = mine;
    if (p.bias) y += p.bias[n];
    if (p.add) y += p.add[(size_t)(m % p.add_mod) * p.ldadd + n];
    if (p.act == UD_ACT_GELU) y = 0.5f * y * (1.0f + erff(y * 0.70710678118654752440f));
    float* o = p.out + (size_t)m * p.ldc + n;
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

}  // namespace

extern "C" int ud_linear_f32(const UdLinearF32* desc, void* stream) {
  const UdLinearF32& d = *desc;
  if (!d.x || !d.W || !d.out || d.M <= 0 || d.N <= 0 || d.K <= 0 || (d.K & 3) || (d.ldx & 3) || (d.ldw & 3) || (d.add && d.add_mod <= 0)) {
    ud_set_error("ud_linear_f32: bad argument (K, ldx, ldw % 4 == 0)");
    return UD_ERR_BAD_ARG;
  }
  hipLaunchKernelGGL(linear_f32_kernel, dim3((d.N + 3) / 4, (d.M + 31) / 32), dim3(256), 0, (hipStream_t)stream, d);
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
