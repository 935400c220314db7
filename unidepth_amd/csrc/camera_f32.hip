// fp32 island for the camera head (see UdLinearF32 in include/unidepth_hip.h for why): 4 tokens per image, ~9 M weights,
// latency-bound; plain fp32 FMAs are the right tool (no MFMA: at M = 4*B rows the matrix pipe would idle anyway).
#include "ud_common.h"

namespace {

// one wave per output column n, one lane per row m (<= 64 rows per grid.y slice): W[n, :] is wave-uniform (broadcast
// loads), x[m, :] streams per lane from L1/L2 (the whole activation is <= 256 KB).
__global__ __launch_bounds__(256) void linear_f32_kernel(const UdLinearF32 p) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (n >= p.N) return;
  const int m = blockIdx.y * 64 + lane;
  const int mm = m < p.M ? m : p.M - 1;
  const f32x4* xr = (const f32x4*)(p.x + (size_t)mm * p.ldx);
  const f32x4* wr = (const f32x4*)(p.W + (size_t)n * p.ldw);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const int k4 = p.K >> 2;
#pragma unroll 8
  for (int k = 0; k < k4; ++k) {
    const f32x4 xv = xr[k];
    const f32x4 wv = wr[k];
    a0 = fmaf(xv[0], wv[0], a0);
    a1 = fmaf(xv[1], wv[1], a1);
    a2 = fmaf(xv[2], wv[2], a2);
    a3 = fmaf(xv[3], wv[3], a3);
  }
  float y = (a0 + a1) + (a2 + a3);
  if (p.bias) y += p.bias[n];
  if (p.add) y += p.add[(size_t)(mm % p.add_mod) * p.ldadd + n];
  if (p.act == UD_ACT_GELU) y = 0.5f * y * (1.0f + erff(y * 0.70710678118654752440f));
  if (m < p.M) {
    float* o = p.out + (size_t)m * p.ldc + n;
    *o = p.accumulate ? *o + y : y;
  }
}

__global__ void attention_small_kernel(const float* q, const float* kv, float* out, int B, int T, int H, int C, float scale) {
  const int idx = blockIdx.x * 64 + threadIdx.x;
  if (idx >= B * H * T) return;
  const int i = idx % T;
  const int h = (idx / T) % H;
  const int b = idx / (T * H);
  const int hd = C / H;
  const float* qi = q + (size_t)(b * T + i) * C + h * hd;
  float s[8];
  float mx = -1e30f;
  for (int j = 0; j < T; ++j) {
    const float* kj = kv + (size_t)(b * T + j) * 2 * C + h * hd;
    float acc = 0.f;
    for (int d = 0; d < hd; ++d) acc = fmaf(qi[d], kj[d], acc);
    s[j] = acc * scale;
    mx = fmaxf(mx, s[j]);
  }
  float den = 0.f;
  for (int j = 0; j < T; ++j) {
    s[j] = expf(s[j] - mx);
    den += s[j];
  }
  const float inv = 1.0f / den;
  float* o = out + (size_t)(b * T + i) * C + h * hd;
  for (int d = 0; d < hd; ++d) {
    float acc = 0.f;
    for (int j = 0; j < T; ++j) acc = fmaf(s[j], kv[(size_t)(b * T + j) * 2 * C + C + h * hd + d], acc);
    o[d] = acc * inv;
  }
}

}  // namespace

extern "C" int ud_linear_f32(const UdLinearF32* desc, void* stream) {
  const UdLinearF32& d = *desc;
  if (!d.x || !d.W || !d.out || d.M <= 0 || d.N <= 0 || d.K <= 0 || (d.K & 3) || (d.ldx & 3) || (d.ldw & 3) || (d.add && d.add_mod <= 0)) {
    ud_set_error("ud_linear_f32: bad argument (K, ldx, ldw % 4 == 0)");
    return UD_ERR_BAD_ARG;
  }
  hipLaunchKernelGGL(linear_f32_kernel, dim3((d.N + 3) / 4, (d.M + 63) / 64), dim3(256), 0, (hipStream_t)stream, d);
  UD_CHECK_LAUNCH("ud_linear_f32 launch");
  return UD_OK;
}

extern "C" int ud_attention_small_f32(const float* q, const float* kv, float* out, int B, int T, int H, int C, float scale, void* stream) {
  if (!q || !kv || !out || B <= 0 || T <= 0 || T > 8 || H <= 0 || C % H) {
    ud_set_error("ud_attention_small_f32: bad argument (T <= 8)");
    return UD_ERR_BAD_ARG;
  }
  hipLaunchKernelGGL(attention_small_kernel, dim3((B * H * T + 63) / 64), dim3(64), 0, (hipStream_t)stream, q, kv, out, B, T, H, C, scale);
  UD_CHECK_LAUNCH("ud_attention_small_f32 launch");
  return UD_OK;
}
