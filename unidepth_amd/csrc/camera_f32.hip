// fp32 island for the camera head (see UdLinearF32 in include/unidepth_hip.h for why): 4 tokens per image, ~9 M weights,
// latency-bound; plain fp32 FMAs are the right tool (no MFMA: at M = 4*B rows the matrix pipe would idle anyway).
#include "ud_common.h"

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
