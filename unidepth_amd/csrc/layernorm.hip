// LayerNorm statistics kernel: fp32 rows -> normalised fp16 rows (affine folded into the consumer GEMM's weights).
// HBM-bound: one wave64 per row, 16-byte loads, two-pass (mean, then centred variance) entirely in registers,
// wavefront reductions via cross-lane shuffles.  Algorithmic bytes per row: 4*D read + 2*D written.
#include "ud_common.h"

namespace {

template <int NIT, bool F32OUT>  // D <= NIT * 256
__global__ __launch_bounds__(256) void layernorm_kernel(const UdLayerNorm p) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= p.rows) return;
  // cls_y: rows_per_img + 1 rows per image, the first one = the row in front of the others (class token), normalised to fp32
  const int rpi = p.rows_per_img + (p.cls_y ? 1 : 0);
  const int img = r / rpi;
  const int pp = r - img * rpi - (p.cls_y ? 1 : 0);
  const bool is_cls = pp < 0;
  const size_t irow = (size_t)img * p.in_rows_per_img + pp + p.in_row_off;
  const size_t orow = (size_t)img * p.out_rows_per_img + (is_cls ? 0 : pp) + p.out_row_off;
  const float* x = p.x + irow * p.ldx;
  const float* addp = p.add ? p.add + (size_t)(pp + p.out_row_off) * p.ldx : nullptr;
  half_t* y = (half_t*)p.y + (F32OUT ? 0 : orow * p.ldy);
  float* yf = (float*)p.y + (F32OUT ? orow * p.ldy : 0);
  f32x4 v[NIT];
  float s = 0.0f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = it * 256 + lane * 4;
    if (c < p.D) {
      v[it] = *(const f32x4*)(x + c);
      if (addp) v[it] += *(const f32x4*)(addp + c);
      s += (v[it][0] + v[it][1]) + (v[it][2] + v[it][3]);
    } else {
      v[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  const float mean = ud_wave_sum(s) / (float)p.D;
  float q = 0.0f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = it * 256 + lane * 4;
    if (c < p.D) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[it][e] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(ud_wave_sum(q) / (float)p.D + p.eps);
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = it * 256 + lane * 4;
    if (c < p.D) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[it][e] - mean) * rstd;
      if (p.gamma) o = o * *(const f32x4*)(p.gamma + c) + (p.beta ? *(const f32x4*)(p.beta + c) : (f32x4){0.f, 0.f, 0.f, 0.f});
      if constexpr (F32OUT) {
        *(f32x4*)(yf + c) = o;
      } else if (is_cls) {
        *(f32x4*)(p.cls_y + (size_t)img * p.ldcls + c) = o;
      } else {
        half4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (half_t)o[e];
        *(half4*)(y + c) = h;
      }
    }
  }
}

// the reduction step of a LayerNorm folded into two GEMMs (UdGemm.row_stats_out -> row_stats_in): 16 lanes per row, lane q loads the
// (sum, sum of squares) pair of slab q (a wave reads 4 rows = 512 contiguous bytes), fixed butterfly order over the 16 lanes -- the same
// order for every row wherever it sits, so results are bit-reproducible and independent of the image's position in the batch.
// (One thread per row with a serial loop over its 16 strided pairs took 6.5 us for 11008 rows; this form is launch-bound.)
__global__ __launch_bounds__(256) void row_stats_finalize_kernel(const float* __restrict__ part, float* __restrict__ stats, int M, int slabs, float inv_d, float eps) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int m = gid >> 4, q = gid & 15;
  f32x2 v = {0.f, 0.f};
  if (m < M && q < slabs) v = *(const f32x2*)(part + ((size_t)m * slabs + q) * 2);
  float s1 = v[0], s2 = v[1];
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  if (m < M && q == 0) {
    const float mean = s1 * inv_d;
    const float var = fmaxf(__builtin_fmaf(-mean, mean, s2 * inv_d), 0.0f);
    f32x2 o;
    o[0] = rsqrtf(var + eps);
    o[1] = -mean * o[0];
    *(f32x2*)(stats + 2 * (size_t)m) = o;
  }
}

}  // namespace

extern "C" int ud_row_stats_finalize(const float* partials, float* stats, int M, int slabs, int D, float eps, void* stream) {
  if (!partials || !stats || M <= 0 || slabs <= 0 || slabs > 16 || D <= 0) {
    ud_set_error("ud_row_stats_finalize: bad argument (1 <= slabs <= 16)");
    return UD_ERR_BAD_ARG;
  }
  hipLaunchKernelGGL(row_stats_finalize_kernel, dim3((M + 15) / 16), dim3(256), 0, (hipStream_t)stream, partials, stats, M, slabs, 1.0f / (float)D, eps);
  UD_CHECK_LAUNCH("ud_row_stats_finalize launch");
  return UD_OK;
}

extern "C" int ud_layernorm_f32_f16(const UdLayerNorm* desc, void* stream) {
  const UdLayerNorm& d = *desc;
  if (!d.x || !d.y || d.rows <= 0 || d.D <= 0 || (d.D & 3) || d.D > 2048 || (d.ldx & 3) || (d.ldy & 3) || d.rows_per_img <= 0) {
    ud_set_error("ud_layernorm_f32_f16: bad argument (D % 4 == 0, D <= 2048)");
    return UD_ERR_BAD_ARG;
  }
  if (d.cls_y && (d.out_f32 || d.add || d.in_row_off < 1 || (d.ldcls & 3) || d.ldcls < d.D || d.rows % (d.rows_per_img + 1))) {
    ud_set_error("ud_layernorm_f32_f16: cls_y needs fp16 y, no add, in_row_off >= 1, ldcls % 4 == 0 and rows = images x (rows_per_img + 1)");
    return UD_ERR_BAD_ARG;
  }
  dim3 grid((d.rows + 3) / 4);
  hipStream_t s = (hipStream_t)stream;
#define UD_LN_LAUNCH(NIT)                                                                                  \
  do {                                                                                                    \
    if (d.out_f32) hipLaunchKernelGGL((layernorm_kernel<NIT, true>), grid, dim3(256), 0, s, d);           \
    else hipLaunchKernelGGL((layernorm_kernel<NIT, false>), grid, dim3(256), 0, s, d);                    \
  } while (0)
  if (d.D <= 256) UD_LN_LAUNCH(1);
  else if (d.D <= 512) UD_LN_LAUNCH(2);
  else if (d.D <= 1024) UD_LN_LAUNCH(4);
  else UD_LN_LAUNCH(8);
#undef UD_LN_LAUNCH
  UD_CHECK_LAUNCH("ud_layernorm_f32_f16 launch");
  return UD_OK;
}
