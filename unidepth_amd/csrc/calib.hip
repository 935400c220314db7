// Box calibration for the measurement line (bench.py `roofline.attainable_this_box`): what the matrix pipes of THIS MI355X sustain on a pure
// MFMA instruction stream with random fp16 operands.  Not on any product path.  The datasheet peak (2.5 PFLOP/s dense fp16) assumes 2.4 GHz; an
// MFMA-dense stream on data that toggles the multiplier arrays runs power-limited at 1.6-1.8 GHz (profiles/r02_mfma_attainable.txt: 1462 TFLOP/s
// on one box, 1284 on another in round 6), and boxes of this pool differ by +-4 % on every kernel -- a per-box number lets a reader tell a slow
// box from a regression.
//
// Stream: every wave holds 4 A and 4 B fragments (random fp16 from `operands`, one 16-byte load per lane and fragment) and 8 independent
// 32 x 32 fp32 accumulators (128 VGPRs); an iteration = 16 v_mfma_f32_32x32x16_f16 (a[i] x b[j] for the 4 x 4 pairs into accumulator (i*4+j) & 7:
// dependent MFMAs are 8 instructions apart, the pipe never waits), nothing else.  4 waves per workgroup (one per SIMD), 4 workgroups per CU's
// worth of grid (1024 workgroups): two waves per SIMD resident, as in the product GEMM.  FLOP per launch = grid * 4 * iters * 16 * 32768.
#include "ud_common.h"

namespace {
__global__ __launch_bounds__(256) void mfma_stream_kernel(const half8* __restrict__ operands, int iters, float* __restrict__ sink) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const half8* src = operands + ((size_t)(blockIdx.x & 63) * 4 + wv) * 8 * 64 + lane;
  half8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = src[i * 64];
    b[i] = src[(4 + i) * 64];
  }
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[(i * 4 + j) & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[(i * 4 + j) & 7], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) sink[blockIdx.x * 256 + threadIdx.x] = s;        // keeps the stream alive; never true for the random operands used
}

// The product GEMM family's own instruction: v_mfma_f32_16x16x32_f16, 4 A x 4 B fragments, 16 independent accumulators of 4 registers (a 64 x 64 register
// tile per wave); 8 waves per workgroup, one workgroup per CU's worth of grid (two waves per SIMD).  On random operands this stream sustains MORE than the
// 32 x 32 x 16 one above (1.90 against 1.2-1.7 PFLOP/s, profiles/r06_kloop_ablation.txt): bench.py runs both and reports the higher as attainable_this_box.
__global__ __launch_bounds__(512) void mfma_stream16_kernel(const half8* __restrict__ operands, int iters, float* __restrict__ sink) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const half8* src = operands + ((size_t)(blockIdx.x & 31) * 8 + wv) * 8 * 64 + lane;
  half8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = src[i * 64];
    b[i] = src[(4 + i) * 64];
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
  if (s == 123.456f) sink[blockIdx.x * 512 + threadIdx.x] = s;
}
}  // namespace

// operands: 64 * 4 * 8 * 64 * 16 bytes = 2 MiB of fp16 values (|x| <~ 1 keeps the fp32 accumulators finite for millions of iterations);
// sink: grid * 256 floats.  Returns the FLOP count of one launch through *flop_out.
extern "C" int ud_calib_mfma_stream(const void* operands, int iters, int workgroups, void* sink, double* flop_out, void* stream) {
  if (!operands || !sink || iters < 1 || workgroups < 1) {
    ud_set_error("ud_calib_mfma_stream: bad arguments");
    return UD_ERR_BAD_ARG;
  }
  hipLaunchKernelGGL(mfma_stream_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, (const half8*)operands, iters, (float*)sink);
  UD_CHECK_LAUNCH("ud_calib_mfma_stream launch");
  if (flop_out) *flop_out = (double)workgroups * 4.0 * iters * 16.0 * 32768.0;
  return UD_OK;
}

// the 16 x 16 x 32 stream: workgroups of 8 waves; sink: workgroups * 512 floats; same operand buffer
extern "C" int ud_calib_mfma_stream16(const void* operands, int iters, int workgroups, void* sink, double* flop_out, void* stream) {
  if (!operands || !sink || iters < 1 || workgroups < 1) {
    ud_set_error("ud_calib_mfma_stream16: bad arguments");
    return UD_ERR_BAD_ARG;
  }
  hipLaunchKernelGGL(mfma_stream16_kernel, dim3(workgroups), dim3(512), 0, (hipStream_t)stream, (const half8*)operands, iters, (float*)sink);
  UD_CHECK_LAUNCH("ud_calib_mfma_stream16 launch");
  if (flop_out) *flop_out = (double)workgroups * 8.0 * iters * 16.0 * 16384.0;
  return UD_OK;
}
