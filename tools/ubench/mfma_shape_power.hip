// Which MFMA shape sustains more fp16 FLOP/s when the chip is POWER limited (random operands)?  (gfx950, round 6)
// A wave keeps a 64 x 64 output tile in registers and multiplies it by register-resident operand fragments, nothing else in the loop:
//   k16: v_mfma_f32_16x16x32_f16, 4 A x 4 B fragments of 16 x 32, 16 accumulators of 4 registers   (the product GEMM's shape)
//   k32: v_mfma_f32_32x32x16_f16, 2 A x 2 B fragments of 32 x 16 per k-step, two k-steps, 4 accumulators of 16 registers   (the attention kernel's shape)
// Same FLOP per loop trip (2 * 64 * 64 * 32), same accumulator footprint (64 registers).  The operands are either CONSTANT per lane or RANDOM fp16 (every
// fragment different, sign / exponent / mantissa bits toggling between consecutive MFMAs); long launches (~10 ms) so that the power controller settles.
// Prints TFLOP/s per shape and operand fill at 1 and 2 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_shape_power mfma_shape_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k16(const half8* __restrict__ src, float* out, int iters) {
  half8 a[4], b[4];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = src[(size_t)t * 8 + i]; b[i] = src[(size_t)t * 8 + 4 + i]; }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
  out[t] = s;
}
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k32(const half8* __restrict__ src, float* out, int iters) {
  half8 a[2][2], b[2][2];          // [k-step][fragment]
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i >> 1][i & 1] = src[(size_t)t * 8 + i]; b[i >> 1][i & 1] = src[(size_t)t * 8 + 4 + i]; }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
  out[t] = s;
}

template <class F> double run(F launch, double flop) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); launch(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < 4; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return flop * 4 / (ms * 1e-3) / 1e12;
}
int main() {
  const int threads = 256 * 512;
  float* out; hipMalloc(&out, threads * 4);
  std::vector<unsigned short> h((size_t)threads * 64);
  half8* src[2];
  srand(1);
  for (int mode = 0; mode < 2; ++mode) {
    for (size_t i = 0; i < h.size(); ++i) {
      if (mode == 0) h[i] = 0x3c00;                                           // 1.0 everywhere
      else h[i] = (unsigned short)(((rand() & 1) << 15) | ((12 + rand() % 6) << 10) | (rand() & 0x3ff));   // +-2^-3 .. 2^2, random mantissa
    }
    hipMalloc(&src[mode], h.size() * 2);
    hipMemcpy(src[mode], h.data(), h.size() * 2, hipMemcpyHostToDevice);
  }
  const int iters = 60000;
  for (int rep = 0; rep < 2; ++rep)
    for (int waves : {4, 8}) {
      const double flop = 256.0 * waves * iters * 2.0 * 64 * 64 * 32;
      for (int mode = 0; mode < 2; ++mode) {
        const half8* s = src[mode];
        double t16 = waves == 4 ? run([&] { hipLaunchKernelGGL(k16<4>, dim3(256), dim3(256), 0, 0, s, out, iters); }, flop)
                                : run([&] { hipLaunchKernelGGL(k16<8>, dim3(256), dim3(512), 0, 0, s, out, iters); }, flop);
        double t32 = waves == 4 ? run([&] { hipLaunchKernelGGL(k32<4>, dim3(256), dim3(256), 0, 0, s, out, iters); }, flop)
                                : run([&] { hipLaunchKernelGGL(k32<8>, dim3(256), dim3(512), 0, 0, s, out, iters); }, flop);
        printf("%d waves/CU, %s operands: 16x16x32_f16 %.0f TFLOP/s   32x32x16_f16 %.0f TFLOP/s   (ratio %.3f)\n", waves, mode ? "random  " : "constant", t16, t32, t32 / t16);
      }
    }
  return 0;
}
