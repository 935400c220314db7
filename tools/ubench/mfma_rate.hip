// MFMA issue-rate micro-benchmark (gfx950): 8 waves per workgroup (2 per SIMD), register-resident operands, 16 independent
// accumulators per wave.  Prints TFLOP/s for v_mfma_f32_16x16x32_f16 and v_mfma_f32_32x32x16_f16.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k16(float* out, int iters) {
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f - i * 0.01f); }
  f32x4 acc[32];
  for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k32(float* out, int iters) {
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f - i * 0.01f); }
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class F> double run(F launch, double flop) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return flop * 5 / (ms * 1e-3) / 1e12;
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  const int iters = 2000;
  for (int waves : {4, 8}) {
    double f16 = 256.0 * waves * iters * 32 * 2.0 * 16 * 16 * 32, f32 = 256.0 * waves * iters * 16 * 2.0 * 32 * 32 * 16;
    double t16 = waves == 4 ? run([&] { hipLaunchKernelGGL(k16<4>, dim3(256), dim3(256), 0, 0, out, iters); }, f16)
                            : run([&] { hipLaunchKernelGGL(k16<8>, dim3(256), dim3(512), 0, 0, out, iters); }, f16);
    double t32 = waves == 4 ? run([&] { hipLaunchKernelGGL(k32<4>, dim3(256), dim3(256), 0, 0, out, iters); }, f32)
                            : run([&] { hipLaunchKernelGGL(k32<8>, dim3(256), dim3(512), 0, 0, out, iters); }, f32);
    printf("%d waves/CU: 16x16x32_f16 %.0f TFLOP/s   32x32x16_f16 %.0f TFLOP/s\n", waves, t16, t32);
  }
  return 0;
}
