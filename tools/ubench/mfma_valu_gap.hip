// How many VALU instructions hide under one v_mfma_f32_32x32x16_f16 on a gfx950 SIMD?  (Round 5: three structurally different attention kernels all show
// MFMA-busy % + VALU-busy % ~ 90 % of the SIMD cycles -- the two pipes take turns.)  One wave per SIMD (or two), a loop of
//     { 1 MFMA on one of two alternating accumulators;  N independent VALU instructions on registers the MFMA does not touch }
// timed for N = 0 .. 16, with the accumulators in VGPRs ("+v": what -amdgpu-mfma-vgpr-form gives the attention kernel) and in AGPRs ("+a"), and with
// v_fma_f32 or v_exp_f32 as the filler.  ns per iteration; an MFMA alone = 32 cycles.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_gap.hip -o tools/ubench/mfma_valu_gap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(16))) float f32x16;


template <int N, int FORM, int TRANS>
__global__ __launch_bounds__(64) void kern(float* out, int iters) {
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (threadIdx.x + i)); b[i] = (_Float16)(0.02f * (threadIdx.x * 3 + i)); }
  f32x16 c0, c1;
  for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = 0.001f * (threadIdx.x + i);
  const float y = 0.999f;
  for (int it = 0; it < iters; ++it) {
    if constexpr (FORM == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b));
#pragma unroll
    for (int k = 0; k < N; ++k) {
      if constexpr (TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(x[k & 7]));
      else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[k & 7]) : "v"(y));
    }
    if constexpr (FORM == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c1) : "v"(a), "v"(b));
#pragma unroll
    for (int k = 0; k < N; ++k) {
      if constexpr (TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(x[(k + 3) & 7]));
      else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[(k + 3) & 7]) : "v"(y));
    }
  }
  float acc = 0.f;
  for (int r = 0; r < 16; ++r) acc += c0[r] + c1[r];
  for (int i = 0; i < 8; ++i) acc += x[i];
  out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int N, int FORM, int TRANS>
static float run(float* out, int grid, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((kern<N, FORM, TRANS>), dim3(grid), dim3(64), 0, 0, out, iters);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((kern<N, FORM, TRANS>), dim3(grid), dim3(64), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  return best * 1e6f / (2.0f * iters);      // ns per {MFMA + N fillers}
}

template <int FORM, int TRANS>
static void sweep(float* out, int grid, const char* tag) {
  const int iters = 20000;
  printf("%-44s", tag);
  printf(" N=0 %6.2f", run<0, FORM, TRANS>(out, grid, iters));
  printf(" | 2 %6.2f", run<2, FORM, TRANS>(out, grid, iters));
  printf(" | 4 %6.2f", run<4, FORM, TRANS>(out, grid, iters));
  printf(" | 5 %6.2f", run<5, FORM, TRANS>(out, grid, iters));
  printf(" | 6 %6.2f", run<6, FORM, TRANS>(out, grid, iters));
  printf(" | 8 %6.2f", run<8, FORM, TRANS>(out, grid, iters));
  printf(" | 12 %6.2f", run<12, FORM, TRANS>(out, grid, iters));
  printf(" | 16 %6.2f  ns per MFMA + N fillers\n", run<16, FORM, TRANS>(out, grid, iters));
}

int main() {
  float* out;
  hipMalloc(&out, 4096 * 64 * 4);
  for (int wps : {1, 2}) {
    const int grid = 1024 * wps;                        // 64-thread blocks: 4 per CU = one wave per SIMD
    char t[128];
    snprintf(t, sizeof(t), "%d wave/SIMD, acc in VGPR, filler v_fma_f32", wps); sweep<0, 0>(out, grid, t);
    snprintf(t, sizeof(t), "%d wave/SIMD, acc in AGPR, filler v_fma_f32", wps); sweep<1, 0>(out, grid, t);
    snprintf(t, sizeof(t), "%d wave/SIMD, acc in VGPR, filler v_exp_f32", wps); sweep<0, 1>(out, grid, t);
    snprintf(t, sizeof(t), "%d wave/SIMD, acc in AGPR, filler v_exp_f32", wps); sweep<1, 1>(out, grid, t);
  }
  return 0;
}
