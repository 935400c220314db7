// Do one wave's MFMAs and ANOTHER wave's VALU work overlap on a gfx950 SIMD?  (round 4: the attention kernel's PMC shows the matrix pipe
// busy 37 % and the VALU 59 % of every SIMD's cycles -- summing to ~100 %, i.e. no overlap at 4 waves per SIMD.)
// One "tile" per loop iteration and wave, shaped like the attention kernel's tile without LDS / DMA / barrier:
//   8 x v_mfma_f32_32x32x16_f16 into two score accumulators  ->  softmax-like VALU on the 32 scores (max chain, exp, sum, pack)
//   ->  8 x MFMA into two output accumulators with the packed scores as B operand.
// Compiled TWICE: overlap_v (-mllvm -amdgpu-mfma-vgpr-form=1: accumulators in VGPRs, what the product attention build uses) and
// overlap_a (default: hipcc keeps MFMA results in AGPRs and copies with v_accvgpr_read/write where the VALU touches them).
// FORM 1 (only meaningful in the vgpr-form build): the OUTPUT accumulators pinned to AGPRs by inline asm ("+a"), scores stay in VGPRs.
// MODE bits: 1 = MFMAs present, 2 = VALU present, 4 = s_setprio(1) around the second MFMA group (as the product does).
// W = workgroups of 256 threads per CU = waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(2))) _Float16 half2v;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#ifndef KNAME
#define KNAME kern
#endif

template <int MODE, int FORM>
__global__ __launch_bounds__(256, 4) void KNAME(float* out, const _Float16* in, int iters, float thr) {
  half8 qf[4], kf[4];          // the second key block / d block uses the same fragments rotated by one (distinct chains, no extra registers)
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i < 8; ++i) {
      qf[k][i] = in[(threadIdx.x * 8 + i + k * 2048) & 8191];
      kf[k][i] = in[(threadIdx.x * 8 + i + k * 2048 + 4096) & 8191];
    }
  f32x16 o[2];
  for (int d = 0; d < 2; ++d)
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_i = 0.f, l_i = 0.f;
  for (int it = 0; it < iters; ++it) {
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = -m_i;
      if constexpr ((MODE & 1) && FORM == 2) {
        // scores in AGPRs too: C = 0 for the first MFMA, the VALU reads them through v_accvgpr_read (compiler-inserted copies)
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=a"(s[kb]) : "v"(kf[kb]), "v"(qf[0]));
#pragma unroll
        for (int ks = 1; ks < 4; ++ks) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(s[kb]) : "v"(kf[(ks + kb) & 3]), "v"(qf[ks]));
        if (kb == 1) asm volatile("s_nop 15" : "+a"(s[0]), "+a"(s[1]));
      } else if constexpr (MODE & 1) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[(ks + kb) & 3], qf[ks], s[kb], 0, 0, 0);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(s[kb][r]));     // opaque: the VALU-only variant must not constant-fold
      }
    }
    half8 pf[2][2];
    if constexpr (MODE & 2) {
      float mt = s[0][0];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) mt = fmaxf(fmaxf(mt, s[kb][r]), s[kb][r + 1]);
      if (__any(mt > thr)) {      // never taken (thr is huge); keeps the max chain alive
        m_i += mt;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[kb][r] -= mt;
      }
      float ls = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            f32x2 pv;
            pv[0] = __builtin_amdgcn_exp2f(s[kb][t * 8 + e]);
            pv[1] = __builtin_amdgcn_exp2f(s[kb][t * 8 + e + 1]);
            ls += pv[0] + pv[1];
            const half2v ph = __builtin_convertvector(pv, half2v);
            pf[kb][t][e] = ph[0];
            pf[kb][t][e + 1] = ph[1];
          }
      l_i += ls;
    } else {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int t = 0; t < 2; ++t) pf[kb][t] = qf[kb * 2 + t];
      asm volatile("" ::"v"(s[0]), "v"(s[1]));
    }
    if constexpr (MODE & 1) {
      if constexpr (MODE & 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            if constexpr (FORM >= 1) {
              asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(o[db]) : "v"(kf[(kb * 2 + t + db) & 3]), "v"(pf[kb][t]));
            } else {
              o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[(kb * 2 + t + db) & 3], pf[kb][t], o[db], 0, 0, 0);
            }
          }
      if constexpr (MODE & 4) __builtin_amdgcn_s_setprio(0);
    } else {
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int t = 0; t < 2; ++t) asm volatile("" ::"v"(pf[kb][t]));
    }
  }
  float acc = l_i;
  for (int d = 0; d < 2; ++d)
    for (int r = 0; r < 16; ++r) acc += o[d][r];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

#ifdef OVERLAP_MAIN
// the second compile unit (AGPR form) provides kern_a<...> through these wrappers
void launch_a(int mode, int grid, float* out, const _Float16* in, int iters, float thr);
#endif

template <int MODE, int FORM>
static void launch_t(int grid, float* out, const _Float16* in, int iters, float thr) {
  hipLaunchKernelGGL((KNAME<MODE, FORM>), dim3(grid), dim3(256), 0, 0, out, in, iters, thr);
}
static void launch_here(int mode, int form, int grid, float* out, const _Float16* in, int iters, float thr) {
  if (form == 2) {
    switch (mode) {
      case 3: launch_t<3, 2>(grid, out, in, iters, thr); break;
      case 7: launch_t<7, 2>(grid, out, in, iters, thr); break;
      default: break;
    }
    return;
  }
  if (form == 1) {
    switch (mode) {
      case 3: launch_t<3, 1>(grid, out, in, iters, thr); break;
      case 7: launch_t<7, 1>(grid, out, in, iters, thr); break;
      case 1: launch_t<1, 1>(grid, out, in, iters, thr); break;
      default: break;
    }
    return;
  }
  switch (mode) {
    case 1: launch_t<1, 0>(grid, out, in, iters, thr); break;
    case 2: launch_t<2, 0>(grid, out, in, iters, thr); break;
    case 3: launch_t<3, 0>(grid, out, in, iters, thr); break;
    case 7: launch_t<7, 0>(grid, out, in, iters, thr); break;
    default: break;
  }
}
#ifndef OVERLAP_MAIN
void launch_a(int mode, int grid, float* out, const _Float16* in, int iters, float thr) { launch_here(mode, 0, grid, out, in, iters, thr); }
#else
int main() {
  float* out;
  _Float16* in;
  hipMalloc(&out, 256 * 8 * 256 * 4);
  hipMalloc(&in, 8192 * 2);
  {
    _Float16 h[8192];
    srand(1);
    for (int i = 0; i < 8192; ++i) h[i] = (_Float16)((rand() % 2001 - 1000) * 0.001f);
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  }
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto timeit = [&](auto fn) {
    fn();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      fn();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    return best * 1e6 / iters;   // ns per iteration (all waves advance together)
  };
  const char* mname[8] = {"", "MFMA only (16/iter)", "VALU only", "MFMA+VALU", "", "", "", "MFMA+VALU, setprio around 2nd group"};
  printf("ns per tile-iteration of ONE wave slot (W waves per SIMD share the SIMD; ideal MFMA-only = W x 16 x 32 cycles)\n");
  for (int W : {1, 2, 4}) {
    const int grid = 256 * W;
    for (int mode : {1, 2, 3, 7}) {
      float tv = timeit([&] { launch_here(mode, 0, grid, out, in, iters, 1e30f); });
      float ta = timeit([&] { launch_a(mode, grid, out, in, iters, 1e30f); });
      float th = (mode == 2) ? 0.f : timeit([&] { launch_here(mode, 1, grid, out, in, iters, 1e30f); });
      float t2 = (mode == 3 || mode == 7) ? timeit([&] { launch_here(mode, 2, grid, out, in, iters, 1e30f); }) : 0.f;
      printf("W=%d  %-38s  vgpr-form %8.1f  default-build %8.1f  O in AGPR %8.1f  S and O in AGPR %8.1f   per wave-tile: %7.1f / %7.1f / %7.1f / %7.1f ns\n", W, mname[mode],
             tv, ta, th, t2, tv / W, ta / W, th / W, t2 / W);
    }
  }
  return 0;
}
#endif
