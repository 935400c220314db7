// Shared device helpers for the gfx950 (CDNA4, wave64) kernels.  Written for MI355X only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/unidepth_hip.h"

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(2))) _Float16 half2v;
typedef __attribute__((ext_vector_type(4))) _Float16 half4;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define UD_WAVE 64

// async global -> LDS copy, 16 B per lane.  LDS destination = wave-uniform base + lane*16 (hardware rule),
// global source is per lane: swizzles are applied on the SOURCE address, never on the destination.
__device__ __forceinline__ void ud_glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float ud_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float ud_lrelu(float x) { return x > 0.0f ? x : 0.01f * x; }
__device__ __forceinline__ float ud_act(float x, int act) {
  return act == UD_ACT_GELU ? ud_gelu_erf(x) : (act == UD_ACT_LRELU ? ud_lrelu(x) : x);
}

__device__ __forceinline__ float ud_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

void ud_set_error(const char* msg);
#define UD_CHECK_LAUNCH(name)                          \
  do {                                                 \
    hipError_t e_ = hipGetLastError();                 \
    if (e_ != hipSuccess) {                            \
      ud_set_error(name);                              \
      return UD_ERR_LAUNCH;                            \
    }                                                  \
  } while (0)
