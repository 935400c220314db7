// Shared device helpers for the gfx950 (CDNA4, wave64) kernels.  Written for MI355X only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/unidepth_hip.h"

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(2))) _Float16 half2v;
typedef __attribute__((ext_vector_type(4))) _Float16 half4;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define UD_WAVE 64

// async global -> LDS copy, 16 B per lane.  LDS destination = wave-uniform base + lane*16 (hardware rule),
// global source is per lane: swizzles are applied on the SOURCE address, never on the destination.
__device__ __forceinline__ void ud_glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// the same copy through a buffer descriptor: address = base(SGPR x4) + voff (one VGPR, bytes) + soff (SGPR, bytes); no 64-bit
// per-lane pointer, no per-K-tile address VALU, and an offset >= num_records reads as zeros (conv padding for free).
typedef __amdgpu_buffer_rsrc_t ud_rsrc_t;
__device__ __forceinline__ ud_rsrc_t ud_make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void ud_bufl16(ud_rsrc_t r, unsigned voff, int soff, void* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, soff, 0, 0);
}

// exact-erf GELU (nn.GELU default).  erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. fp32 round-off level and
// three orders below the fp16 rounding of the stored result), written on |x| so that no sign fix-up is needed:
//   gelu(x) = x Phi(x) = max(x, 0) - |x| q,   q = Phi(-|x|) = 0.5 poly(t) exp(-x^2 / 2),   t = 1 / (1 + p |x| / sqrt 2)
// 1 v_rcp + 1 v_exp + 11 plain VALU (abs / neg are operand modifiers) -- erff() is ~45 instructions.
__device__ __forceinline__ float ud_gelu_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
  float poly = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
  poly = fmaf(poly, t, 0.5f * 1.421413741f);
  poly = fmaf(poly, t, 0.5f * -0.284496736f);
  poly = fmaf(poly, t, 0.5f * 0.254829592f);
  poly *= t;
  const float e = __builtin_amdgcn_exp2f(x * x * (-0.5f * 1.4426950408889634f));
  return fmaf(-ax, poly * e, fmaxf(x, 0.0f));
}
__device__ __forceinline__ float ud_lrelu(float x) { return x > 0.0f ? x : 0.01f * x; }
__device__ __forceinline__ float ud_clampexp(float x) { return __expf(fminf(fmaxf(x, -10.0f), 10.0f)); }
__device__ __forceinline__ float ud_act(float x, int act) {
  return act == UD_ACT_GELU ? ud_gelu_erf(x) : (act == UD_ACT_LRELU ? ud_lrelu(x) : x);
}

__device__ __forceinline__ float ud_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

void ud_set_error(const char* msg);
#define UD_MAX_DEVICES 64
bool ud_attr_once(bool (&done)[UD_MAX_DEVICES]);   // true if the calling kernel's launch attribute was already set on the current device
#ifdef UD_TOOLS
int ud_debug_flags_host();   // tools builds only -- bisect switches (api.cpp): bit0 attention: no deferred max; bit1 GELU via erff; bit2 no LDS-staged stores; bit3 128x128 tiles only; bit4 plain 128x128 kernel at low tile counts (no 4-stage ring); bit5 no K split across CUs
#else
static inline int ud_debug_flags_host() { return 0; }
#endif
#define UD_CHECK_LAUNCH(name)                          \
  do {                                                 \
    hipError_t e_ = hipGetLastError();                 \
    if (e_ != hipSuccess) {                            \
      ud_set_error(name);                              \
      return UD_ERR_LAUNCH;                            \
    }                                                  \
  } while (0)
