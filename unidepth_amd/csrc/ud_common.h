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

__device__ __forceinline__ void ud_bufl4(ud_rsrc_t r, unsigned voff, int soff, void* lds_wave_base) {     // 4 B per lane: LDS = base + lane * 4
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 4, (int)voff, soff, 0, 0);
}

// exact-erf GELU (nn.GELU default), transcendental-free:  gelu(x) = x Phi(x) = max(x, 0) - u q(u),  u = min(|x|, 5),  q(u) = Phi(-u)
// as a degree-13 polynomial in t = 0.4 u - 1 (Chebyshev interpolant of 0.5 erfc(u / sqrt 2) on [0, 5], monomial form, Horner in fp32).
// |error| <= 2.6e-6 absolute for every x (fp32 emulation against the fp64 erfc form on 4M points; beyond |x| = 5 the tail is frozen
// at 5 Phi(-5) = 1.4e-6) -- two orders below the fp16 rounding of the stored result.  17 plain FMA-rate ops per element and, in
// the packed form used by the full-tile GEMM epilogues, 9.5 issue slots (v_pk_fma_f32 on two elements); the previous
// Abramowitz-Stegun form cost 13 ops + v_rcp + v_exp (quarter rate: ~21 slots), 5 us per 256 x 256 tile of fc1.
// The scalar and the packed function perform the same operations in the same order: bit-identical results, so an element's value
// does not depend on whether its tile took the straight-line or the edge-tile epilogue.
#define UD_GELU_COEFFS                                                                                                   \
  {6.210224237e-03f, -4.382255673e-02f, 1.368843615e-01f, -2.395744771e-01f, 2.327018231e-01f, -6.588349491e-02f,        \
   -1.309080124e-01f, 1.644788533e-01f, -2.503387816e-02f, -7.997140288e-02f, 4.101016745e-02f, 1.517937891e-02f,        \
   -1.086505502e-02f, -4.060750653e-04f}
__device__ __forceinline__ float ud_gelu_erf(float x) {
  constexpr float cf[14] = UD_GELU_COEFFS;
  const float u = fminf(fabsf(x), 5.0f);
  const float t = fmaf(u, 0.4f, -1.0f);
  float q = cf[13];
#pragma unroll
  for (int k = 12; k >= 0; --k) q = fmaf(q, t, cf[k]);
  return fmaf(-u, q, fmaxf(x, 0.0f));
}
__device__ __forceinline__ f32x2 ud_gelu_erf2(f32x2 x) {
  constexpr float cf[14] = UD_GELU_COEFFS;
  const f32x2 u = {fminf(fabsf(x[0]), 5.0f), fminf(fabsf(x[1]), 5.0f)};
  const f32x2 t = __builtin_elementwise_fma(u, (f32x2){0.4f, 0.4f}, (f32x2){-1.0f, -1.0f});
  f32x2 q = {cf[13], cf[13]};
#pragma unroll
  for (int k = 12; k >= 0; --k) q = __builtin_elementwise_fma(q, t, (f32x2){cf[k], cf[k]});
  return __builtin_elementwise_fma(-u, q, (f32x2){fmaxf(x[0], 0.0f), fmaxf(x[1], 0.0f)});
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
