// K-loop experiment for round 3 (never part of the product): a 256 x 256 output tile computed by FOUR waves (2 x 2, 128 x 128 per wave, one
// wave per SIMD, 256 accumulator registers) with v_mfma_f32_32x32x16_f16 -- the only MFMA shape that keeps the matrix pipe full
// from a single wave (tools/ubench/mfma_rate: 1 wave / SIMD reaches 2132 TFLOP/s with 32x32x16 but 1395 with 16x16x32) -- against
// the product's 8-wave 16x16x32 layout.  Per 64-deep K-tile the 4-wave layout reads 128 KB of fragments from LDS instead of 192 KB.
// Operands: A [M, K], W [N, K] fp16 row-major (K contiguous); C [M, N] fp32 = A W^T.  Same LDS image as the product kernel: [row][64
// halves], 16-byte chunks XOR-swizzled on the SOURCE address, operands copied by buffer_load ... lds.
// build: hipcc --offload-arch=gfx950 -O3 -o gemm4w gemm4w.hip ; run on the GPU box: ./gemm4w [M N K] [swz]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void bufl16(rsrc_t r, unsigned voff, int soff, void* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, soff, 0, 0);
}

constexpr int STAGE = 65536;   // A 256 rows x 128 B | B 256 rows x 128 B

// SWZ: 0 = chunk ^ ((row >> 1) & 7) (the product's), 1 = chunk ^ (row & 7), 2 = chunk ^ ((row >> 2) & 7)
template <int SWZ>
__device__ __forceinline__ int swz_of(int row) { return SWZ == 0 ? (row >> 1) & 7 : SWZ == 1 ? row & 7 : (row >> 2) & 7; }

// ABL (compile time): 0 full, 2 no operand DMA after the first K-tile, 3 no DMA and no barrier, 4 MFMAs only (no fragment reads either)
template <int SWZ, int ABL = 0>
__global__ __launch_bounds__(256) void gemm4w_kernel(const half_t* A, const half_t* W, float* C, int M, int N, int K, int same) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv >> 1, wn = wv & 1;
  const int tiles_n = N >> 8;
  const int m0 = (blockIdx.x / tiles_n) << 8, n0 = (blockIdx.x % tiles_n) << 8;
  const int nk = K >> 6;
  const rsrc_t rA = make_rsrc(A, (unsigned)((size_t)M * K * 2)), rW = make_rsrc(W, (unsigned)((size_t)N * K * 2));
  // loader: thread owns chunks tid + 256 i -> rows (tid >> 3) + 32 i, physical chunk tid & 7 (holds logical chunk (tid & 7) ^ swz(row))
  unsigned pa[8], pb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = (tid >> 3) + 32 * i;
    const int c = (tid & 7) ^ swz_of<SWZ>(row);
    pa[i] = ((unsigned)((same == 1 ? 0 : m0) + row) * (unsigned)K + c * 8) * 2u;      // same: every block streams tile (0, 0): operands stay in L2
    pb[i] = ((unsigned)((same == 1 ? 0 : n0) + row) * (unsigned)K + c * 8) * 2u;
  }
  auto issue = [&](int kt, int stg) {
    char* sb = smem + stg * STAGE + wv * 1024;
#pragma unroll
    for (int i = 0; i < 8; ++i) bufl16(rA, pa[i], kt * 128, sb + i * 4096);
#pragma unroll
    for (int i = 0; i < 8; ++i) bufl16(rW, pb[i], kt * 128, sb + 32768 + i * 4096);
  };
  // fragments of k16-step s: rows base + 32 I + (lane & 31), logical chunk 2 s + (lane >> 5)
  int aoff[4], boff[4];
  {
    const int ra = wm * 128 + (lane & 31), rb = wn * 128 + (lane & 31);      // + 32 I keeps swz only if 32 I does not change it:
#pragma unroll
    for (int s = 0; s < 4; ++s) {                                             // (row >> k) & 7 with 32 I added: bits >= 5 change for k >= 3 only
      aoff[s] = ra * 128 + (((2 * s + (lane >> 5)) ^ swz_of<SWZ>(ra)) << 4);
      boff[s] = 32768 + rb * 128 + (((2 * s + (lane >> 5)) ^ swz_of<SWZ>(rb)) << 4);
    }
  }
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  half8 fa[2][4], fb[2][4];
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int stg = 0;
  {
    const char* sb = smem;
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[0][i] = *(const half8*)(sb + aoff[0] + i * 4096);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[0][j] = *(const half8*)(sb + boff[0] + j * 4096);
  }
  for (int kt = 0; kt < nk; ++kt) {
    const char* sb = smem + stg * STAGE;
    const char* sbn = smem + (stg ^ 1) * STAGE;
    if (kt + 1 < nk && ABL < 2) issue(kt + 1, stg ^ 1);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;
      if constexpr (ABL >= 4) {
      } else if (s < 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[nxt][i] = *(const half8*)(sb + aoff[s + 1] + i * 4096);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[nxt][j] = *(const half8*)(sb + boff[s + 1] + j * 4096);
      } else {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (ABL < 3) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + 1 < nk) {
#pragma unroll
          for (int i = 0; i < 4; ++i) fa[nxt][i] = *(const half8*)(sbn + aoff[0] + i * 4096);
#pragma unroll
          for (int j = 0; j < 4; ++j) fb[nxt][j] = *(const half8*)(sbn + boff[0] + j * 4096);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[ABL >= 4 ? 0 : cur][j], fa[ABL >= 4 ? 0 : cur][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    stg ^= 1;
  }
  // D[n_local][m_local]: lane holds m_local = lane & 31, register r -> n_local = 8 (r >> 2) + 4 (lane >> 5) + (r & 3)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 128 + i * 32 + (lane & 31);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * 128 + j * 32 + 8 * g + 4 * (lane >> 5);
        f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        *(f32x4*)(C + (size_t)m * N + n) = v;
      }
  }
}

// ---- the same tile with a DEEPER operand pipeline: a 2-stage LDS ring fed by buffer_load ... lds gives one K-tile of lead, so a K-tile
// cannot be shorter than the memory latency under load (~1.2-1.5 us: exactly what the product kernel's K-tile takes); the matrix pipes
// would need 1 us.  160 KB of LDS hold no third 64 KB stage, but at one wave per SIMD the register file does: K-tiles t + 2 and t + 3
// travel through VGPRs (2 x 64 registers per lane) and are written to the stage that K-tile t frees (ds_write_b128): three K-tiles of lead.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
struct T0 { static constexpr int value = 0; };
struct T1 { static constexpr int value = 1; };
template <int SWZ>
__global__ __launch_bounds__(256) void gemm4w_rp_kernel(const half_t* A, const half_t* W, float* C, int M, int N, int K, int same) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv >> 1, wn = wv & 1;
  const int tiles_n = N >> 8;
  const int m0 = (blockIdx.x / tiles_n) << 8, n0 = (blockIdx.x % tiles_n) << 8;
  const int nk = K >> 6;
  const rsrc_t rA = make_rsrc(A, (unsigned)((size_t)M * K * 2)), rW = make_rsrc(W, (unsigned)((size_t)N * K * 2));
  unsigned pa[8], pb[8];
  int wofs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = (tid >> 3) + 32 * i;
    const int c = (tid & 7) ^ swz_of<SWZ>(row);
    pa[i] = ((unsigned)((same == 1 ? 0 : m0) + row) * (unsigned)K + c * 8) * 2u;      // same: every block streams tile (0, 0): operands stay in L2
    pb[i] = ((unsigned)((same == 1 ? 0 : n0) + row) * (unsigned)K + c * 8) * 2u;
    wofs[i] = row * 128 + (tid & 7) * 16;
  }
  u32x4 ga[2][8], gb[2][8];
  auto gload = [&](auto SET, int kt) {
    constexpr int set = decltype(SET)::value;
    kt = kt < nk ? kt : nk - 1;                    // uniform load count per K-tile: the in-loop wait is vmcnt(16)
#pragma unroll
    for (int i = 0; i < 8; ++i) ga[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rA, (int)pa[i], kt * 128, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) gb[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rW, (int)pb[i], kt * 128, 0);
  };
  auto lwrite = [&](auto SET, int stg) {
    constexpr int set = decltype(SET)::value;
    char* sb = smem + stg * STAGE;
#pragma unroll
    for (int i = 0; i < 8; ++i) *(u32x4*)(sb + wofs[i]) = ga[set][i];
#pragma unroll
    for (int i = 0; i < 8; ++i) *(u32x4*)(sb + 32768 + wofs[i]) = gb[set][i];
  };
  int aoff[4], boff[4];
  {
    const int ra = wm * 128 + (lane & 31), rb = wn * 128 + (lane & 31);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      aoff[s] = ra * 128 + (((2 * s + (lane >> 5)) ^ swz_of<SWZ>(ra)) << 4);
      boff[s] = 32768 + rb * 128 + (((2 * s + (lane >> 5)) ^ swz_of<SWZ>(rb)) << 4);
    }
  }
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  half8 fa[2][4], fb[2][4];
  gload(T0{}, 0); gload(T1{}, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lwrite(T0{}, 0); lwrite(T1{}, 1);
  gload(T0{}, 2); gload(T1{}, 3);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  {
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[0][i] = *(const half8*)(smem + aoff[0] + i * 4096);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[0][j] = *(const half8*)(smem + boff[0] + j * 4096);
  }
  // one K-tile: stage STG holds tile kt; register set STG holds tile kt + 2 (landed or landing), set STG ^ 1 tile kt + 3 (just issued)
  auto ktile = [&](auto STGT, int kt) {
    constexpr int stg = decltype(STGT)::value;
    const char* sb = smem + stg * STAGE;
    const char* sbn = smem + (stg ^ 1) * STAGE;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;
      if (s < 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[nxt][i] = *(const half8*)(sb + aoff[s + 1] + i * 4096);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[nxt][j] = *(const half8*)(sb + boff[s + 1] + j * 4096);
      } else {
        asm volatile("s_waitcnt vmcnt(16)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");      // tile kt + 2 is in registers; own fragment reads done
        __builtin_amdgcn_s_barrier();                                                      // every wave is done with stage stg
        asm volatile("" ::: "memory");
        if constexpr (stg == 0) lwrite(T0{}, 0); else lwrite(T1{}, 1);                      // tile kt + 2 -> the stage tile kt leaves
        if constexpr (stg == 0) gload(T0{}, kt + 4); else gload(T1{}, kt + 4);              // and its register set takes tile kt + 4
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[nxt][i] = *(const half8*)(sbn + aoff[0] + i * 4096);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[nxt][j] = *(const half8*)(sbn + boff[0] + j * 4096);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[cur][j], fa[cur][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  for (int kt = 0; kt < nk; kt += 2) {
    ktile(T0{}, kt);
    ktile(T1{}, kt + 1);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 128 + i * 32 + (lane & 31);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * 128 + j * 32 + 8 * g + 4 * (lane >> 5);
        f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        *(f32x4*)(C + (size_t)m * N + n) = v;
      }
  }
}

__global__ void ref_kernel(const half_t* A, const half_t* W, const int* mi, const int* ni, float* out, int K, int cnt) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= cnt) return;
  double s = 0;
  for (int k = 0; k < K; ++k) s += (double)(float)A[(size_t)mi[t] * K + k] * (double)(float)W[(size_t)ni[t] * K + k];
  out[t] = (float)s;
}

template <int SWZ, bool RP = false, int ABL = 0>
void run(int same, const half_t* dA, const half_t* dW, float* dC, int M, int N, int K, const std::vector<int>& mi, const std::vector<int>& ni, int* dmi, int* dni, float* dref) {
  auto kern = RP ? gemm4w_rp_kernel<SWZ> : gemm4w_kernel<SWZ, ABL>;
  if (ABL) same = ABL;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
  const int blocks = (M >> 8) * (N >> 8);
  auto launch = [&]() { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 2 * STAGE, 0, dA, dW, dC, M, N, K, same); };
  hipMemset(dC, 0, (size_t)M * N * 4);
  launch(); hipDeviceSynchronize();
  std::vector<float> c((size_t)M * N), ref(mi.size());
  hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(ref_kernel, dim3((mi.size() + 63) / 64), dim3(64), 0, 0, dA, dW, dmi, dni, dref, K, (int)mi.size());
  hipMemcpy(ref.data(), dref, ref.size() * 4, hipMemcpyDeviceToHost);
  double maxerr = 0;
  if (!same) for (size_t t = 0; t < mi.size(); ++t) maxerr = fmax(maxerr, fabs(c[(size_t)mi[t] * N + ni[t]] - ref[t]) / (fabs(ref[t]) + 1.0));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); for (int i = 0; i < 10; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("gemm4w%s%s swz %d: M %d N %d K %d, %d tiles: %.1f us, %.0f TFLOP/s, max rel err vs fp64 reference on %zu samples %.2e\n", RP ? " +register prefetch (3 K-tiles of lead)" : "", same == 1 ? " [all blocks on tile (0,0)]" : same == 2 ? " [ablation: no operand DMA]" : same == 3 ? " [ablation: no DMA, no barrier]" : same == 4 ? " [ablation: MFMAs only]" : "", SWZ, M, N, K, blocks,
         ms * 100.0, 2.0 * M * N * K * 10 / (ms * 1e-3) / 1e12, mi.size(), maxerr);
}

int main(int argc, char** argv) {
  int M = 16384, N = 4096, K = 4096;           // 1024 tiles of 256 x 256 = 4 full rounds on 256 CUs, 64 K-tiles per tile
  if (argc >= 4) { M = atoi(argv[1]); N = atoi(argv[2]); K = atoi(argv[3]); }
  const bool constant = argc >= 5;               // 5th argument: constant operands instead of random ones (data-dependent power / clocks)
  std::vector<half_t> hA((size_t)M * K), hW((size_t)N * K);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : hA) v = constant ? (half_t)0.25f : (half_t)rnd();
  for (auto& v : hW) v = constant ? (half_t)0.03125f : (half_t)(rnd() * 0.1f);
  printf("operands: %s\n", constant ? "constant" : "random");
  half_t *dA, *dW; float *dC, *dref; int *dmi, *dni;
  hipMalloc(&dA, hA.size() * 2); hipMalloc(&dW, hW.size() * 2); hipMalloc(&dC, (size_t)M * N * 4);
  hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
  std::vector<int> mi, ni;
  for (int t = 0; t < 4096; ++t) { s = s * 1664525u + 1013904223u; mi.push_back((s >> 4) % M); s = s * 1664525u + 1013904223u; ni.push_back((s >> 4) % N); }
  hipMalloc(&dmi, mi.size() * 4); hipMalloc(&dni, ni.size() * 4); hipMalloc(&dref, mi.size() * 4);
  hipMemcpy(dmi, mi.data(), mi.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dni, ni.data(), ni.size() * 4, hipMemcpyHostToDevice);
  run<0>(0, dA, dW, dC, M, N, K, mi, ni, dmi, dni, dref);
  run<0, true>(0, dA, dW, dC, M, N, K, mi, ni, dmi, dni, dref);
  run<0, false, 2>(0, dA, dW, dC, M, N, K, mi, ni, dmi, dni, dref);
  run<0, false, 3>(0, dA, dW, dC, M, N, K, mi, ni, dmi, dni, dref);
  run<0, false, 4>(0, dA, dW, dC, M, N, K, mi, ni, dmi, dni, dref);
  return 0;
}
