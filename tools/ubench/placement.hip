// Workgroup placement probe (gfx950, round 6): where does the dispatcher put the workgroups of a one-round launch whose workgroups are sized so
// that exactly TWO fit a CU (4 waves, 80 KB of LDS, <= 256 VGPRs)?  Every workgroup records HW_REG_HW_ID, HW_REG_XCC_ID, its start / end
// s_memrealtime and spins for ~20 us.  The host prints: distinct CUs used, workgroups per CU, how many pairs are co-resident (overlapping
// lifetimes), which block indices share a CU, and the SIMDs the four waves of a workgroup land on.
// This decides the design of the two-workgroups-per-CU GEMM (csrc/gemm_duo.hip): whether both members of a pair are resident together, and
// whether a static rule on blockIdx identifies the pair (else: one atomic per workgroup on a per-CU counter).
// build: hipcc --offload-arch=gfx950 -O3 -o placement placement.hip ; run: ./placement [workgroups=464] [lds_bytes=81920]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>

struct Rec { unsigned hwid, xcc; unsigned long long t0, t1; unsigned simd[4]; };

__global__ __launch_bounds__(256) void probe(Rec* out, int spin_ticks) {
  extern __shared__ char smem[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  smem[threadIdx.x] = (char)threadIdx.x;         // touch the allocation
  unsigned long long t1 = t0;
  while ((long long)(t1 - t0) < spin_ticks) t1 = __builtin_amdgcn_s_memrealtime();
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x].simd[wv] = (hw >> 4) & 3;
  if (threadIdx.x == 0) {
    out[blockIdx.x].hwid = hw;
    out[blockIdx.x].xcc = xcc & 15;
    out[blockIdx.x].t0 = t0;
    out[blockIdx.x].t1 = t1;
  }
}

int main(int argc, char** argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 464;
  const int lds = argc > 2 ? atoi(argv[2]) : 81920;
  Rec* d;
  hipMalloc(&d, sizeof(Rec) * nwg);
  if (hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) { printf("cannot reserve %d B of LDS\n", lds); return 1; }
  std::vector<Rec> h(nwg);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(d, 0, sizeof(Rec) * nwg);
    hipLaunchKernelGGL(probe, dim3(nwg), dim3(256), lds, 0, d, 2000);      // s_memrealtime ticks at 100 MHz: 2000 ticks = 20 us
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, sizeof(Rec) * nwg, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;
    unsigned long long tmin = ~0ull;
    for (int i = 0; i < nwg; ++i) { cu[(h[i].xcc << 8) | ((h[i].hwid >> 8) & 0xff)].push_back(i); tmin = std::min(tmin, h[i].t0); }
    int hist[8] = {0}, copairs = 0, pairs = 0, same_par = 0, d32 = 0;
    double latest = 0;
    for (auto& kv : cu) {
      auto& v = kv.second;
      hist[std::min<size_t>(v.size(), 7)]++;
      for (int i : v) latest = std::max(latest, (double)(h[i].t0 - tmin) / 100.0);
      if (v.size() == 2) {
        ++pairs;
        const Rec &a = h[v[0]], &b = h[v[1]];
        if (a.t0 < b.t1 && b.t0 < a.t1) ++copairs;
        if (((v[0] >> 3) & 1) == ((v[1] >> 3) & 1)) ++same_par;
        if (std::abs((v[0] >> 3) - (v[1] >> 3)) == 32 || std::abs((v[0] >> 3) - (v[1] >> 3)) == 1) ++d32;
      }
    }
    printf("rep %d: %d workgroups x %d B LDS: %zu distinct (xcc, se, sh, cu); CUs with 1 / 2 / 3+ workgroups: %d / %d / %d; co-resident pairs %d of %d; latest start %.1f us after the first\n",
           rep, nwg, lds, cu.size(), hist[1], hist[2], hist[3] + hist[4] + hist[5] + hist[6] + hist[7], copairs, pairs, latest);
    printf("        pairs with equal parity of the in-XCD index (blockIdx >> 3): %d; pairs whose in-XCD indices differ by 1 or 32: %d\n", same_par, d32);
    if (rep == 0) {
      int shown = 0;
      for (auto& kv : cu) {
        if (shown++ >= 12) break;
        printf("        xcc %u cu-key 0x%02x:", kv.first >> 8, kv.first & 0xff);
        for (int i : kv.second) printf("  block %d (in-XCD %d, start +%.2f us, SIMDs %u%u%u%u)", i, i >> 3, (double)(h[i].t0 - tmin) / 100.0, h[i].simd[0], h[i].simd[1], h[i].simd[2], h[i].simd[3]);
        printf("\n");
      }
    }
  }
  return 0;
}
