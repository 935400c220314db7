// Cost of exchanging fp32 partial tiles between workgroups on DIFFERENT XCDs inside one kernel (the building block of
// split-K / stream-K joins): 256 workgroups x 256 threads, each holds a 128 x 128 fp32 partial (64 KB, 64 floats per thread).
// Pairs are blocks (b, b ^ 1): consecutive block ids sit on different XCDs (block b -> XCD b % 8), i.e. different, mutually
// non-coherent L2s.  Modes:
//   0  write own partial, no fence                                      (store cost alone)
//   1  write + __threadfence()                                          (agent-scope release: L2 write-back)
//   2  write + fence + ticket (atomicAdd); the LAST arriver of the pair fences (acquire), reads the partner's partial and checks it
//   3  as 2 without any fence: partials are written and read with system-scope cache policy (`sc0 sc1`: write-through /
//      L2-bypassing, the memory-side Infinity Cache is coherent), ordered by s_waitcnt vmcnt(0) around the ticket atomic
// Prints microseconds per launch and, for mode 2, the number of mismatching floats (must be 0).
// build: hipcc --offload-arch=gfx950 -O3 -o partial_exchange partial_exchange.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>
__global__ __launch_bounds__(256) void exch(float* ws, unsigned* tickets, unsigned* bad, int round, int spin) {
  const int b = blockIdx.x, t = threadIdx.x;
  __shared__ unsigned ticket;
  // stand-in for the K loop: a little ALU work so that arrival times differ between partners
  float v = (float)(b * 256 + t) + (float)round;
  for (int i = 0; i < spin * (1 + (b & 3)); ++i) v = __builtin_fmaf(v, 1.0000001f, 0.0f);
  f32x4* mine = (f32x4*)(ws + (size_t)b * 16384);
  if (MODE == 3) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const f32x4 val = {(float)(b + round), (float)t, (float)i, 1.0f};
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 2" ::"v"(mine + i * 256 + t), "v"(val) : "memory");   // s_nop: the data VGPRs of a 16-byte store must not be rewritten in the next cycles (the compiler only guards its own stores)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) ticket = atomicAdd(&tickets[b >> 1], 1u);
    __syncthreads();
    if (ticket & 1u) {
      const f32x4* other = (const f32x4*)(ws + (size_t)(b ^ 1) * 16384);
      f32x4 o[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(o[i]) : "v"(other + i * 256 + t) : "memory");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      unsigned wrong = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        asm volatile("" : "+v"(o[i]));
        wrong += (o[i][0] != (float)((b ^ 1) + round)) + (o[i][1] != (float)t) + (o[i][2] != (float)i) + (o[i][3] != 1.0f);
      }
      if (wrong) atomicAdd(bad, wrong);
    }
    if (v == 12345.678f) ws[0] = v;
    return;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) mine[i * 256 + t] = (f32x4){(float)(b + round), (float)t, (float)i, 1.0f};
  if (MODE >= 1) __threadfence();
  if (MODE >= 2) {
    __syncthreads();
    if (t == 0) ticket = atomicAdd(&tickets[b >> 1], 1u);
    __syncthreads();
    if (ticket & 1u) {                                     // second arriver of the pair (tickets are never reset: parity)
      __threadfence();
      const f32x4* other = (const f32x4*)(ws + (size_t)(b ^ 1) * 16384);
      unsigned wrong = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const f32x4 o = other[i * 256 + t];
        wrong += (o[0] != (float)((b ^ 1) + round)) + (o[1] != (float)t) + (o[2] != (float)i) + (o[3] != 1.0f);
      }
      if (wrong) atomicAdd(bad, wrong);
    }
  }
  if (v == 12345.678f) ws[0] = v;                          // keep the ALU loop alive
}

template <int MODE>
float run(float* ws, unsigned* tk, unsigned* bad, int spin) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(exch<MODE>, dim3(256), dim3(256), 0, 0, ws, tk, bad, i, spin);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(exch<MODE>, dim3(256), dim3(256), 0, 0, ws, tk, bad, 3 + i, spin);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.0f / 50;
}

int main() {
  float* ws; unsigned *tk, *bad;
  hipMalloc(&ws, 256 * 16384 * 4); hipMalloc(&tk, 128 * 4); hipMalloc(&bad, 4);
  hipMemset(tk, 0, 128 * 4); hipMemset(bad, 0, 4);
  {  // the fence-free variant on differently-typed workspaces: ordinary (coarse-grained, L2-cached per XCD), fine-grained, uncached
    const unsigned flags[3] = {0, hipDeviceMallocFinegrained, hipDeviceMallocUncached};
    const char* names[3] = {"hipMalloc", "fine-grained", "uncached"};
    for (int k = 0; k < 3; ++k) {
      float* w2 = nullptr;
      if ((k == 0 ? hipMalloc(&w2, 256 * 16384 * 4) : hipExtMallocWithFlags((void**)&w2, 256 * 16384 * 4, flags[k])) != hipSuccess) { printf("%s: alloc failed\n", names[k]); continue; }
      hipMemset(tk, 0, 128 * 4); hipMemset(bad, 0, 4);
      const float t0 = run<0>(w2, tk, bad, 0);
      hipMemset(tk, 0, 128 * 4);
      const float t3 = run<3>(w2, tk, bad, 0);
      unsigned h = 0; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
      printf("%-12s workspace: store only %.1f us, fence-free exchange %.1f us, mismatches %u\n", names[k], t0, t3, h);
      hipFree(w2);
    }
    hipMemset(tk, 0, 128 * 4); hipMemset(bad, 0, 4);
  }
  for (int spin : {0, 2000}) {
    const float t0 = run<0>(ws, tk, bad, spin), t1 = run<1>(ws, tk, bad, spin), t2 = run<2>(ws, tk, bad, spin);
    hipMemset(tk, 0, 128 * 4);
    const float t3 = run<3>(ws, tk, bad, spin);
    unsigned h = 0; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("spin %4d: store only %.1f us, + release fence %.1f us, + ticket, acquire, partner read %.1f us, fence-free sc0 sc1 variant %.1f us per launch (16 MB of partials); mismatches %u\n",
           spin, t0, t1, t2, t3, h);
  }
  return 0;
}
