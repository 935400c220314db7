// What do rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the access patterns of the large-tile GEMM?
// MI355X_MICROARCH.md (HBM): FETCH_SIZE = half the bytes of a wide coalesced streaming read, "other access widths and WRITE_SIZE are
// uncalibrated: calibrate on a known byte count in your own access pattern".  The patterns here, each over a buffer of known size read or
// written exactly once (512 MiB: past the 256 MiB Infinity Cache):
//   rd_wide     16 B per lane, 1 KiB contiguous per wave instruction                     (the guide's calibrated case)
//   rd_seg64    16 B per lane, 16 rows x 64 B per wave instruction, rows 4 KiB apart     (gemm256_kernel: fp32 residual -> accumulators)
//   rd_seg128   16 B per lane, 8 rows x 128 B per wave instruction, rows 2 KiB apart     (gemm256_kernel: operand K-tiles, [row][64 halves])
//   rd_lds128   the same 8 x 128 B shape through buffer_load ... lds                      (what the operand loader really issues)
//   wr_wide     16 B per lane, 1 KiB contiguous per wave instruction
//   wr_seg64    16 rows x 64 B per wave instruction, rows 4 KiB apart                    (gemm256_kernel: fp32 output rows from the accumulators)
//   wr_seg64h   the same 16 x 64 B shape with rows 2 KiB apart                           (the fp16 copy written beside the fp32 stream)
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/fetch_calib.hip -o tools/ubench/fetch_calib
// Run:   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/f -o f -- tools/ubench/fetch_calib   (and --pmc WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr size_t BYTES = 512ull << 20;

// every kernel: grid of 2048 workgroups x 256 threads, each wave walks its share; `sink` keeps the loads alive
__global__ __launch_bounds__(256) void rd_wide(const f32x4* __restrict__ p, float* sink, size_t n16) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) a += p[i];
  if (a[0] + a[1] + a[2] + a[3] == 1234.5f) *sink = a[0];
}

// matrix of fp32 [rows][1024] (4 KiB rows); a wave instruction covers rows r0 .. r0+15, 64 B (16 floats) of columns: lane l -> row l & 15, 16-byte piece l >> 4
__global__ __launch_bounds__(256) void rd_seg64(const float* __restrict__ p, float* sink, int rows) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int waves = gridDim.x * 4, w = blockIdx.x * 4 + wv;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  // units: (row block of 16) x (64-byte column piece: 64 per row)
  const long units = (long)(rows / 16) * 64;
  for (long u = w; u < units; u += waves) {
    const long rb = u / 64, cp = u % 64;
    a += *(const f32x4*)(p + (size_t)(rb * 16 + (lane & 15)) * 1024 + cp * 16 + 4 * (lane >> 4));
  }
  if (a[0] + a[1] + a[2] + a[3] == 1234.5f) *sink = a[0];
}

// matrix of 2-byte elements [rows][1024] (2 KiB rows); a wave instruction covers 8 rows x 128 B: lane l -> row l >> 3, 16-byte piece l & 7
__global__ __launch_bounds__(256) void rd_seg128(const char* __restrict__ p, float* sink, int rows) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int waves = gridDim.x * 4, w = blockIdx.x * 4 + wv;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  const long units = (long)(rows / 8) * 16;          // 16 pieces of 128 B per row
  for (long u = w; u < units; u += waves) {
    const long rb = u / 16, cp = u % 16;
    a += *(const f32x4*)(p + (size_t)(rb * 8 + (lane >> 3)) * 2048 + cp * 128 + 16 * (lane & 7));
  }
  if (a[0] + a[1] + a[2] + a[3] == 1234.5f) *sink = a[0];
}

__global__ __launch_bounds__(256) void rd_lds128(const char* __restrict__ p, float* sink, int rows) {
  __shared__ __attribute__((aligned(16))) char lds[4 * 1024];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int waves = gridDim.x * 4, w = blockIdx.x * 4 + wv;
  const long units = (long)(rows / 8) * 16;
  for (long u = w; u < units; u += waves) {
    const long rb = u / 16, cp = u % 16;
    const char* src = p + (size_t)(rb * 8 + (lane >> 3)) * 2048 + cp * 128 + 16 * (lane & 7);
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src, (void __attribute__((address_space(3)))*)(lds + wv * 1024), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (((float*)lds)[threadIdx.x] == 1234.5f) *sink = 1.f;
}

__global__ __launch_bounds__(256) void wr_wide(f32x4* __restrict__ p, size_t n16) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) p[i] = v;
}

__global__ __launch_bounds__(256) void wr_seg64(float* __restrict__ p, int rows) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int waves = gridDim.x * 4, w = blockIdx.x * 4 + wv;
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  const long units = (long)(rows / 16) * 64;
  for (long u = w; u < units; u += waves) {
    const long rb = u / 64, cp = u % 64;
    *(f32x4*)(p + (size_t)(rb * 16 + (lane & 15)) * 1024 + cp * 16 + 4 * (lane >> 4)) = v;
  }
}

// the fp16 copy of the same accumulator rows (after the lane exchange of the epilogue): 16 B per lane, 16 rows x 64 B, rows 2 KiB apart
__global__ __launch_bounds__(256) void wr_seg64h(char* __restrict__ p, int rows) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int waves = gridDim.x * 4, w = blockIdx.x * 4 + wv;
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  const long units = (long)(rows / 16) * 32;          // 32 pieces of 64 B per 2 KiB row
  for (long u = w; u < units; u += waves) {
    const long rb = u / 32, cp = u % 32;
    *(f32x4*)(p + (size_t)(rb * 16 + (lane & 15)) * 2048 + cp * 64 + 16 * (lane >> 4)) = v;
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  char* buf;
  float* sink;
  CK(hipMalloc(&buf, BYTES));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(buf, 0, BYTES));
  const int G = 2048;
  const int rows4k = (int)(BYTES / 4096), rows2k = (int)(BYTES / 2048);
  for (int rep = 0; rep < 2; ++rep) {
    rd_wide<<<G, 256>>>((const f32x4*)buf, sink, BYTES / 16);
    rd_seg64<<<G, 256>>>((const float*)buf, sink, rows4k);
    rd_seg128<<<G, 256>>>(buf, sink, rows2k);
    rd_lds128<<<G, 256>>>(buf, sink, rows2k);
    wr_wide<<<G, 256>>>((f32x4*)buf, BYTES / 16);
    wr_seg64<<<G, 256>>>((float*)buf, rows4k);
    wr_seg64h<<<G, 256>>>(buf, rows2k);
  }
  CK(hipDeviceSynchronize());
  printf("each kernel touches %zu bytes exactly once per launch\n", BYTES);
  return 0;
}
