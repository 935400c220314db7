// The reference's two native extensions, as gfx950 kernels (SURVEY.md section 8, row "next-4"):
//   * brute-force K nearest neighbours over padded point clouds  (reference: unidepth/ops/knn/src/knn.cu:27-251, knn_cpu.cpp:13-70;
//     consumer: utils/chamfer_distance.py:143-144 -> utils/evaluation_depth.py:12-34 chamfer / F1 metrics, always D = 3, K = 1)
//   * patch gather around integer centres  (reference: unidepth/ops/extract_patches/src/cuda/extract_patches_kernel.cu:65-95 behind
//     modules/patch_extractor.py:16-42; consumer: ops/losses/local_ssi.py:278-288)
// Forward only: the engine is inference / evaluation, the reference's backward kernels belong to training.
//
// KNN result definition (bit-exact, tested against the reference's own CPU implementation compiled from its sources):
//   dist(i, j) = sum_d (p1[i][d] - p2[j][d])^2   summed d = 0 .. D-1 in fp32, NOT contracted into FMAs (so that the value is the same
//   on every backend; the reference's nvcc build fuses, its CPU build does not -- we follow the one that can be run here);
//   the K kept neighbours are the K lexicographically smallest (dist, j) pairs, returned ascending -- what knn_cpu.cpp's
//   priority queue of (dist, index) tuples produces, and for K = 1 also what the CUDA MinK produces (first minimum wins a tie).
//   Rows i >= lengths1[n] and slots k >= lengths2[n] are zero (distance and index), as the reference pads.
//
// Roofline: VALU-bound (8 fp32 ops + compare/select per pair for D = 3; 12 B of p2 per point are broadcast from LDS to 256 queries).
// Layout: one query per lane kept in registers, the p2 cloud streamed through LDS in tiles of 1024 points padded to 4 (D <= 4) or
// 8 floats (D <= 8) so that one ds_read_b128 (broadcast, conflict-free) feeds a whole pair.  The K best are a sorted register list
// (fully unrolled insertion; no dynamic register indexing).  K = 1 can split P2 over blockIdx.z to fill 256 CUs when P1 is small:
// partial winners meet in a 64-bit atomicMin on (dist bits << 32 | j) -- distances are >= 0, so the integer order IS the
// lexicographic (dist, j) order.
#include "ud_common.h"

namespace {

// built with -ffp-contract=off (csrc/build.sh): the distance sums below must round the multiply and the add separately
#pragma clang fp contract(off)

constexpr int KNN_TILE = 1024;

template <int DC>
struct KnnTile {                       // floats per staged point
  static constexpr int STRIDE = DC == 3 ? 4 : DC;
};

// DC: dims computed per pair (3: D <= 3, 4: D == 4, 8: D <= 8, zero padded -- adding (0-0)^2 = +0 leaves the fp32 sum unchanged);
// DC == 0: any D <= 32, query re-read from LDS.  KT: capacity of the register list (K <= KT).
template <int DC, int KT>
__global__ __launch_bounds__(256) void knn_kernel(const UdKnn p, const int span) {
  constexpr int STRIDE = DC == 0 ? 1 : KnnTile<DC == 0 ? 4 : DC>::STRIDE;
  extern __shared__ float lds[];       // DC != 0: KNN_TILE * STRIDE floats;  DC == 0: tile_pts * D + 256 * D (queries)
  const int tid = threadIdx.x;
  const int n = blockIdx.y;
  const int q = blockIdx.x * 256 + tid;
  const int D = p.D;
  long long l1 = p.lengths1 ? p.lengths1[n] : (long long)p.P1;
  long long l2 = p.lengths2 ? p.lengths2[n] : (long long)p.P2;
  const int len1 = (int)(l1 < 0 ? 0 : (l1 > p.P1 ? p.P1 : l1));
  const int len2 = (int)(l2 < 0 ? 0 : (l2 > p.P2 ? p.P2 : l2));
  const bool active = q < len1;
  const int j_begin = blockIdx.z * span;
  const int j_end = min(len2, j_begin + span);
  const float* P1p = p.p1 + ((size_t)n * p.P1 + (active ? q : 0)) * D;
  const float* P2p = p.p2 + (size_t)n * p.P2 * D;

  float qv[DC == 0 ? 1 : DC];
  int tile_pts = KNN_TILE;
  float* qlds = nullptr;
  if constexpr (DC != 0) {
#pragma unroll
    for (int d = 0; d < DC; ++d) qv[d] = (active && d < D) ? P1p[d] : 0.0f;
  } else {
    tile_pts = 256;
    qlds = lds + tile_pts * D;
    for (int d = 0; d < D; ++d) qlds[d * 256 + tid] = active ? P1p[d] : 0.0f;     // [d][lane]: conflict-free per d
  }

  float dk[KT];
  int ik[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) { dk[k] = __builtin_inff(); ik[k] = 0; }

  for (int base = j_begin; base < j_end; base += tile_pts) {
    const int cnt = min(tile_pts, j_end - base);
    __syncthreads();
    if constexpr (DC != 0) {
      const float* src = P2p + (size_t)base * D;
      if (D == STRIDE) {
        for (int e = tid; e < cnt * STRIDE; e += 256) lds[e] = src[e];
      } else {
        for (int e = tid; e < cnt * STRIDE; e += 256) {
          const int pt = e / STRIDE, d = e - pt * STRIDE;
          lds[e] = d < D ? src[pt * D + d] : 0.0f;
        }
      }
    } else {
      const float* src = P2p + (size_t)base * D;
      for (int e = tid; e < cnt * D; e += 256) lds[e] = src[e];
    }
    __syncthreads();
    if (!active) continue;
#pragma unroll 4
    for (int j = 0; j < cnt; ++j) {
      float dist;
      if constexpr (DC == 0) {
        dist = 0.0f;
        for (int d = 0; d < D; ++d) {
          const float diff = qlds[d * 256 + tid] - lds[j * D + d];
          dist += p.norm == 2 ? diff * diff : fabsf(diff);
        }
      } else {
        float pv[STRIDE];
        if constexpr (STRIDE == 4) {
          const f32x4 v = *(const f32x4*)(lds + j * 4);
          pv[0] = v[0]; pv[1] = v[1]; pv[2] = v[2]; pv[3] = v[3];
        } else {
          const f32x4 v0 = *(const f32x4*)(lds + j * 8), v1 = *(const f32x4*)(lds + j * 8 + 4);
          pv[0] = v0[0]; pv[1] = v0[1]; pv[2] = v0[2]; pv[3] = v0[3]; pv[4] = v1[0]; pv[5] = v1[1]; pv[6] = v1[2]; pv[7] = v1[3];
        }
        const float d0 = qv[0] - pv[0];
        dist = p.norm == 2 ? d0 * d0 : fabsf(d0);
#pragma unroll
        for (int d = 1; d < DC; ++d) {
          const float diff = qv[d] - pv[d];
          dist += p.norm == 2 ? diff * diff : fabsf(diff);
        }
      }
      const int jj = base + j;
      if (dist < dk[KT - 1]) {
#pragma unroll
        for (int k = KT - 1; k > 0; --k) {
          if (dist < dk[k - 1]) { dk[k] = dk[k - 1]; ik[k] = ik[k - 1]; }
          else if (dist < dk[k]) { dk[k] = dist; ik[k] = jj; }
        }
        if (dist < dk[0]) { dk[0] = dist; ik[0] = jj; }
      }
    }
  }

  if (q >= p.P1) return;
  if (gridDim.z > 1) {                 // K == 1 split: merge through the packed atomic (p.work pre-filled with ~0)
    if (active && dk[0] < __builtin_inff()) {
      const unsigned long long key = ((unsigned long long)__float_as_uint(dk[0]) << 32) | (unsigned)ik[0];
      atomicMin(p.work + (size_t)n * p.P1 + q, key);
    }
    return;
  }
  const int kv = active ? min(p.K, len2) : 0;
  float* od = p.dists + ((size_t)n * p.P1 + q) * p.K;
  long long* oi = p.idx + ((size_t)n * p.P1 + q) * p.K;
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    if (k < p.K) {
      od[k] = k < kv ? dk[k] : 0.0f;
      oi[k] = k < kv ? (long long)ik[k] : 0ll;
    }
  }
}

// K = 1, D <= 3 (the Chamfer / F1 metrics): two queries per lane so that the distance arithmetic is packed fp32 (v_pk_add/mul_f32: 8
// packed ops per 2 pairs), and the running minimum is kept WITHOUT its index (one v_min3_f32 per query per two points); the index is
// recovered only for the rare 8-point chunk whose minimum beats the best so far, by re-running that chunk with the strict-<
// update (same instructions, so bit-identical distances; first minimum still wins).  ~5.4 issue slots per pair instead of 11.
constexpr int KNN1_CHUNK = 8;

__device__ __forceinline__ f32x2 knn1_dist(const f32x2 qx, const f32x2 qy, const f32x2 qz, const f32x4 v, const bool l2) {
  const f32x2 dx = qx - (f32x2){v[0], v[0]}, dy = qy - (f32x2){v[1], v[1]}, dz = qz - (f32x2){v[2], v[2]};
  if (l2) return (dx * dx + dy * dy) + dz * dz;
  return ((f32x2){fabsf(dx[0]), fabsf(dx[1])} + (f32x2){fabsf(dy[0]), fabsf(dy[1])}) + (f32x2){fabsf(dz[0]), fabsf(dz[1])};
}

template <bool L2>
__global__ __launch_bounds__(256) void knn1_d3_kernel(const UdKnn p, const int span) {
  __shared__ f32x4 tile[KNN_TILE];
  const int tid = threadIdx.x;
  const int n = blockIdx.y;
  const int q0 = blockIdx.x * 512 + tid, q1 = q0 + 256;
  const int D = p.D;
  long long l1 = p.lengths1 ? p.lengths1[n] : (long long)p.P1;
  long long l2 = p.lengths2 ? p.lengths2[n] : (long long)p.P2;
  const int len1 = (int)(l1 < 0 ? 0 : (l1 > p.P1 ? p.P1 : l1));
  const int len2 = (int)(l2 < 0 ? 0 : (l2 > p.P2 ? p.P2 : l2));
  const bool a0 = q0 < len1, a1 = q1 < len1;
  const int j_begin = blockIdx.z * span;
  const int j_end = min(len2, j_begin + span);
  const float* P1n = p.p1 + (size_t)n * p.P1 * D;
  const float* P2p = p.p2 + (size_t)n * p.P2 * D;
  float c0[3], c1[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    c0[d] = (a0 && d < D) ? P1n[(size_t)q0 * D + d] : 0.0f;
    c1[d] = (a1 && d < D) ? P1n[(size_t)q1 * D + d] : 0.0f;
  }
  const f32x2 qx = {c0[0], c1[0]}, qy = {c0[1], c1[1]}, qz = {c0[2], c1[2]};
  float b0 = __builtin_inff(), b1 = __builtin_inff();
  int i0 = 0, i1 = 0;

  for (int base = j_begin; base < j_end; base += KNN_TILE) {
    const int cnt = min(KNN_TILE, j_end - base);
    const int cnt8 = (cnt + KNN1_CHUNK - 1) & ~(KNN1_CHUNK - 1);
    __syncthreads();
    for (int e = tid; e < cnt8; e += 256) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (e < cnt) {
        const float* src = P2p + (size_t)(base + e) * D;
        v[0] = src[0];
        if (D > 1) v[1] = src[1];
        if (D > 2) v[2] = src[2];
      } else {
        v[0] = __builtin_inff();         // chunk padding: distance +inf, never a winner
      }
      tile[e] = v;
    }
    __syncthreads();
    if (!(a0 || a1)) continue;
    for (int c = 0; c < cnt8; c += KNN1_CHUNK) {
      float m0 = b0, m1 = b1;
#pragma unroll
      for (int u = 0; u < KNN1_CHUNK; u += 2) {
        const f32x2 da = knn1_dist(qx, qy, qz, tile[c + u], L2), db = knn1_dist(qx, qy, qz, tile[c + u + 1], L2);
        m0 = fminf(fminf(m0, da[0]), db[0]);
        m1 = fminf(fminf(m1, da[1]), db[1]);
      }
      if (m0 < b0 || m1 < b1) {
#pragma unroll
        for (int u = 0; u < KNN1_CHUNK; ++u) {
          const f32x2 d = knn1_dist(qx, qy, qz, tile[c + u], L2);
          if (d[0] < b0) { b0 = d[0]; i0 = base + c + u; }
          if (d[1] < b1) { b1 = d[1]; i1 = base + c + u; }
        }
      }
    }
  }

  const bool split = gridDim.z > 1;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int q = h ? q1 : q0;
    const bool act = h ? a1 : a0;
    const float b = h ? b1 : b0;
    const int bi = h ? i1 : i0;
    if (q >= p.P1) continue;
    const size_t o = (size_t)n * p.P1 + q;
    if (split) {
      if (act && b < __builtin_inff()) atomicMin(p.work + o, ((unsigned long long)__float_as_uint(b) << 32) | (unsigned)bi);
    } else {
      const bool ok = act && len2 > 0;
      p.dists[o] = ok ? b : 0.0f;
      p.idx[o] = ok ? (long long)bi : 0ll;
    }
  }
}

__global__ void knn_fill_kernel(unsigned long long* w, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) w[i] = ~0ull;
}

__global__ void knn_unpack_kernel(const unsigned long long* w, float* dists, long long* idx, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long v = w[i];
  const bool none = v == ~0ull;
  dists[i] = none ? 0.0f : __uint_as_float((unsigned)(v >> 32));
  idx[i] = none ? 0ll : (long long)(unsigned)(v & 0xffffffffull);
}

template <int DC>
int knn_launch_k(const UdKnn& d, dim3 grid, int span, size_t lds_bytes, hipStream_t s) {
#define UD_KNN_GO(KT) hipLaunchKernelGGL((knn_kernel<DC, KT>), grid, dim3(256), lds_bytes, s, d, span)
  if (d.K == 1) UD_KNN_GO(1);
  else if (d.K == 2) UD_KNN_GO(2);
  else if (d.K <= 4) UD_KNN_GO(4);
  else if (d.K <= 8) UD_KNN_GO(8);
  else if (d.K <= 16) UD_KNN_GO(16);
  else UD_KNN_GO(32);
#undef UD_KNN_GO
  return 0;
}

// out[b][n][c][i][j] = in[b][c][cy - h/2 + i - pad_h][cx - w/2 + j - pad_w], zero outside the image: the zero padding the reference's
// module materialises with F.pad (patch_extractor.py:26-37) is a bounds test here, so no padded copy of the image is made.
__global__ __launch_bounds__(256) void extract_patches_kernel(const UdExtractPatches p, const long long total) {
  const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
  if (o >= total) return;
  const int j = (int)(o % p.w);
  long long r = o / p.w;
  const int i = (int)(r % p.h); r /= p.h;
  const int c = (int)(r % p.C); r /= p.C;
  const int n = (int)(r % p.N);
  const int b = (int)(r / p.N);
  const int cy = p.centers[((size_t)b * p.N + n) * 2], cx = p.centers[((size_t)b * p.N + n) * 2 + 1];
  const int y = cy - p.h / 2 + i - p.pad_h, x = cx - p.w / 2 + j - p.pad_w;
  float v = 0.0f;
  if (y >= 0 && y < p.H && x >= 0 && x < p.W) v = p.in[(((size_t)b * p.C + c) * p.H + y) * p.W + x];
  p.out[o] = v;
}

}  // namespace

extern "C" int ud_knn_split(const UdKnn* desc) {
  // number of P2 slices the K == 1 search is cut into (1 = no workspace needed)
  const UdKnn& d = *desc;
  if (d.K != 1 || !d.work || d.P2 <= 0) return 1;
  const int per = d.D <= 3 ? 512 : 256;       // queries per block (two per lane on the packed D <= 3 path)
  const long long qblocks = (long long)((d.P1 + per - 1) / per) * d.N;
  if (qblocks >= 1024) return 1;
  long long s = (1024 + qblocks - 1) / qblocks;
  const long long smax = (d.P2 + 2 * KNN_TILE - 1) / (2 * KNN_TILE);
  if (s > smax) s = smax;
  if (s > 65535) s = 65535;
  return (int)(s < 1 ? 1 : s);
}

extern "C" int ud_knn_points(const UdKnn* desc, void* stream) {
  const UdKnn& d = *desc;
  if (!d.p1 || (!d.p2 && d.P2 > 0) || !d.dists || !d.idx || d.N <= 0 || d.P1 <= 0 || d.P2 < 0 || d.D < 1 || d.D > 32 || d.K < 1 || d.K > 32 ||
      (d.norm != 1 && d.norm != 2) || d.N > 65535) {
    ud_set_error("ud_knn_points: bad argument (1 <= D <= 32, 1 <= K <= 32, norm 1|2, N <= 65535)");
    return UD_ERR_BAD_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  const int S = ud_knn_split(desc);
  const int span = S > 1 ? (int)(((long long)d.P2 + S - 1) / S) : (d.P2 > 0 ? d.P2 : 1);
  const bool fast1 = d.K == 1 && d.D <= 3;
  dim3 grid(fast1 ? (d.P1 + 511) / 512 : (d.P1 + 255) / 256, d.N, S);
  const size_t nq = (size_t)d.N * d.P1;
  if (S > 1) hipLaunchKernelGGL(knn_fill_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, d.work, nq);
  if (fast1) {
    if (d.norm == 2) hipLaunchKernelGGL((knn1_d3_kernel<true>), grid, dim3(256), 0, s, d, span);
    else hipLaunchKernelGGL((knn1_d3_kernel<false>), grid, dim3(256), 0, s, d, span);
  } else if (d.D <= 3) knn_launch_k<3>(d, grid, span, KNN_TILE * 4 * sizeof(float), s);
  else if (d.D == 4) knn_launch_k<4>(d, grid, span, KNN_TILE * 4 * sizeof(float), s);
  else if (d.D <= 8) knn_launch_k<8>(d, grid, span, KNN_TILE * 8 * sizeof(float), s);
  else knn_launch_k<0>(d, grid, span, (size_t)(256 * d.D) * 2 * sizeof(float), s);
  if (S > 1) hipLaunchKernelGGL(knn_unpack_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, d.work, d.dists, d.idx, nq);
  UD_CHECK_LAUNCH("ud_knn_points launch");
  return UD_OK;
}

extern "C" int ud_extract_patches(const UdExtractPatches* desc, void* stream) {
  const UdExtractPatches& d = *desc;
  if (d.B < 0 || d.C < 0 || d.H < 0 || d.W < 0 || d.N < 0 || d.h < 0 || d.w < 0) {
    ud_set_error("ud_extract_patches: negative dimension");
    return UD_ERR_BAD_ARG;
  }
  const long long total = (long long)d.B * d.N * d.C * d.h * d.w;
  if (total == 0) return UD_OK;          // empty outputs may come with null pointers
  if (!d.in || !d.out || !d.centers) {
    ud_set_error("ud_extract_patches: null pointer");
    return UD_ERR_BAD_ARG;
  }
  hipLaunchKernelGGL(extract_patches_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d, total);
  UD_CHECK_LAUNCH("ud_extract_patches launch");
  return UD_OK;
}
