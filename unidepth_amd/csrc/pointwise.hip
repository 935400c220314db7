// HBM-bound / latency-bound kernels of the infer() path: pre-processing + im2col, camera tail, ray generation,
// ray angle embedding (+LayerNorm statistics), bilinear resamplers, output assembly, layout change.
// All are coalesced along the fastest-varying dimension, 8/16-byte accesses where layout allows, wave64 reductions.
#include "ud_common.h"
#include "camera_models.h"

namespace {

// ------------------------------------------------------------------------------------------------ preprocess
// one thread per (patch row m, k) element; k fastest -> coalesced 2-byte stores, near-coalesced source reads.
__global__ __launch_bounds__(256) void preprocess_kernel(const UdPreprocess p) {
  const int hgrid = p.Hn / 14, wgrid = p.Wn / 14;
  const long long total = (long long)p.B * hgrid * wgrid * 588;
  const float sy = (float)p.Hp / (float)p.Hn, sx = (float)p.Wp / (float)p.Wn;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int k = (int)(idx % 588);
    const long long m = idx / 588;
    const int px = (int)(m % wgrid);
    const int py = (int)((m / wgrid) % hgrid);
    const int img = (int)(m / ((long long)wgrid * hgrid));
    const int c = k / 196, ij = k - c * 196;
    const int i = ij / 14, j = ij - i * 14;
    const int oy = py * 14 + i, ox = px * 14 + j;
    // bilinear, align_corners=False, source = zero-padded normalised image [Hp, Wp]
    float fy = sy * ((float)oy + 0.5f) - 0.5f;
    float fx = sx * ((float)ox + 0.5f) - 0.5f;
    fy = fy < 0.0f ? 0.0f : fy;
    fx = fx < 0.0f ? 0.0f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < p.Hp - 1), x1 = x0 + (x0 < p.Wp - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    auto sample = [&](int yy, int xx) -> float {
      yy -= p.pad_t;
      xx -= p.pad_l;
      if ((unsigned)yy >= (unsigned)p.H || (unsigned)xx >= (unsigned)p.W) return 0.0f;   // pad = 0 after normalisation
      const size_t off = (((size_t)img * 3 + c) * p.H + yy) * p.W + xx;
      float v = p.is_u8 ? (float)((const unsigned char*)p.rgb)[off] : ((const float*)p.rgb)[off];
      if (p.normalize) v = (v / 255.0f - p.mean[c]) * p.inv_std[c];
      return v;
    };
    float v;
    if (ly == 0.0f && lx == 0.0f) {
      v = sample(y0, x0);
    } else {
      const float v00 = sample(y0, x0), v01 = sample(y0, x1), v10 = sample(y1, x0), v11 = sample(y1, x1);
      v = (1.0f - ly) * ((1.0f - lx) * v00 + lx * v01) + ly * ((1.0f - lx) * v10 + lx * v11);
    }
    ((half_t*)p.patches)[m * p.ldp + k] = (half_t)v;
  }
}

__global__ void fill_rows_kernel(float* dst, const float* src, int n_img, int rows_per_img, int row_off, int D, int ld) {
  const int img = blockIdx.y;
  for (int c = blockIdx.x * 256 + threadIdx.x; c < D; c += gridDim.x * 256)
    dst[((size_t)img * rows_per_img + row_off) * ld + c] = src[c];
}

// ------------------------------------------------------------------------------------------------ camera tail
__global__ void camera_kernel(const float* raw, int raw_stride, float* intr4, float* K33, float* Kinv33, float* Kpost33, int B,
                              int Hn, int Wn, float rf, int pad_l, int pad_t) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const float* o = raw + (size_t)b * 4 * raw_stride;
  const float diag = sqrtf((float)Hn * (float)Hn + (float)Wn * (float)Wn);
  const float fx = expf(o[0]) * (0.7f * diag);
  const float fy = expf(o[raw_stride]) * (0.7f * diag);
  const float cx = (1.0f / (1.0f + expf(-o[2 * raw_stride]))) * (float)Wn;
  const float cy = (1.0f / (1.0f + expf(-o[3 * raw_stride]))) * (float)Hn;
  intr4[b * 4 + 0] = fx; intr4[b * 4 + 1] = fy; intr4[b * 4 + 2] = cx; intr4[b * 4 + 3] = cy;
  float* K = K33 + b * 9;
  K[0] = fx; K[1] = 0.f; K[2] = cx; K[3] = 0.f; K[4] = fy; K[5] = cy; K[6] = 0.f; K[7] = 0.f; K[8] = 1.f;
  float* Ki = Kinv33 + b * 9;
  Ki[0] = 1.0f / fx; Ki[1] = 0.f; Ki[2] = -cx / fx; Ki[3] = 0.f; Ki[4] = 1.0f / fy; Ki[5] = -cy / fy; Ki[6] = 0.f; Ki[7] = 0.f; Ki[8] = 1.f;
  float* Kp = Kpost33 + b * 9;
  Kp[0] = fx / rf; Kp[1] = 0.f; Kp[2] = cx / rf - (float)pad_l; Kp[3] = 0.f; Kp[4] = fy / rf; Kp[5] = cy / rf - (float)pad_t;
  Kp[6] = 0.f; Kp[7] = 0.f; Kp[8] = 1.f;
}

__global__ __launch_bounds__(256) void rays_kernel(const float* Kinv33, float* rays, int nb, int Hn, int Wn, int gt_mode) {
  const int b = blockIdx.y;
  const float* Ki = Kinv33 + b * 9;
  const float k0 = Ki[0], k1 = Ki[1], k2 = Ki[2], k3 = Ki[3], k4 = Ki[4], k5 = Ki[5], k6 = Ki[6], k7 = Ki[7], k8 = Ki[8];
  const int HW = Hn * Wn;
  for (int pix = blockIdx.x * 256 + threadIdx.x; pix < HW; pix += gridDim.x * 256) {
    const int v = pix / Wn, u = pix - v * Wn;
    const float uf = (float)u + 0.5f, vf = (float)v + 0.5f;
    float x = k0 * uf + k1 * vf + k2;
    float y = k3 * uf + k4 * vf + k5;
    float z = k6 * uf + k7 * vf + k8;
    float nmin = 1e-5f;
    if (gt_mode == 1) {
      const float zc = fmaxf(z, 1e-4f);
      x /= zc; y /= zc; z /= zc;
      nmin = 1e-4f;
    } else if (gt_mode == 2) {
      // EUCM (utils/camera.py:307-328): k0..k5 = fx, fy, cx, cy, alpha, beta at network resolution
      const float mx = (uf - k2) / k0, my = (vf - k3) / k1;
      const float r2 = mx * mx + my * my;
      const float sv = 1.0f - (2.0f * k4 - 1.0f) * k5 * r2;
      const float mz = (1.0f - k5 * k4 * k4 * r2) / (k4 * sqrtf(fmaxf(sv, 1e-5f)) + (1.0f - k4));
      const float cf = 1.0f / sqrtf(mx * mx + my * my + mz * mz + 1e-5f);
      x = cf * mx; y = cf * my; z = fmaxf(cf * mz, 1e-3f);
      nmin = 1e-4f;
    } else if (gt_mode == 3) {
      // Spherical / equirectangular (utils/camera.py:371-386): k4, k5 = image width, height; k6, k7 = half fields of view (rad)
      const float lon = (uf - 0.5f * (k4 - 1.0f)) / (k4 - 1.0f) * (2.0f * k6);
      const float lat = (vf - 0.5f * (k5 - 1.0f)) / (k5 - 1.0f) * (2.0f * k7);
      const float cl = cosf(lat);
      x = cl * sinf(lon); y = sinf(lat); z = cl * cosf(lon);
      const float n1 = 1.0f / fmaxf(sqrtf(x * x + y * y + z * z), 1e-5f);
      x *= n1; y *= n1; z *= n1;
      nmin = 1e-4f;
    }
    const float nrm = sqrtf(x * x + y * y + z * z);
    const float inv = 1.0f / (nrm < nmin ? nmin : nrm);        // == fmaxf(nrm, nmin) for numbers; a NaN camera (the camera head's loud failure,
    float* r = rays + (size_t)b * 3 * HW + pix;                // UdCameraHead.sync_ws) stays NaN in all three components instead of being clamped away
    r[0] = x * inv; r[HW] = y * inv; r[2 * (size_t)HW] = z * inv;
  }
}

// ------------------------------------------------------------------------------------------------ iterative GT cameras
// OPENCV / Fisheye624 (utils/camera.py:496-694 / :778-974): the radial solver is a trust-region Newton loop that the reference
// leaves for ALL pixels at once, as soon as the largest |residual| of the image drops below 1e-3.  To reproduce that the loop
// is unrolled over launches: init (tangential / thin-prism Newton, theta_0, trust radius) -> up to 10 step kernels -> final.
// Step kernel `it` first looks at maxres[it], the largest residual at theta_it written by its predecessor with one atomicMax per
// wave (non-negative floats order like their bit patterns), and returns at once when the reference would have left the loop;
// maxres[] is cleared per call, so every later step kernel returns too.  Per-pixel state: float4 {xr, yr, theta, trust radius}.
__device__ __forceinline__ bool cam_use_radial(const float* p) {
  return fabsf(p[4]) + fabsf(p[5]) + fabsf(p[6]) + fabsf(p[7]) + fabsf(p[8]) + fabsf(p[9]) > 1e-6f;
}

__device__ __forceinline__ void cam_publish_max(float r, unsigned* slot) {
  // NaN compares above every finite value as a bit pattern as well: the loop then runs on, as `nan < eps` is false in the reference
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float other = __shfl_xor(r, o, 64);
    r = (other > r || other != other) ? other : r;
  }
  if ((threadIdx.x & 63) == 0) atomicMax(slot, __float_as_uint(r));
}

__global__ __launch_bounds__(256) void rays_iter_init_kernel(const float* params, float4* st, unsigned* maxres, int Hn, int Wn, int nk) {
  const float* p = params;
  const bool use_tan = fabsf(p[10]) + fabsf(p[11]) > 1e-6f;
  const bool use_prism = fabsf(p[12]) + fabsf(p[13]) + fabsf(p[14]) + fabsf(p[15]) > 1e-6f;
  const bool use_radial = cam_use_radial(p);
  const int HW = Hn * Wn;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  float r = 0.0f;
  if (pix < HW) {
    const int v = pix / Wn, u = pix - v * Wn;
    const float ud = ((float)u + 0.5f - p[2]) / p[0], vd = ((float)v + 0.5f - p[3]) / p[1];
    float xr, yr;
    ud_cam_undistort_tanprism(ud, vd, p[10], p[11], p[12], p[13], p[14], p[15], use_tan, use_prism, (use_tan || use_prism) ? 10 : 0, xr, yr);
    const float rnorm = sqrtf(xr * xr + yr * yr);
    st[pix] = make_float4(xr, yr, rnorm, 0.1f);
    if (use_radial) r = fabsf(ud_cam_radial_residual(p + 4, nk, rnorm, rnorm));
  }
  if (use_radial) cam_publish_max(r, maxres);
}

__global__ __launch_bounds__(256) void rays_iter_step_kernel(const float* params, float4* st, unsigned* maxres, int HW, int it, int nk) {
  if (!cam_use_radial(params)) return;
  if (__uint_as_float(maxres[it]) < UD_CAM_EPS) return;            // the reference left the loop at (or before) this iteration
  const int pix = blockIdx.x * 256 + threadIdx.x;
  float r = 0.0f;
  if (pix < HW) {
    float4 s = st[pix];
    const float rnorm = sqrtf(s.x * s.x + s.y * s.y);
    ud_cam_radial_step(params + 4, nk, rnorm, s.z, s.w);
    st[pix] = s;
    r = fabsf(ud_cam_radial_residual(params + 4, nk, s.z, rnorm));
  }
  cam_publish_max(r, maxres + it + 1);
}

__global__ __launch_bounds__(256) void rays_iter_final_kernel(const float4* st, float* rays, int HW, int tan_theta) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= HW) return;
  const float4 s = st[pix];
  float x, y, z;
  ud_cam_finish_radial(s.x, s.y, sqrtf(s.x * s.x + s.y * s.y), s.z, tan_theta, x, y, z);
  rays[pix] = x; rays[HW + pix] = y; rays[2 * (size_t)HW + pix] = z;
}

__global__ __launch_bounds__(256) void rays_mei_kernel(const float* params, float* rays, int Hn, int Wn) {
  const int HW = Hn * Wn;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= HW) return;
  const int v = pix / Wn, u = pix - v * Wn;
  float x, y, z;
  ud_cam_mei(params, (float)u + 0.5f, (float)v + 0.5f, x, y, z);
  rays[pix] = x; rays[HW + pix] = y; rays[2 * (size_t)HW + pix] = z;
}

// ------------------------------------------------------------------------------------------------ ray embedding
// one wave per output token: lanes split the (<= 4*scale^2) taps of the separable antialias triangle filter,
// wave reduction, then each lane evaluates C/64 sine bands; LayerNorm statistics over C via wave reductions.
__global__ __launch_bounds__(256) void ray_embed_kernel(const UdRayEmbed p) {
  const int lane = threadIdx.x & 63;
  const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int hw = p.h * p.w;
  if (tok >= p.nb * hw) return;
  const int img = tok / hw;
  const int t = tok - img * hw;
  const int ty = t / p.w, tx = t - ty * p.w;
  const float sy = (float)p.Hn / (float)p.h, sx = (float)p.Wn / (float)p.w;
  const float supy = sy >= 1.0f ? sy : 1.0f, supx = sx >= 1.0f ? sx : 1.0f;
  const float invy = sy >= 1.0f ? 1.0f / sy : 1.0f, invx = sx >= 1.0f ? 1.0f / sx : 1.0f;
  const float cy = sy * ((float)ty + 0.5f), cx = sx * ((float)tx + 0.5f);
  int ymin = (int)(cy - supy + 0.5f); ymin = ymin < 0 ? 0 : ymin;
  int ymax = (int)(cy + supy + 0.5f); ymax = ymax > p.Hn ? p.Hn : ymax;
  int xmin = (int)(cx - supx + 0.5f); xmin = xmin < 0 ? 0 : xmin;
  int xmax = (int)(cx + supx + 0.5f); xmax = xmax > p.Wn ? p.Wn : xmax;
  const int ny = ymax - ymin, nx = xmax - xmin;
  const size_t HW = (size_t)p.Hn * p.Wn;
  const float* r = p.rays + (size_t)img * 3 * HW;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, wsum = 0.f;
  for (int i = lane; i < ny * nx; i += 64) {
    const int iy = i / nx, ix = i - iy * nx;
    float wy = 1.0f - fabsf(((float)(iy + ymin) - cy + 0.5f) * invy);
    float wx = 1.0f - fabsf(((float)(ix + xmin) - cx + 0.5f) * invx);
    wy = wy < 0.f ? 0.f : wy;
    wx = wx < 0.f ? 0.f : wx;
    const float w = wy * wx;
    const size_t off = (size_t)(iy + ymin) * p.Wn + (ix + xmin);
    a0 += w * r[off]; a1 += w * r[off + HW]; a2 += w * r[off + 2 * HW];
    wsum += w;
  }
  a0 = ud_wave_sum(a0); a1 = ud_wave_sum(a1); a2 = ud_wave_sum(a2); wsum = ud_wave_sum(wsum);
  const float iw = 1.0f / wsum;
  float x = a0 * iw, y = a1 * iw, z = a2 * iw;
  const float inv = 1.0f / fmaxf(sqrtf(x * x + y * y + z * z), 1e-4f);
  x *= inv; y *= inv; z *= inv;
  const float polar = acosf(z);
  const float xc = fmaxf(fabsf(x), 1e-3f) * (x >= 0.0f ? 1.0f : -1.0f);
  const float az = atan2f(y, xc);
  const int nbands = p.C >> 1;
  const float PI = 3.14159265358979323846f;
  // C <= 512 -> up to 8 values per lane
  float vals[8];
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int c = it * 64 + lane;
    float v = 0.f;
    if (c < p.C) {
      const bool second = c >= nbands;
      const float ang = second ? az : polar;
      const float sc = p.scales[second ? c - nbands : c];
      v = sinf(ang * sc * PI);
      s += v;
    }
    vals[it] = v;
  }
  const float mean = ud_wave_sum(s) / (float)p.C;
  float q = 0.f;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int c = it * 64 + lane;
    if (c < p.C) {
      const float d = vals[it] - mean;
      q += d * d;
    }
  }
  const float rstd = rsqrtf(ud_wave_sum(q) / (float)p.C + p.eps);
  half_t* yrow = (half_t*)p.xhat + ((size_t)img * p.rows_per_img + t) * p.ldy;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int c = it * 64 + lane;
    if (c < p.C) yrow[c] = (half_t)((vals[it] - mean) * rstd);
  }
}

// ------------------------------------------------------------------------------------------------ x2 up-sampling
// thread = 4 channels of one output pixel (16-byte fp32 loads); mode 1 additionally normalises the pixel over C.
template <int MODE>
__global__ __launch_bounds__(256) void upsample2x_kernel(const UdUpsample2x p) {
  const int Ho = p.H * 2, Wo = p.W * 2;
  const int G = p.C >> 2;                     // threads per pixel
  const int ppb = 256 / G;                    // pixels per block (G divides into 256 with remainder ignored)
  const int tl = threadIdx.x;
  const int pl = tl / G, cg = tl - pl * G;
  __shared__ float red[2][256];
  // grid: x = tiles of ppb output pixels along a row, y = output row (image * Ho + oy): no per-thread divisions (the first
  // version decoded a flat 64-bit pixel index with three 64-bit div/mod per thread and ran at 1.9 TB/s)
  for (int row = blockIdx.y; row < p.B * Ho; row += gridDim.y) {
  const int b = row / Ho, oy = row - b * Ho;
  for (int base = blockIdx.x * ppb; base < Wo; base += gridDim.x * ppb) {
    const int ox = base + pl;
    const bool active = pl < ppb && ox < Wo;
    const long long pix = (long long)row * Wo + ox;
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (active) {
      float fy = 0.5f * ((float)oy + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
      float fx = 0.5f * ((float)ox + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
      const int y0 = (int)fy, x0 = (int)fx;
      const int y1 = y0 + (y0 < p.H - 1), x1 = x0 + (x0 < p.W - 1);
      const float ly = fy - (float)y0, lx = fx - (float)x0;
      const float* in = (const float*)p.in + (size_t)b * (p.in_img_rows > 0 ? p.in_img_rows : p.H * p.W) * p.ldin + cg * 4;
      const f32x4 v00 = *(const f32x4*)(in + ((size_t)y0 * p.W + x0) * p.ldin);
      const f32x4 v01 = *(const f32x4*)(in + ((size_t)y0 * p.W + x1) * p.ldin);
      const f32x4 v10 = *(const f32x4*)(in + ((size_t)y1 * p.W + x0) * p.ldin);
      const f32x4 v11 = *(const f32x4*)(in + ((size_t)y1 * p.W + x1) * p.ldin);
      v = (1.0f - ly) * ((1.0f - lx) * v00 + lx * v01) + ly * ((1.0f - lx) * v10 + lx * v11);
    }
    if constexpr (MODE == 0) {
      if (active) *(f32x4*)((float*)p.out + (size_t)pix * p.ldy + cg * 4) = v;
    } else {
      // LayerNorm statistics across the G threads of the pixel.  G a power of two <= 64 (C = 64 / 128 / 256): the pixel's threads are
      // an aligned lane group of one wave -> xor-shuffle butterflies, no LDS, no barriers (the LDS version below ran at 1.85 TB/s).
      if ((G & (G - 1)) == 0 && G <= 64) {
        float s1 = (v[0] + v[1]) + (v[2] + v[3]);
        for (int o = G >> 1; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
        const float mean = s1 / (float)p.C;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) q += (v[e] - mean) * (v[e] - mean);
        for (int o = G >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        if (active) {
          const float rstd = rsqrtf(q / (float)p.C + p.eps);
          half4 h;
#pragma unroll
          for (int e = 0; e < 4; ++e) h[e] = (half_t)((v[e] - mean) * rstd);
          *(half4*)((half_t*)p.out + (size_t)pix * p.ldy + cg * 4) = h;
        }
        continue;
      }
      red[0][tl] = (v[0] + v[1]) + (v[2] + v[3]);
      __syncthreads();
      float mean = 0.f;
      if (active) {
        for (int g = 0; g < G; ++g) mean += red[0][pl * G + g];
        mean /= (float)p.C;
      }
      float q = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) q += (v[e] - mean) * (v[e] - mean);
      red[1][tl] = q;
      __syncthreads();
      if (active) {
        float var = 0.f;
        for (int g = 0; g < G; ++g) var += red[1][pl * G + g];
        const float rstd = rsqrtf(var / (float)p.C + p.eps);
        half4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (half_t)((v[e] - mean) * rstd);
        *(half4*)((half_t*)p.out + (size_t)pix * p.ldy + cg * 4) = h;
      }
      __syncthreads();
    }
  }
  }
}

// mode 1 for C = 8 * 2^k (the x8 map of every backbone: C/4 = 128 / 96 -> no / 64): thread = 8 channels of one output pixel, the
// pixel's C/8 threads are an aligned lane group of one wave (xor-shuffle LayerNorm statistics), 16-byte fp16 stores
// (the 4-channel version stored 8 bytes per lane and ran at 2.0 TB/s).
__global__ __launch_bounds__(256) void upsample2x_ln8_kernel(const UdUpsample2x p) {
  // Round 6: a thread produces the TWO output rows that interpolate between the same pair of source rows (oy = 2 j - 1 and 2 j: source rows j - 1 and j, weights
  // 0.25 / 0.75 and 0.75 / 0.25; the first and the last output row stand alone) from ONE set of four source-pixel loads: half the L1 / L2 read traffic of one
  // output pixel per thread (1.4 GB per launch at bs = 8, the kernel's bound).  Per output the same expression with the same weights as before: same bits.
  const int Ho = p.H * 2, Wo = p.W * 2;
  const int G = p.C >> 3;
  const int ppb = 256 / G;
  const int tl = threadIdx.x;
  const int pl = tl / G, cg = tl - pl * G;
  const int pairs = p.H + 1;                     // row pairs per image: j = 0 .. H
  for (int row = blockIdx.y; row < p.B * pairs; row += gridDim.y) {
    const int b = row / pairs, j = row - b * pairs;
    const int ya = j > 0 ? j - 1 : 0, yb = j < p.H ? j : p.H - 1;
    const int oyA = 2 * j - 1, oyB = 2 * j;      // oyA valid for j >= 1, oyB valid for j < H
    // the weights the one-output kernel computed for these rows: fy = 0.5 (oy + 0.5) - 0.5 clamped at 0, ly = fy - floor(fy)
    const float lyA = 0.25f;
    const float lyB = j == 0 ? 0.0f : 0.75f;
    const float* img = (const float*)p.in + (size_t)b * (p.in_img_rows > 0 ? p.in_img_rows : p.H * p.W) * p.ldin + cg * 8;
    for (int base = blockIdx.x * ppb; base < Wo; base += gridDim.x * ppb) {
      const int ox = base + pl;
      const bool active = ox < Wo;
      f32x4 vA[2], vB[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) { vA[h] = (f32x4){0.f, 0.f, 0.f, 0.f}; vB[h] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      if (active) {
        float fx = 0.5f * ((float)ox + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
        const int x0 = (int)fx;
        const int x1 = x0 + (x0 < p.W - 1);
        const float lx = fx - (float)x0;
        const float* r00 = img + ((size_t)ya * p.W + x0) * p.ldin;
        const float* r01 = img + ((size_t)ya * p.W + x1) * p.ldin;
        const float* r10 = img + ((size_t)yb * p.W + x0) * p.ldin;
        const float* r11 = img + ((size_t)yb * p.W + x1) * p.ldin;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 v00 = *(const f32x4*)(r00 + 4 * h), v01 = *(const f32x4*)(r01 + 4 * h);
          const f32x4 v10 = *(const f32x4*)(r10 + 4 * h), v11 = *(const f32x4*)(r11 + 4 * h);
          const f32x4 top = (1.0f - lx) * v00 + lx * v01, bot = (1.0f - lx) * v10 + lx * v11;
          vA[h] = (1.0f - lyA) * top + lyA * bot;
          vB[h] = (1.0f - lyB) * top + lyB * bot;
        }
      }
      auto finish = [&](const f32x4 (&v)[2], int oy, bool valid) {
        float s1 = ((v[0][0] + v[0][1]) + (v[0][2] + v[0][3])) + ((v[1][0] + v[1][1]) + (v[1][2] + v[1][3]));
        for (int o = G >> 1; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
        const float mean = s1 / (float)p.C;
        float q = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e) q += (v[h][e] - mean) * (v[h][e] - mean);
        for (int o = G >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        if (active && valid) {
          const float rstd = rsqrtf(q / (float)p.C + p.eps);
          half8 hv;
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) hv[4 * h + e] = (half_t)((v[h][e] - mean) * rstd);
          *(half8*)((half_t*)p.out + (((size_t)b * Ho + oy) * Wo + ox) * p.ldy + cg * 8) = hv;
        }
      };
      finish(vA, oyA, j >= 1);
      finish(vB, oyB, j < p.H);
    }
  }
}

// ------------------------------------------------------------------------------------------------ align_corners=True resize
__global__ __launch_bounds__(256) void resize_ac_kernel(const UdResizeAC p) {
  const int CG = p.C >> 3;
  const float sy = p.Hout > 1 ? (float)(p.Hin - 1) / (float)(p.Hout - 1) : 0.f;
  const float sx = p.Wout > 1 ? (float)(p.Win - 1) / (float)(p.Wout - 1) : 0.f;
  // grid: x = chunks of 256 (pixel, channel-group) items along an output row, y = output row (gb * Hout + oy)
  const int row_items = p.Wout * CG;
  for (int row = blockIdx.y; row < p.G * p.B * p.Hout; row += gridDim.y) {
  const int gbi = row / p.Hout, oy = row - gbi * p.Hout;
  const long long gb = gbi;
  for (int it = blockIdx.x * 256 + threadIdx.x; it < row_items; it += gridDim.x * 256) {
    const int ox = it / CG, cg = it - ox * CG;
    const long long pix = (long long)row * p.Wout + ox;
    const float fy = sy * (float)oy, fx = sx * (float)ox;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < p.Hin - 1), x1 = x0 + (x0 < p.Win - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const half_t* in = (const half_t*)p.in + (size_t)gb * p.Hin * p.Win * p.C + cg * 8;
    const half8 h00 = *(const half8*)(in + ((size_t)y0 * p.Win + x0) * p.C);
    const half8 h01 = *(const half8*)(in + ((size_t)y0 * p.Win + x1) * p.C);
    const half8 h10 = *(const half8*)(in + ((size_t)y1 * p.Win + x0) * p.C);
    const half8 h11 = *(const half8*)(in + ((size_t)y1 * p.Win + x1) * p.C);
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float top = (1.0f - lx) * (float)h00[e] + lx * (float)h01[e];
      const float bot = (1.0f - lx) * (float)h10[e] + lx * (float)h11[e];
      o[e] = (half_t)((1.0f - ly) * top + ly * bot);
    }
    *(half8*)((half_t*)p.out + (size_t)pix * p.C + cg * 8) = o;
  }
  }
}

// ------------------------------------------------------------------------------------------------ output assembly
// MODE 0: bilinear, MODE 1: bicubic (both align_corners=False, as F.interpolate in unidepthv2.py:80-89 computes them: bicubic with
// A = -0.75, un-clamped source coordinate, border-clamped taps).  points = resample(rays_net * radius_net), rays = resample(rays_net)
// renormalised, radius = |points|, depth = points_z (unidepthv2.py:318-338).
__device__ __forceinline__ void ud_cubic_w(float t, float (&w)[4]) {
  const float A = -0.75f;
  const float x0 = t + 1.0f, x2 = 1.0f - t, x3 = 2.0f - t;
  w[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
  w[1] = ((A + 2.0f) * t - (A + 3.0f)) * t * t + 1.0f;
  w[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
  w[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}

template <int MODE>
__global__ __launch_bounds__(256) void finalize_kernel(const UdFinalize p) {
  const long long total = (long long)p.B * p.Ho * p.Wo;
  const float sy = (float)p.Hn / (float)p.Hp, sx = (float)p.Wn / (float)p.Wp;
  const size_t HWn = (size_t)p.Hn * p.Wn, HWo = (size_t)p.Ho * p.Wo;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int ox = (int)(idx % p.Wo);
    const int oy = (int)((idx / p.Wo) % p.Ho);
    const int b = (int)(idx / ((long long)p.Wo * p.Ho));
    const float* rad = p.radius_net + (size_t)b * HWn;
    const float* cf = p.conf_net + (size_t)b * HWn;
    const float* ry = p.rays_net + (size_t)(p.nb_rays == 1 ? 0 : b) * 3 * HWn;
    float conf = 0.f, pt[3] = {0.f, 0.f, 0.f}, rr[3] = {0.f, 0.f, 0.f};
    float fy = sy * ((float)(oy + p.pad_t) + 0.5f) - 0.5f;
    float fx = sx * ((float)(ox + p.pad_l) + 0.5f) - 0.5f;
    if constexpr (MODE == 0) {
      fy = fy < 0.f ? 0.f : fy;
      fx = fx < 0.f ? 0.f : fx;
      const int y0 = (int)fy, x0 = (int)fx;
      const int y1 = y0 + (y0 < p.Hn - 1), x1 = x0 + (x0 < p.Wn - 1);
      const float ly = fy - (float)y0, lx = fx - (float)x0;
      const float w00 = (1.0f - ly) * (1.0f - lx), w01 = (1.0f - ly) * lx, w10 = ly * (1.0f - lx), w11 = ly * lx;
      const size_t o00 = (size_t)y0 * p.Wn + x0, o01 = (size_t)y0 * p.Wn + x1, o10 = (size_t)y1 * p.Wn + x0, o11 = (size_t)y1 * p.Wn + x1;
      const float r00 = rad[o00], r01 = rad[o01], r10 = rad[o10], r11 = rad[o11];
      conf = w00 * cf[o00] + w01 * cf[o01] + w10 * cf[o10] + w11 * cf[o11];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float* rc = ry + ch * HWn;
        const float a = rc[o00], bq = rc[o01], cq = rc[o10], d = rc[o11];
        rr[ch] = w00 * a + w01 * bq + w10 * cq + w11 * d;
        pt[ch] = w00 * (a * r00) + w01 * (bq * r01) + w10 * (cq * r10) + w11 * (d * r11);
      }
    } else {
      const float ffy = floorf(fy), ffx = floorf(fx);
      float wy[4], wx[4];
      ud_cubic_w(fy - ffy, wy);
      ud_cubic_w(fx - ffx, wx);
      const int iy = (int)ffy, ix = (int)ffx;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int yy = iy - 1 + i;
        yy = yy < 0 ? 0 : (yy > p.Hn - 1 ? p.Hn - 1 : yy);
        float cacc = 0.f, pacc[3] = {0.f, 0.f, 0.f}, racc[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int xx = ix - 1 + j;
          xx = xx < 0 ? 0 : (xx > p.Wn - 1 ? p.Wn - 1 : xx);
          const size_t o = (size_t)yy * p.Wn + xx;
          const float r = rad[o];
          cacc += wx[j] * cf[o];
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) {
            const float a = ry[ch * HWn + o];
            racc[ch] += wx[j] * a;
            pacc[ch] += wx[j] * (a * r);
          }
        }
        conf += wy[i] * cacc;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) { rr[ch] += wy[i] * racc[ch]; pt[ch] += wy[i] * pacc[ch]; }
      }
    }
    const size_t po = (size_t)oy * p.Wo + ox;
    p.confidence[(size_t)b * HWo + po] = conf;
    p.radius[(size_t)b * HWo + po] = sqrtf(pt[0] * pt[0] + pt[1] * pt[1] + pt[2] * pt[2]);
    p.depth[(size_t)b * HWo + po] = pt[2];
    const float rn = 1.0f / fmaxf(sqrtf(rr[0] * rr[0] + rr[1] * rr[1] + rr[2] * rr[2]), 1e-5f);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      p.points[((size_t)b * 3 + ch) * HWo + po] = pt[ch];
      if (p.nb_rays != 1 || b == 0) p.rays[((size_t)b * 3 + ch) * HWo + po] = rr[ch] * rn;
    }
  }
}

// ------------------------------------------------------------------------------------------------ NHWC -> NCHW
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* in, float* out, int B, int hw, int C, int ld, int rows_per_img) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z;
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int pp = p0 + r, c = c0 + tx;
    tile[r][tx] = (pp < hw && c < C) ? in[((size_t)b * rows_per_img + pp) * ld + c] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int c = c0 + r, pp = p0 + tx;
    if (pp < hw && c < C) out[((size_t)b * C + c) * hw + pp] = tile[tx][r];
  }
}

inline int grid_for(long long total, int per_block = 256, int cap = 256 * 16) {
  long long g = (total + per_block - 1) / per_block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int ud_preprocess_patches(const UdPreprocess* desc, void* stream) {
  const UdPreprocess& d = *desc;
  if (!d.rgb || !d.patches || d.B <= 0 || d.Hn % 14 || d.Wn % 14 || d.ldp < 588 || d.Hp < d.H + d.pad_t || d.Wp < d.W + d.pad_l) {
    ud_set_error("ud_preprocess_patches: bad argument");
    return UD_ERR_BAD_ARG;
  }
  const long long total = (long long)d.B * (d.Hn / 14) * (d.Wn / 14) * 588;
  hipLaunchKernelGGL(preprocess_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, d);
  UD_CHECK_LAUNCH("ud_preprocess_patches launch");
  return UD_OK;
}

extern "C" int ud_fill_rows_f32(float* dst, const float* src, int n_img, int rows_per_img, int row_off, int D, int ld, void* stream) {
  if (!dst || !src || n_img <= 0 || D <= 0) { ud_set_error("ud_fill_rows_f32: bad argument"); return UD_ERR_BAD_ARG; }
  hipLaunchKernelGGL(fill_rows_kernel, dim3((D + 255) / 256, n_img), dim3(256), 0, (hipStream_t)stream, dst, src, n_img, rows_per_img, row_off, D, ld);
  UD_CHECK_LAUNCH("ud_fill_rows_f32 launch");
  return UD_OK;
}

extern "C" int ud_camera_intrinsics(const float* raw, int raw_stride, float* intr4, float* K33, float* Kinv33, float* Kpost33, int B,
                                    int Hn, int Wn, float resize_factor, int pad_l, int pad_t, void* stream) {
  if (!raw || !intr4 || !K33 || !Kinv33 || !Kpost33 || B <= 0 || raw_stride <= 0 || !(resize_factor > 0.f)) {
    ud_set_error("ud_camera_intrinsics: bad argument");
    return UD_ERR_BAD_ARG;
  }
  hipLaunchKernelGGL(camera_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, raw, raw_stride, intr4, K33, Kinv33, Kpost33, B, Hn,
                     Wn, resize_factor, pad_l, pad_t);
  UD_CHECK_LAUNCH("ud_camera_intrinsics launch");
  return UD_OK;
}

extern "C" int ud_rays_from_kinv(const float* Kinv33, float* rays, int nb, int Hn, int Wn, int gt_mode, void* stream) {
  if (!Kinv33 || !rays || nb <= 0 || Hn <= 0 || Wn <= 0) { ud_set_error("ud_rays_from_kinv: bad argument"); return UD_ERR_BAD_ARG; }
  hipLaunchKernelGGL(rays_kernel, dim3(grid_for((long long)Hn * Wn, 256, 1024), nb), dim3(256), 0, (hipStream_t)stream, Kinv33, rays, nb, Hn, Wn, gt_mode);
  UD_CHECK_LAUNCH("ud_rays_from_kinv launch");
  return UD_OK;
}

extern "C" int ud_rays_from_camera(const float* params, float* rays, float* scratch, int Hn, int Wn, int model, void* stream) {
  if (!params || !rays || Hn <= 0 || Wn <= 0 || model < 4 || model > 6 || (model != 6 && !scratch)) {
    ud_set_error("ud_rays_from_camera: bad argument (model 4 OPENCV, 5 Fisheye624, 6 MEI; scratch needed for 4 and 5)");
    return UD_ERR_BAD_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  const int HW = Hn * Wn;
  const dim3 grid((HW + 255) / 256);
  if (model == 6) {
    hipLaunchKernelGGL(rays_mei_kernel, grid, dim3(256), 0, s, params, rays, Hn, Wn);
    UD_CHECK_LAUNCH("ud_rays_from_camera (MEI) launch");
    return UD_OK;
  }
  float4* st = (float4*)scratch;
  unsigned* maxres = (unsigned*)(scratch + 4 * (size_t)HW);
  const int nk = model == 5 ? 6 : 3;
  if (hipMemsetAsync(maxres, 0, 16 * sizeof(unsigned), s) != hipSuccess) { ud_set_error("ud_rays_from_camera: memset failed"); return UD_ERR_LAUNCH; }
  hipLaunchKernelGGL(rays_iter_init_kernel, grid, dim3(256), 0, s, params, st, maxres, Hn, Wn, nk);
  for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(rays_iter_step_kernel, grid, dim3(256), 0, s, params, st, maxres, HW, it, nk);
  hipLaunchKernelGGL(rays_iter_final_kernel, grid, dim3(256), 0, s, (const float4*)st, rays, HW, model == 5 ? 1 : 0);
  UD_CHECK_LAUNCH("ud_rays_from_camera launch");
  return UD_OK;
}

extern "C" int ud_ray_embed(const UdRayEmbed* desc, void* stream) {
  const UdRayEmbed& d = *desc;
  if (!d.rays || !d.scales || !d.xhat || d.nb <= 0 || d.C > 512 || (d.C & 1) || d.h <= 0 || d.w <= 0) {
    ud_set_error("ud_ray_embed: bad argument (C <= 512)");
    return UD_ERR_BAD_ARG;
  }
  const int ntok = d.nb * d.h * d.w;
  hipLaunchKernelGGL(ray_embed_kernel, dim3((ntok + 3) / 4), dim3(256), 0, (hipStream_t)stream, d);
  UD_CHECK_LAUNCH("ud_ray_embed launch");
  return UD_OK;
}

extern "C" int ud_upsample2x_nhwc(const UdUpsample2x* desc, void* stream) {
  const UdUpsample2x& d = *desc;
  if (!d.in || !d.out || d.B <= 0 || (d.C & 3) || d.C > 1024 || (d.ldin & 3) || (d.ldy & 3)) {
    ud_set_error("ud_upsample2x_nhwc: bad argument");
    return UD_ERR_BAD_ARG;
  }
  const int G = d.C >> 2;
  const int ppb = 256 / G;
  const int rows = d.B * 2 * d.H;
  const dim3 grid((2 * d.W + ppb - 1) / ppb, rows < 65535 ? rows : 65535);
  const int G8 = d.C >> 3;
  if (d.mode == 1 && (d.C & 7) == 0 && (G8 & (G8 - 1)) == 0 && G8 <= 64 && (d.ldy & 7) == 0) {
    const int prow = d.B * (d.H + 1);                            // row PAIRS (upsample2x_ln8_kernel)
    const dim3 grid8((2 * d.W + 256 / G8 - 1) / (256 / G8), prow < 65535 ? prow : 65535);
    hipLaunchKernelGGL(upsample2x_ln8_kernel, grid8, dim3(256), 0, (hipStream_t)stream, d);
  } else if (d.mode == 0) hipLaunchKernelGGL(upsample2x_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, d);
  else hipLaunchKernelGGL(upsample2x_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, d);
  UD_CHECK_LAUNCH("ud_upsample2x_nhwc launch");
  return UD_OK;
}

extern "C" int ud_resize_ac_nhwc_f16(const UdResizeAC* desc, void* stream) {
  const UdResizeAC& d = *desc;
  if (!d.in || !d.out || d.G <= 0 || d.B <= 0 || (d.C & 7)) { ud_set_error("ud_resize_ac_nhwc_f16: bad argument"); return UD_ERR_BAD_ARG; }
  const int rows = d.G * d.B * d.Hout;
  const dim3 grid((d.Wout * (d.C >> 3) + 255) / 256, rows < 65535 ? rows : 65535);
  hipLaunchKernelGGL(resize_ac_kernel, grid, dim3(256), 0, (hipStream_t)stream, d);
  UD_CHECK_LAUNCH("ud_resize_ac_nhwc_f16 launch");
  return UD_OK;
}

extern "C" int ud_finalize_outputs(const UdFinalize* desc, void* stream) {
  const UdFinalize& d = *desc;
  if (!d.radius_net || !d.conf_net || !d.rays_net || !d.confidence || !d.radius || !d.depth || !d.points || !d.rays || d.B <= 0 ||
      d.Ho + d.pad_t > d.Hp || d.Wo + d.pad_l > d.Wp || d.mode < 0 || d.mode > 1) {
    ud_set_error("ud_finalize_outputs: bad argument (mode: 0 bilinear, 1 bicubic)");
    return UD_ERR_BAD_ARG;
  }
  const dim3 grid(grid_for((long long)d.B * d.Ho * d.Wo, 256, 256 * 32));
  if (d.mode == 1) hipLaunchKernelGGL(finalize_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, d);
  else hipLaunchKernelGGL(finalize_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, d);
  UD_CHECK_LAUNCH("ud_finalize_outputs launch");
  return UD_OK;
}

extern "C" int ud_nhwc_to_nchw_f32(const float* in, float* out, int B, int hw, int C, int ld, int rows_per_img, void* stream) {
  if (!in || !out || B <= 0 || hw <= 0 || C <= 0) { ud_set_error("ud_nhwc_to_nchw_f32: bad argument"); return UD_ERR_BAD_ARG; }
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((hw + 63) / 64, (C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, in, out, B, hw, C, ld, rows_per_img);
  UD_CHECK_LAUNCH("ud_nhwc_to_nchw_f32 launch");
  return UD_OK;
}
