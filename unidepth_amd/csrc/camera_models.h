// Per-pixel arithmetic of the iterative ground-truth camera models (reference unidepth/utils/camera.py: OPENCV :412-694,
// Fisheye624 :697-974, MEI :977-1082).  Plain float functions with no device intrinsics: pointwise.hip compiles them into the
// ray kernels, tests/test_camera_models_cpu.py compiles the same header for the host to check the arithmetic without a GPU.
//
// Parameter vectors (after the crop/resize bookkeeping of infer(), one camera):
//   OPENCV / Fisheye624 : [fx, fy, cx, cy, k1..k6, p1, p2, s1..s4]  (16 floats; OPENCV uses k1..k3, k4..k6 must be 0)
//   MEI                 : [fx, fy, cx, cy, k1, k2, p1, p2, xi]
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define UD_CAM_FN __host__ __device__ __forceinline__
#else
#define UD_CAM_FN static inline
#endif

#define UD_CAM_EPS 1e-3f          /* convergence / guard threshold of the OPENCV and Fisheye624 solvers (camera.py:497,779) */
#define UD_CAM_EPS_MEI 1e-6f      /* camera.py:986 */

// Newton iterations undoing tangential (p0, p1) and thin-prism (s0..s3) distortion: solves dist(xr, yr) = (ud, vd) starting
// from (ud, vd), a fixed number of iterations (camera.py:517-585 / :799-867 / :1000-1046).  The reference builds the Jacobian
// from a matrix of ONES (`new_ones(B, N, 2, 2)`), so with thin-prism terms but no tangential terms its off-diagonals start
// at 1, not 0 -- reproduced here, since the results are what has to match.
UD_CAM_FN void ud_cam_undistort_tanprism(float ud, float vd, float p0, float p1, float s0, float s1, float s2, float s3,
                                         int use_tan, int use_prism, int iters, float& xr_out, float& yr_out) {
  float xr = ud, yr = vd;
  for (int it = 0; it < iters; ++it) {
    const float xr2 = xr * xr, yr2 = yr * yr;
    const float rd = xr2 + yr2;
    float ex = xr, ey = yr;
    float a = 1.0f, b = 1.0f, c = 1.0f, d = 1.0f;
    if (use_tan) {
      ex = ex + ((2.0f * xr2 + rd) * p0 + 2.0f * xr * yr * p1);
      ey = ey + ((2.0f * yr2 + rd) * p1 + 2.0f * xr * yr * p0);
      a = 1.0f + 6.0f * xr * p0 + 2.0f * yr * p1;
      b = 2.0f * (xr * p1 + yr * p0);
      c = b;
      d = 1.0f + 6.0f * yr * p1 + 2.0f * xr * p0;
    }
    if (use_prism) {
      const float rd4 = rd * rd;
      ex = ex + (s0 * rd + s1 * rd4);
      ey = ey + (s2 * rd + s3 * rd4);
      const float t1 = 2.0f * (s0 + 2.0f * s1 * rd);
      const float t2 = 2.0f * (s2 + 2.0f * s3 * rd);
      a += xr * t1; b += yr * t1;
      c += xr * t2; d += yr * t2;
    }
    const float det = 1.0f / (a * d - b * c);
    const float e = ud - ex, f = vd - ey;
    xr = xr + ((det * d) * e + (det * -b) * f);
    yr = yr + ((det * -c) * e + (det * a) * f);
  }
  xr_out = xr;
  yr_out = yr;
}

// Radial polynomial th * (1 + sum_j k_j th^(2j+2)) and its derivative 1 + sum_j (2j+3) k_j th^(2j+2), j < nk
// (camera.py:603-622: nk = 3 for OPENCV, 6 for Fisheye624).
UD_CAM_FN float ud_cam_radial(const float* k, int nk, float th, float& dth) {
  const float t2 = th * th;
  float pw = t2, s = 0.0f, ds = 0.0f;
  for (int j = 0; j < nk; ++j) {
    s += pw * k[j];
    ds += (2.0f * (float)j + 3.0f) * k[j] * pw;
    pw *= t2;
  }
  dth = 1.0f + ds;
  return (1.0f + s) * th;
}

UD_CAM_FN float ud_cam_radial_residual(const float* k, int nk, float th, float rnorm) {
  float dth;
  return ud_cam_radial(k, nk, th, dth) - rnorm;
}

// One trust-region Newton iteration on theta (camera.py:603-691 / :885-971).  `delta` is the pixel's trust radius.
// The caller decides whether the iteration runs at all: the reference stops ALL pixels as soon as the largest |residual|
// over the image is below UD_CAM_EPS (`if torch.max(torch.abs(residual)) < eps: break`).
UD_CAM_FN void ud_cam_radial_step(const float* k, int nk, float rnorm, float& th, float& delta) {
  float dth;
  const float res = ud_cam_radial(k, nk, th, dth) - rnorm;
  const float safe = fabsf(dth) < UD_CAM_EPS ? UD_CAM_EPS : dth;
  const float step = -res / safe;
  const float pred = -(res * step);
  const float sn = fabsf(step);
  const float step_scaled = sn > delta ? step * (delta / sn) : step;
  const float th_new = th + step_scaled;
  const float res_new = ud_cam_radial_residual(k, nk, th_new, rnorm);
  const float actual = fabsf(res) - fabsf(res_new);
  float rho = actual / pred;
  if (actual == 0.0f && pred == 0.0f) rho = 1.0f;
  if (rho > 0.5f) delta = fminf(2.0f * delta, 1.0f);
  if (rho < 0.2f) delta = 0.25f * delta;
  if (rho > 0.1f) th = th_new;
}

// Final direction of the OPENCV (tan_theta = 0) / Fisheye624 (tan_theta = 1) models, then Camera.get_rays' normalisation
// (camera.py:693-698, :971-976, :88-92).
UD_CAM_FN void ud_cam_finish_radial(float xr, float yr, float rnorm, float th, int tan_theta, float& x, float& y, float& z) {
  const bool close = fabsf(th) < UD_CAM_EPS && fabsf(rnorm) < UD_CAM_EPS;
  const float sc = (tan_theta ? tanf(th) : th) / rnorm;
  const float dx = close ? xr : sc * xr, dy = close ? yr : sc * yr;
  const float inv = 1.0f / fmaxf(sqrtf(dx * dx + dy * dy + 1.0f), 1e-4f);
  x = dx * inv; y = dy * inv; z = inv;
}

// MEI (unified omnidirectional) model, fully per pixel: 20 tangential Newton iterations, 20 radial ones, lifting to the unit
// sphere model with mirror parameter xi (camera.py:985-1082), then get_rays' normalisation.
UD_CAM_FN void ud_cam_mei(const float* p, float u, float v, float& x, float& y, float& z) {
  const float k1 = p[4], k2 = p[5], p0 = p[6], p1 = p[7], xi = p[8];
  const int use_radial = fabsf(k1) + fabsf(k2) > 1e-6f;
  const int use_tan = fabsf(p0) + fabsf(p1) > 1e-6f;
  const float ud = (u - p[2]) / p[0], vd = (v - p[3]) / p[1];
  float xr, yr;
  ud_cam_undistort_tanprism(ud, vd, p0, p1, 0.f, 0.f, 0.f, 0.f, 1, 0, use_tan ? 20 : 0, xr, yr);
  const float rnorm = sqrtf(xr * xr + yr * yr);
  float th = rnorm;
  for (int it = 0; it < (use_radial ? 20 : 0); ++it) {
    const float t2 = th * th, t4 = t2 * t2;
    const float thr = (1.0f + k1 * t2 + k2 * t4) * th;
    const float dth = 1.0f + 3.0f * k1 * t2 + 5.0f * k2 * t4;
    float step = (rnorm - thr) / dth;
    if (!(fabsf(dth) > UD_CAM_EPS_MEI)) step = (step > 0.f ? 1.f : (step < 0.f ? -1.f : 0.f)) * UD_CAM_EPS_MEI * 10.0f;
    th = th + step;
  }
  const bool close = fabsf(th) < UD_CAM_EPS_MEI && fabsf(rnorm) < UD_CAM_EPS_MEI;
  const float dx = close ? xr : th * xr / rnorm, dy = close ? yr : th * yr / rnorm;
  const float rho = sqrtf(dx * dx + dy * dy);
  const float rho2 = rho * rho;
  const float sq = sqrtf(1.0f + (1.0f - xi * xi) * rho2);
  float pz = 1.0f - xi * (rho2 + 1.0f) / (xi + sq);
  if (xi == 1.0f) pz = (1.0f - rho2) / 2.0f;
  const float inv = 1.0f / fmaxf(sqrtf(dx * dx + dy * dy + pz * pz), 1e-4f);
  x = dx * inv; y = dy * inv; z = pz * inv;
}
