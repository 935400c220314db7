// TEST INFRASTRUCTURE ONLY: host build of unidepth_amd/csrc/camera_models.h (the arithmetic the ray kernels run per pixel),
// driven the way pointwise.hip drives it -- init pass, up to 10 step passes gated by the image-wide maximum residual, final
// pass -- so tests/test_camera_models_cpu.py can check it against the oracle without a GPU.  Never loaded by the product.
#include "../../unidepth_amd/csrc/camera_models.h"
#include <vector>

extern "C" int cam_host_rays(const float* p, int model, int Hn, int Wn, float* rays) {
  const int HW = Hn * Wn;
  if (model == 6) {
    for (int pix = 0; pix < HW; ++pix) {
      const int v = pix / Wn, u = pix - v * Wn;
      ud_cam_mei(p, (float)u + 0.5f, (float)v + 0.5f, rays[pix], rays[HW + pix], rays[2 * HW + pix]);
    }
    return 0;
  }
  const int nk = model == 5 ? 6 : 3;
  const int use_tan = fabsf(p[10]) + fabsf(p[11]) > 1e-6f;
  const int use_prism = fabsf(p[12]) + fabsf(p[13]) + fabsf(p[14]) + fabsf(p[15]) > 1e-6f;
  const int use_radial = fabsf(p[4]) + fabsf(p[5]) + fabsf(p[6]) + fabsf(p[7]) + fabsf(p[8]) + fabsf(p[9]) > 1e-6f;
  std::vector<float> xr(HW), yr(HW), th(HW), delta(HW, 0.1f);
  int steps = 0;
  for (int pix = 0; pix < HW; ++pix) {
    const int v = pix / Wn, u = pix - v * Wn;
    const float ud = ((float)u + 0.5f - p[2]) / p[0], vd = ((float)v + 0.5f - p[3]) / p[1];
    ud_cam_undistort_tanprism(ud, vd, p[10], p[11], p[12], p[13], p[14], p[15], use_tan, use_prism, (use_tan || use_prism) ? 10 : 0, xr[pix], yr[pix]);
    th[pix] = sqrtf(xr[pix] * xr[pix] + yr[pix] * yr[pix]);
  }
  for (int it = 0; it < (use_radial ? 10 : 0); ++it) {
    float mx = 0.0f;
    bool nan = false;
    for (int pix = 0; pix < HW; ++pix) {
      const float rn = sqrtf(xr[pix] * xr[pix] + yr[pix] * yr[pix]);
      const float r = fabsf(ud_cam_radial_residual(p + 4, nk, th[pix], rn));
      if (r != r) nan = true;
      if (r > mx) mx = r;
    }
    if (!nan && mx < UD_CAM_EPS) break;
    ++steps;
    for (int pix = 0; pix < HW; ++pix) {
      const float rn = sqrtf(xr[pix] * xr[pix] + yr[pix] * yr[pix]);
      ud_cam_radial_step(p + 4, nk, rn, th[pix], delta[pix]);
    }
  }
  for (int pix = 0; pix < HW; ++pix) {
    const float rn = sqrtf(xr[pix] * xr[pix] + yr[pix] * yr[pix]);
    ud_cam_finish_radial(xr[pix], yr[pix], rn, th[pix], model == 5, rays[pix], rays[HW + pix], rays[2 * HW + pix]);
  }
  return steps;
}
