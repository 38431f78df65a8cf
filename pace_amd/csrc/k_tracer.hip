// Horizontal tracer advection (Fortran tracer_2d_1l): the stencils of fv3core/pace/fv3core/stencils/tracer_2d_1l.py:19-170
// around FiniteVolumeTransport(hord_tr = 8) (the monotone-PPM instance of the transport kernel, k_fvtp2d.hip).
// All HBM-bound streaming passes, one thread per (i, j, k), i fastest.
#include "common.h"
#include "kernels.h"

// flux_compute (tracer_2d_1l.py:19-77)
__global__ void __launch_bounds__(256)
k_tracer_flux_compute(Geo g, Met m, const real* __restrict__ cx, const real* __restrict__ cy, real* __restrict__ xfx,
                      real* __restrict__ yfx) {
  PLANE_IJK(g);
  const long c = IDX3(g, i, j, k);
  const long c2 = IDX2(g, i, j);
  if (i >= g.is && i <= g.ie + 1 && j >= g.js - 3 && j <= g.je + 3) {
    const double v = cx[c];
    xfx[c] = (v > 0.0) ? v * m.dxa[c2 - 1] * m.dy[c2] * m.sin_sg3[c2 - 1] : v * m.dxa[c2] * m.dy[c2] * m.sin_sg1[c2];
  }
  if (i >= g.is - 3 && i <= g.ie + 3 && j >= g.js && j <= g.je + 1) {
    const double v = cy[c];
    yfx[c] = (v > 0.0) ? v * m.dya[c2 - g.sj] * m.dx[c2] * m.sin_sg4[c2 - g.sj] : v * m.dya[c2] * m.dx[c2] * m.sin_sg2[c2];
  }
}

// divide_fluxes_by_n_substeps (tracer_2d_1l.py:80-106): origin_full, domain_full(add = (1, 1, 0)) = the whole storage plane
__global__ void __launch_bounds__(256)
k_tracer_divide(Geo g, real* __restrict__ a0, real* __restrict__ a1, real* __restrict__ a2, real* __restrict__ a3,
                real* __restrict__ a4, real* __restrict__ a5, double frac) {
  PLANE_IJK(g);
  const long c = IDX3(g, i, j, k);
  a0[c] = a0[c] * frac;
  a1[c] = a1[c] * frac;
  a2[c] = a2[c] * frac;
  a3[c] = a3[c] * frac;
  a4[c] = a4[c] * frac;
  a5[c] = a5[c] * frac;
}

// apply_mass_flux (tracer_2d_1l.py:115-135), compute domain
__global__ void __launch_bounds__(256)
k_apply_mass_flux(Geo g, Met m, const real* __restrict__ dp1, const real* __restrict__ mfx, const real* __restrict__ mfy,
                  real* __restrict__ dp2) {
  PATCH_IJK(g);
  if (i < g.is || i > g.ie || j < g.js || j > g.je) return;
  const long c = IDX3(g, i, j, k);
  dp2[c] = dp1[c] + (mfx[c] - mfx[c + 1] + mfy[c] - mfy[c + g.sj]) * m.rarea[IDX2(g, i, j)];
}

// apply_tracer_flux (tracer_2d_1l.py:138-158), compute domain
__global__ void __launch_bounds__(256)
k_apply_tracer_flux(Geo g, Met m, real* __restrict__ q, const real* __restrict__ dp1, const real* __restrict__ fx,
                    const real* __restrict__ fy, const real* __restrict__ dp2) {
  PATCH_IJK(g);
  if (i < g.is || i > g.ie || j < g.js || j > g.je) return;
  const long c = IDX3(g, i, j, k);
  q[c] = (q[c] * dp1[c] + (fx[c] - fx[c + 1] + fy[c] - fy[c + g.sj]) * m.rarea[IDX2(g, i, j)]) / dp2[c];
}

// swap_dp (tracer_2d_1l.py:166-170), compute domain
__global__ void __launch_bounds__(256) k_swap_dp(Geo g, real* __restrict__ dp1, real* __restrict__ dp2) {
  PLANE_IJK(g);
  if (i < g.is || i > g.ie || j < g.js || j > g.je) return;
  const long c = IDX3(g, i, j, k);
  const double t = dp1[c];
  dp1[c] = dp2[c];
  dp2[c] = t;
}

int launch_tracer_flux_compute(const Geo& g, const Met& m, const real* cx, const real* cy, real* xfx, real* yfx,
                               hipStream_t st) {
  hipLaunchKernelGGL(k_tracer_flux_compute, plane_grid(g, g.nk), dim3(256), 0, st, g, m, cx, cy, xfx, yfx);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}
int launch_tracer_divide(const Geo& g, real* cxd, real* xfx, real* mfxd, real* cyd, real* yfx, real* mfyd,
                         int n_split, hipStream_t st) {
  hipLaunchKernelGGL(k_tracer_divide, plane_grid(g, g.nk), dim3(256), 0, st, g, cxd, xfx, mfxd, cyd, yfx, mfyd, 1.0 / n_split);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}
int launch_apply_mass_flux(const Geo& g, const Met& m, const real* dp1, const real* mfx, const real* mfy, real* dp2,
                           hipStream_t st) {
  hipLaunchKernelGGL(k_apply_mass_flux, patch_grid(g, g.nk), PATCH_BLOCK, 0, st, g, m, dp1, mfx, mfy, dp2);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}
int launch_apply_tracer_flux(const Geo& g, const Met& m, real* q, const real* dp1, const real* fx, const real* fy,
                             const real* dp2, hipStream_t st) {
  hipLaunchKernelGGL(k_apply_tracer_flux, patch_grid(g, g.nk), PATCH_BLOCK, 0, st, g, m, q, dp1, fx, fy, dp2);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}
int launch_swap_dp(const Geo& g, real* dp1, real* dp2, hipStream_t st) {
  hipLaunchKernelGGL(k_swap_dp, plane_grid(g, g.nk), dim3(256), 0, st, g, dp1, dp2);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}
