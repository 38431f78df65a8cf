// EXPERIMENT (not on the product path): a wave-private, barrier-free formulation of FiniteVolumeTransport for the
// interior of a tile -- the design question left open in DESIGN.md section 8.  One wave = 64 columns (lanes along i),
// marching through the rows of a segment: the two y sweeps work on 6-row register windows of one lane, the two x sweeps
// get their i-neighbours from the adjacent lanes with wave shuffles; no LDS, no barriers, every global access is one
// coalesced row.  Plain transport (fvtp2d.py:262-345 with ord 6, no damping, fluxes written), interior forms only:
// valid where no PPM stencil reaches within 3 cells of a tile edge.  Same expressions as k_fvtp2d.hip -> same bits.
#include "common.h"
#include "kernels.h"

#ifndef PACE_EMU
#define MW 58  // output columns per wave (lanes 3 .. 60)
#ifndef MH
#define MH 12  // output rows per wave
#endif

__device__ __forceinline__ double lane_up(double v) { return __shfl_up(v, 1); }     // value of lane l-1
__device__ __forceinline__ double lane_dn(double v) { return __shfl_down(v, 1); }   // value of lane l+1

// x sweep of one row held one value per lane: the mean value advected through the lane's WEST interface
__device__ __forceinline__ double xsweep(double q0, double c) {
  const double qm1 = lane_up(q0), qm2 = lane_up(qm1), qp1 = lane_dn(q0);
  const double al0 = PPM_P1 * (qm1 + q0) + PPM_P2 * (qm2 + qp1);  // interface value at the west face
  const double alp = lane_dn(al0);                                // east face
  const double bl = al0 - q0, br = alp - q0, b0 = bl + br;
  const bool steep = (3.0 * fabs(b0)) < fabs(bl - br);
  // the cell to the west
  const double br_m = lane_up(br), b0_m = lane_up(b0);
  const bool steep_m = __shfl_up((int)steep, 1) != 0;
  const double mask = (steep_m || steep) ? 1.0 : 0.0;
  if (c > 0.0) {
    const double fx1 = (1.0 - c) * (br_m - c * b0_m);
    return qm1 + fx1 * mask;
  }
  const double fx1 = (1.0 + c) * (bl + c * b0);
  return q0 + fx1 * mask;
}

// y sweep: W[0..5] = rows j-3 .. j+2 of one column; the mean value advected through y-interface j
__device__ __forceinline__ double ysweep(const double* W, double c) {
  const double al_m = PPM_P1 * (W[1] + W[2]) + PPM_P2 * (W[0] + W[3]);
  const double al_0 = PPM_P1 * (W[2] + W[3]) + PPM_P2 * (W[1] + W[4]);
  const double al_p = PPM_P1 * (W[3] + W[4]) + PPM_P2 * (W[2] + W[5]);
  const double bl_m = al_m - W[2], br_m = al_0 - W[2], b0_m = bl_m + br_m;
  const double bl_0 = al_0 - W[3], br_0 = al_p - W[3], b0_0 = bl_0 + br_0;
  const bool s_m = (3.0 * fabs(b0_m)) < fabs(bl_m - br_m);
  const bool s_0 = (3.0 * fabs(b0_0)) < fabs(bl_0 - br_0);
  const double mask = (s_m || s_0) ? 1.0 : 0.0;
  if (c > 0.0) {
    const double fx1 = (1.0 - c) * (br_m - c * b0_m);
    return W[2] + fx1 * mask;
  }
  const double fx1 = (1.0 + c) * (bl_0 + c * b0_0);
  return W[3] + fx1 * mask;
}

__global__ void __launch_bounds__(64)
k_fvtp2d_march(Geo g, Met m, const real* __restrict__ q, const real* __restrict__ crx, const real* __restrict__ cry,
               const real* __restrict__ xfx, const real* __restrict__ yfx, real* __restrict__ fx, real* __restrict__ fy,
               int ib, int nx, int jb, int ny) {
  const int lane = threadIdx.x;
  const int gi = ib + blockIdx.x * MW - 3 + lane;
  const int j0 = jb + blockIdx.y * MH;
  const int k = blockIdx.z;
  const unsigned sj8 = (unsigned)g.sj * (unsigned)sizeof(real);
  const unsigned col = (unsigned)((long)k * g.sk * (long)sizeof(real)) + (unsigned)gi * (unsigned)sizeof(real);  // byte offset of (gi, row 0, k)
  const unsigned col2 = (unsigned)gi * (unsigned)sizeof(real);
#define LDF(p, row) (*(const real*)((const char*)(p) + (col + (unsigned)(row) * sj8)))
#define LD2(p, row) (*(const real*)((const char*)(p) + (col2 + (unsigned)(row) * sj8)))
#define STF(p, row) (*(real*)((char*)(p) + (col + (unsigned)(row) * sj8)))
  const bool out_lane = lane >= 3 && lane <= 60 && gi < ib + nx;
  double Wq[6], Wj[6];      // q and q_j on rows r-5 .. r
  double sx[4];             // inner x values of rows r-3 .. r
  double syin_prev = 0.0, yfx_prev = 0.0;
#pragma unroll
  for (int t = 0; t < 6; ++t) Wq[t] = Wj[t] = 0.0;
#pragma unroll
  for (int t = 0; t < 4; ++t) sx[t] = 0.0;
  const int jend = min(j0 + MH, jb + ny);  // one past the last output row
  for (int r = j0 - 3; r <= jend + 2; ++r) {
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      Wq[t] = Wq[t + 1];
      Wj[t] = Wj[t + 1];
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) sx[t] = sx[t + 1];
    // (a) row r: inner x sweep and q advected in x
    const double qr = LDF(q, r);
    Wq[5] = qr;
    const double sxr = xsweep(qr, LDF(crx, r));
    sx[3] = sxr;
    {
      const double x0 = LDF(xfx, r), x1 = lane_dn(x0);
      const double a = LD2(m.area, r);
      Wj[5] = (qr * a + x0 * sxr - x1 * lane_dn(sxr)) / (a + x0 - x1);
    }
    const int jy = r - 2;  // the y-interface whose stencil is complete now
    if (jy < j0) continue;
    // (b) y sweeps at interface jy
    const double cy = LDF(cry, jy);
    const double syin = ysweep(Wq, cy);
    const double yfx_j = LDF(yfx, jy);
    if (jy < jend) {
      const double outer = ysweep(Wj, cy);
      if (out_lane) STF(fy, jy) = 0.5 * (outer + syin) * yfx_j;
    }
    // (c) + (d) row jy - 1: q advected in y, outer x sweep, x flux
    const int row = jy - 1;
    if (row >= j0) {
      const double a = LD2(m.area, row);
      const double qi = (Wq[2] * a + yfx_prev * syin_prev - yfx_j * syin) / (a + yfx_prev - yfx_j);
      const double outer = xsweep(qi, LDF(crx, row));
      if (out_lane) STF(fx, row) = 0.5 * (outer + sx[0]) * LDF(xfx, row);
    }
    syin_prev = syin;
    yfx_prev = yfx_j;
  }
#undef LDF
#undef LD2
#undef STF
}
#endif

int launch_fvtp2d_march(const Geo& g, const Met& m, const real* q, const real* crx, const real* cry, const real* xfx,
                        const real* yfx, real* fx, real* fy, int ib, int nx, int jb, int ny, int nlev, hipStream_t st) {
#ifdef PACE_EMU
  return PACE_ERR_UNSUPPORTED;
#else
  // interior only: every stencil stays 3+ cells away from the tile edges
  if (ib - 3 < g.is + 3 || ib + nx + 3 > g.ie - 2 || jb - 3 < g.js + 3 || jb + ny + 3 > g.je - 2) return PACE_ERR_ARG;
  const dim3 grid((nx + MW - 1) / MW, (ny + MH - 1) / MH, nlev);
  hipLaunchKernelGGL(k_fvtp2d_march, grid, dim3(64), 0, st, g, m, q, crx, cry, xfx, yfx, fx, fy, ib, nx, jb, ny);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
#endif
}
