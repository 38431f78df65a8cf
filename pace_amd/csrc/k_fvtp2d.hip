// FiniteVolumeTransport (Fortran fv_tp_2d): Putman-Lin 2-D transport fluxes of one scalar.
// Reference: fv3core/pace/fv3core/stencils/fvtp2d.py:262-345 -- 9 stencil launches (copy_corners_y,
// yppm inner, q_i, xppm outer, copy_corners_x, xppm inner, q_j, yppm outer, final_fluxes) through 6
// full 3-D temporaries.  Here: ONE kernel.  A workgroup owns a TI x TJ tile of one level, stages the
// (TI+6) x (TJ+6) footprint of q in LDS, runs the five sweeps tile-locally and writes only the two
// flux fields.  HBM-bound: algorithmic traffic = 5-7 reads + 2 writes of 3-D fields.
#include "common.h"
#include "kernels.h"

#ifndef FV_TI
#define FV_TI 32
#define FV_TJ 16
#endif
#define TI FV_TI
#define TJ FV_TJ
#define QW (TI + 6)
#define QH (TJ + 6)

// EX / EY: whether any x- (y-) interface this workgroup evaluates lies within two cells of a tile edge, where the
// PPM interface values switch to the one-sided forms (xppm.py:148-181).  Block-uniform, so interior workgroups
// (25 of 35 at C192) run straight-line code without the per-interface position tests.
struct FvLds {
  double sq[QH][QW + 1];        // q on [i0-3, i0+TI+3) x [j0-3, j0+TJ+3)
  double syin[TJ + 1][QW + 1];  // inner y sweep: mean advected value on y-interfaces
  double sqi[TJ][QW + 1];       // q advected in y (fvtp2d.py:34-56)
  double sxin[QH][TI + 2];      // inner x sweep on x-interfaces
  double sqj[QH][TI + 1];       // q advected in x (fvtp2d.py:59-77)
};

#define RF 4                              // interfaces per thread in a PPM run
#define GY ((TJ + 1 + RF - 1) / RF)       // runs per column for the TJ+1 y-interfaces
#define GX ((TI + 1 + RF - 1) / RF)       // runs per row for the TI+1 x-interfaces
static_assert(QW * GY <= 256 && QH * GX <= 256, "one PPM run per thread: the tile is too large for 256 threads");

template <int MORD, bool EX, bool EY>
__device__ __forceinline__ void fvtp2d_tile(FvLds& L, const Geo& g, const Met& m, const double* __restrict__ q,
                                            const double* __restrict__ crx, const double* __restrict__ cry,
                                            const double* __restrict__ xfx, const double* __restrict__ yfx,
                                            double* __restrict__ fx, double* __restrict__ fy,
                                            const double* __restrict__ xunit, const double* __restrict__ yunit) {
  auto& sq = L.sq;
  auto& syin = L.syin;
  auto& sqi = L.sqi;
  auto& sxin = L.sxin;
  auto& sqj = L.sqj;

  const int tid = threadIdx.x;
  const int i0 = g.is + blockIdx.x * TI;
  const int j0 = g.js + blockIdx.y * TJ;
  const int k = blockIdx.z;
  const long kb = (long)k * g.sk;
  const int ilo = i0 - 3, jlo = j0 - 3;
  const int sj = g.sj;

  // stage 0: q with corners copied in the y direction (copy_corners_y, corners.py:367-425)
  for (int e = tid; e < QW * QH; e += 256) {
    const int jj = e / QW, ii = e - jj * QW;
    int gi = ilo + ii, gj = jlo + jj;
    double v = 0.0;
    if (gi >= 0 && gi < g.ni && gj >= 0 && gj < g.nj) {
      remap_agrid_y(g, gi, gj);
      v = q[kb + IDX2(g, gi, gj)];
    }
    sq[jj][ii] = v;
  }
  __syncthreads();

  // stage 1: inner y sweep (YPiecewiseParabolic, origin (is-3, js), domain (N+7, N+1)): one run of RF interfaces of one
  // column per thread, lanes along i
  if (tid < QW * GY) {
    const int grp = tid / QW, ii = tid - grp * QW;
    const int jj0 = grp * RF;
    const int gi = ilo + ii, gj0 = j0 + jj0;
    const bool col_ok = gi >= 0 && gi <= g.ni - 1;
    double Q[RF + 5], cc[RF], out[RF];
#pragma unroll
    for (int u = 0; u < RF + 5; ++u) Q[u] = sq[(jj0 + u < QH) ? jj0 + u : QH - 1][ii];  // rows gj0-3 ..
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      const int gj = gj0 + f;
      cc[f] = (col_ok && jj0 + f <= TJ && gj >= g.js && gj <= g.je + 1) ? cry[kb + IDX2(g, gi, gj)] : 0.0;
    }
    const double* dya = m.dya;
    const long col = gi;
    ppm_run<MORD, EY, RF>(Q, cc, gj0, g.js, g.je, [=](int p) { return dya[col + (long)p * sj]; }, out);
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      const int gj = gj0 + f;
      if (jj0 + f <= TJ) syin[jj0 + f][ii] = (col_ok && gj >= g.js && gj <= g.je + 1) ? out[f] : 0.0;
    }
  }
  __syncthreads();

  // stage 2: q_i; and re-stage the corner cells of q for the x direction (copy_corners_x)
  for (int e = tid; e < QW * TJ; e += 256) {
    const int jj = e / QW, ii = e - jj * QW;
    const int gi = ilo + ii, gj = j0 + jj;
    double val = 0.0;
    if (gi >= 0 && gi <= g.ni - 1 && gj >= g.js && gj <= g.je) {
      const long c2 = IDX2(g, gi, gj);
      const double y0 = yfx[kb + c2], y1 = yfx[kb + c2 + g.sj];
      const double a = m.area[c2];
      val = (sq[jj + 3][ii] * a + y0 * syin[jj][ii] - y1 * syin[jj + 1][ii]) / (a + y0 - y1);
    }
    sqi[jj][ii] = val;
  }
  {
    const bool icorner_tile = (i0 == g.is) || (i0 + TI + 3 > g.ie + 1);
    const bool jcorner_tile = (j0 == g.js) || (j0 + TJ + 3 > g.je + 1);
    if (icorner_tile && jcorner_tile) {
      for (int e = tid; e < QW * QH; e += 256) {
        const int jj = e / QW, ii = e - jj * QW;
        int gi = ilo + ii, gj = jlo + jj;
        if (gi >= 0 && gi < g.ni && gj >= 0 && gj < g.nj && (gi < g.is || gi > g.ie) && (gj < g.js || gj > g.je)) {
          remap_agrid_x(g, gi, gj);
          sq[jj][ii] = q[kb + IDX2(g, gi, gj)];
        }
      }
    }
  }
  __syncthreads();

  // stage 3: inner x sweep (XPiecewiseParabolic, origin (is, js-3), domain (N+1, N+7)): one run of RF interfaces of one
  // row per thread
  if (tid < QH * GX) {
    const int jj = tid / GX, grp = tid - jj * GX;
    const int ii0 = grp * RF;
    const int gi0 = i0 + ii0, gj = jlo + jj;
    const bool row_ok = gj >= 0 && gj <= g.nj - 1;
    double Q[RF + 5], cc[RF], out[RF];
#pragma unroll
    for (int u = 0; u < RF + 5; ++u) Q[u] = sq[jj][(ii0 + u < QW) ? ii0 + u : QW - 1];  // columns gi0-3 ..
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      const int gi = gi0 + f;
      cc[f] = (row_ok && ii0 + f <= TI && gi >= g.is && gi <= g.ie + 1) ? crx[kb + IDX2(g, gi, gj)] : 0.0;
    }
    const double* dxa = m.dxa + (long)gj * sj;
    ppm_run<MORD, EX, RF>(Q, cc, gi0, g.is, g.ie, [=](int p) { return dxa[p]; }, out);
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      const int gi = gi0 + f;
      if (ii0 + f <= TI) sxin[jj][ii0 + f] = (row_ok && gi >= g.is && gi <= g.ie + 1) ? out[f] : 0.0;
    }
  }
  __syncthreads();

  // stage 4: q_j
  for (int e = tid; e < TI * QH; e += 256) {
    const int jj = e / TI, ii = e - jj * TI;
    const int gi = i0 + ii, gj = jlo + jj;
    double val = 0.0;
    if (gj >= 0 && gj <= g.nj - 1 && gi >= g.is && gi <= g.ie) {
      const long c2 = IDX2(g, gi, gj);
      const double x0 = xfx[kb + c2], x1 = xfx[kb + c2 + 1];
      const double a = m.area[c2];
      val = (sq[jj][ii + 3] * a + x0 * sxin[jj][ii] - x1 * sxin[jj][ii + 1]) / (a + x0 - x1);
    }
    sqj[jj][ii] = val;
  }
  __syncthreads();

  // stage 5: outer sweeps + final_fluxes (fvtp2d.py:80-119).  The grid has ceil(N / TI) x ceil(N / TJ) workgroups; the
  // N+1-th face row / column (ie+1, je+1) is produced by the workgroup that owns cell ie / je, not by an extra,
  // almost empty row of workgroups.
  if (tid < TJ * GX) {  // outer x on q_i: rows of the tile, runs of x-interfaces
    const int jj = tid / GX, grp = tid - jj * GX;
    const int ii0 = grp * RF;
    const int gi0 = i0 + ii0, gj = j0 + jj;
    double Q[RF + 5], cc[RF], xu[RF], out[RF];
    bool ok[RF];
#pragma unroll
    for (int u = 0; u < RF + 5; ++u) Q[u] = sqi[jj][(ii0 + u < QW) ? ii0 + u : QW - 1];  // q_i at gi0-3 ..
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      const int ii = ii0 + f, gi = gi0 + f;
      ok[f] = gj <= g.je && gi <= g.ie + 1 && (ii < TI || (ii == TI && gi == g.ie + 1));
      const long c = kb + IDX2(g, gi, gj);
      cc[f] = ok[f] ? crx[c] : 0.0;
      xu[f] = ok[f] ? xunit[c] : 0.0;
    }
    const double* dxa = m.dxa + (long)gj * sj;
    ppm_run<MORD, EX, RF>(Q, cc, gi0, g.is, g.ie, [=](int p) { return dxa[p]; }, out);
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      if (ok[f]) fx[kb + IDX2(g, gi0 + f, gj)] = 0.5 * (out[f] + sxin[jj + 3][ii0 + f]) * xu[f];
    }
  }
  if (tid < TI * GY) {  // outer y on q_j: columns of the tile, runs of y-interfaces, lanes along i
    const int grp = tid / TI, ii = tid - grp * TI;
    const int jj0 = grp * RF;
    const int gi = i0 + ii, gj0 = j0 + jj0;
    double Q[RF + 5], cc[RF], yu[RF], out[RF];
    bool ok[RF];
#pragma unroll
    for (int u = 0; u < RF + 5; ++u) Q[u] = sqj[(jj0 + u < QH) ? jj0 + u : QH - 1][ii];  // q_j at gj0-3 ..
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      const int jj = jj0 + f, gj = gj0 + f;
      ok[f] = gi <= g.ie && gj <= g.je + 1 && (jj < TJ || (jj == TJ && gj == g.je + 1));
      const long c = kb + IDX2(g, gi, gj);
      cc[f] = ok[f] ? cry[c] : 0.0;
      yu[f] = ok[f] ? yunit[c] : 0.0;
    }
    const double* dya = m.dya;
    const long col = gi;
    ppm_run<MORD, EY, RF>(Q, cc, gj0, g.js, g.je, [=](int p) { return dya[col + (long)p * sj]; }, out);
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      if (ok[f]) fy[kb + IDX2(g, gi, gj0 + f)] = 0.5 * (out[f] + syin[jj0 + f][ii + 3]) * yu[f];
    }
  }
}

template <int MORD>
__global__ void __launch_bounds__(256) k_fvtp2d(Geo g, Met m, const double* __restrict__ q,
                                                const double* __restrict__ crx, const double* __restrict__ cry,
                                                const double* __restrict__ xfx, const double* __restrict__ yfx,
                                                double* __restrict__ fx, double* __restrict__ fy,
                                                const double* __restrict__ xunit, const double* __restrict__ yunit) {
  // x-interfaces evaluated: i0 .. i0+TI (their al's reach one further each way); special forms at is-1 .. is+1 and
  // ie .. ie+2
  __shared__ FvLds L;
  const int i0 = g.is + blockIdx.x * TI, j0 = g.js + blockIdx.y * TJ;
  const bool ex = (i0 - 1 <= g.is + 1) || (i0 + TI + 1 >= g.ie);
  const bool ey = (j0 - 1 <= g.js + 1) || (j0 + TJ + 1 >= g.je);
  if (ex && ey) fvtp2d_tile<MORD, true, true>(L, g, m, q, crx, cry, xfx, yfx, fx, fy, xunit, yunit);
  else if (ex) fvtp2d_tile<MORD, true, false>(L, g, m, q, crx, cry, xfx, yfx, fx, fy, xunit, yunit);
  else if (ey) fvtp2d_tile<MORD, false, true>(L, g, m, q, crx, cry, xfx, yfx, fx, fy, xunit, yunit);
  else fvtp2d_tile<MORD, false, false>(L, g, m, q, crx, cry, xfx, yfx, fx, fy, xunit, yunit);
}

int launch_fvtp2d(const Geo& g, const Met& m, const double* q, const double* crx, const double* cry,
                  const double* xfx, const double* yfx, double* fx, double* fy, const double* xmf,
                  const double* ymf, int hord, int nlev, hipStream_t st) {
  const dim3 grid((g.n + TI - 1) / TI, (g.n + TJ - 1) / TJ, nlev), block(256);
  const double* xu = xmf ? xmf : xfx;
  const double* yu = ymf ? ymf : yfx;
  if (hord == 5) {
    hipLaunchKernelGGL(k_fvtp2d<5>, grid, block, 0, st, g, m, q, crx, cry, xfx, yfx, fx, fy, xu, yu);
  } else if (hord == 6) {
    hipLaunchKernelGGL(k_fvtp2d<6>, grid, block, 0, st, g, m, q, crx, cry, xfx, yfx, fx, fy, xu, yu);
  } else {
    return PACE_ERR_UNSUPPORTED;
  }
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}
