// XPiecewiseParabolic / YPiecewiseParabolic as stand-alone operators (xppm.py:290-355, yppm.py:290-355): the mean value of q
// advected through every x- (y-) interface of a window, one thread per interface.  Inside the fused transport kernel
// (k_fvtp2d.hip) the same point functions (common.h: ppm_run / ppm_run8) run on LDS tiles; this entry exists for callers that
// use the PPM operators by themselves, as the reference's Translate tests XPPM / YPPM do.  Corner halos are the caller's
// business here, exactly as in the reference class (fvtp2d copies them before calling).
#include "common.h"
#include "kernels.h"

template <int MORD, int AXIS>
__global__ void __launch_bounds__(256)
k_ppm1d(Geo g, Met m, const real* __restrict__ q, const real* __restrict__ c, real* __restrict__ out, int i0, int j0, int k0,
        int ni, int nj) {
  const int i = i0 + blockIdx.x * 64 + (threadIdx.x & 63);
  const int j = j0 + blockIdx.y * 4 + (threadIdx.x >> 6);
  const int k = k0 + blockIdx.z;
  if (i >= i0 + ni || j >= j0 + nj) return;
  const long cc = IDX3(g, i, j, k);
  const long step = (AXIS == 0) ? 1 : g.sj;
  const int pos = (AXIS == 0) ? i : j;
  const int lim = (AXIS == 0) ? g.ni : g.nj;
  // cells pos-3 .. pos+2; values outside the storage are never used by an interface the reference computes (its windows
  // stay 3 cells inside), they are read clamped
  double Q[6];
#pragma unroll
  for (int u = 0; u < 6; ++u) {
    int p = pos - 3 + u;
    p = p < 0 ? 0 : (p > lim - 1 ? lim - 1 : p);
    Q[u] = q[cc + (long)(p - pos) * step];
  }
  const double cv = c[cc];
  double res;
  if (AXIS == 0) {
    const real* dxa = m.dxa;
    auto d = [=](int p) { return dxa[IDX2(g, p, j)]; };
    if (MORD == 8) ppm_run8<true, 1>(Q, &cv, pos, g.is, g.ie, d, &res);
    else res = ppm_flux6<MORD, true>(Q, cv, pos, g.is, g.ie, d);
  } else {
    const real* dya = m.dya;
    auto d = [=](int p) { return dya[IDX2(g, i, p)]; };
    if (MORD == 8) ppm_run8<true, 1>(Q, &cv, pos, g.js, g.je, d, &res);
    else res = ppm_flux6<MORD, true>(Q, cv, pos, g.js, g.je, d);
  }
  out[cc] = res;
}

int launch_ppm1d(const Geo& g, const Met& m, int axis, int iord, const real* q, const real* c, real* out, int i0, int j0,
                 int k0, int ni, int nj, int nk, hipStream_t st) {
  const dim3 grid((ni + 63) / 64, (nj + 3) / 4, nk), block(256);
#define GO(M, A) hipLaunchKernelGGL((k_ppm1d<M, A>), grid, block, 0, st, g, m, q, c, out, i0, j0, k0, ni, nj)
  const int mord = iord < 0 ? -iord : iord;
  if (axis == 0) {
    if (mord == 5) GO(5, 0); else if (mord == 6) GO(6, 0); else if (mord == 8) GO(8, 0); else return PACE_ERR_UNSUPPORTED;
  } else {
    if (mord == 5) GO(5, 1); else if (mord == 6) GO(6, 1); else if (mord == 8) GO(8, 1); else return PACE_ERR_UNSUPPORTED;
  }
#undef GO
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}
