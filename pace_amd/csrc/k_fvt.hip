// FiniteVolumeTransport for the production tilings -- the lean form of k_fvtp2d.hip (same arithmetic, same bits).
// (kernels and launchers; the device code and the design notes are in fvt_core.h)
#include "fvt_core.h"

#if FVT_AVAILABLE
namespace {

template <int MORD, int DMODE, int EPI>
__global__ void __launch_bounds__(256, 4) k_fvt(Geo g, FvMet m, const real* __restrict__ q, const real* __restrict__ crx,
                                                const real* __restrict__ cry, const real* __restrict__ xfx,
                                                const real* __restrict__ yfx, real* __restrict__ fx, real* __restrict__ fy,
                                                const real* __restrict__ xunit, const real* __restrict__ yunit, FvDamp dp) {
  __shared__ FvtLds L;
  const FvTile wg = fv_tile_of_workgroup();
  const int gx = g.n / TI, gy = g.n / TJ;
  const bool ex = wg.bx == 0 || wg.bx == gx - 1, ey = wg.by == 0 || wg.by == gy - 1;
  if (ex && ey) fvt_tile<MORD, DMODE, EPI, true, true>(L, g, m, q, crx, cry, xfx, yfx, fx, fy, xunit, yunit, dp, wg.bx, wg.by, wg.bz);
  else if (ex) fvt_tile<MORD, DMODE, EPI, true, false>(L, g, m, q, crx, cry, xfx, yfx, fx, fy, xunit, yunit, dp, wg.bx, wg.by, wg.bz);
  else if (ey) fvt_tile<MORD, DMODE, EPI, false, true>(L, g, m, q, crx, cry, xfx, yfx, fx, fy, xunit, yunit, dp, wg.bx, wg.by, wg.bz);
  else fvt_tile<MORD, DMODE, EPI, false, false>(L, g, m, q, crx, cry, xfx, yfx, fx, fy, xunit, yunit, dp, wg.bx, wg.by, wg.bz);
}

template <int MORD>
int fvt_launch_mode(int dmode, int epi, dim3 grid, hipStream_t st, const Geo& g, const FvMet& m, const real* q, const real* crx,
                    const real* cry, const real* xfx, const real* yfx, real* fx, real* fy, const real* xu, const real* yu,
                    const FvDamp& dp) {
#define FVT_GO(D, E)                                                                                                         \
  do {                                                                                                                       \
    hipLaunchKernelGGL((k_fvt<MORD, D, E>), grid, dim3(256), 0, st, g, m, q, crx, cry, xfx, yfx, fx, fy, xu, yu, dp);        \
    return PACE_OK;                                                                                                          \
  } while (0)
  if (epi == 0 && dmode == -1) FVT_GO(-1, 0);
  if (epi == 0 && dmode == 1) FVT_GO(1, 0);
  if (epi == 0 && dmode == 0) FVT_GO(0, 0);
  if (epi == 2 && dmode == 0) FVT_GO(3, 1);  // w: damping -> dw / heat_source / diss_est, plain transport -> cell update
  if (epi == 1 && dmode == 2) FVT_GO(2, 1);
#undef FVT_GO
  return PACE_ERR_UNSUPPORTED;
}

}  // namespace
#endif  // FVT_AVAILABLE

bool transport_lean_covers(const Geo& g, int hord) {
#if FVT_AVAILABLE
  return (hord == 5 || hord == 6) && g.n % TI == 0 && g.n % TJ == 0 && g.n >= 2 * TI && g.n >= 2 * TJ && (g.sj & 1) == 0 && (g.sk & 1) == 0;
#else
  (void)g, (void)hord;
  return false;
#endif
}

// Same contract as launch_transport (k_fvtp2d.hip), for the calls this kernel covers; PACE_ERR_UNSUPPORTED otherwise (the
// caller then takes the general kernel).  xmf / ymf resolved by the caller: xu / yu are the unit fluxes.
int launch_transport_lean(const Geo& g, const Met& m, const real* q, const real* crx, const real* cry, const real* xfx,
                          const real* yfx, real* fx, real* fy, const real* xu, const real* yu, int hord, int nlev, int dmode,
                          int epi, const FvDamp& dp, hipStream_t st) {
#if FVT_AVAILABLE
  if (!transport_lean_covers(g, hord) || ((uintptr_t)q & 15) != 0) return PACE_ERR_UNSUPPORTED;  // (16-byte rows of the footprint)
  if (dmode >= 0 && dp.nmax > 2) return PACE_ERR_UNSUPPORTED;
  const dim3 grid(g.n / TI, g.n / TJ, nlev);
  const FvMet fm = fv_met(m);
  int rc;
  if (hord == 5) rc = fvt_launch_mode<5>(dmode, epi, grid, st, g, fm, q, crx, cry, xfx, yfx, fx, fy, xu, yu, dp);
  else if (hord == 6) rc = fvt_launch_mode<6>(dmode, epi, grid, st, g, fm, q, crx, cry, xfx, yfx, fx, fy, xu, yu, dp);
  else return PACE_ERR_UNSUPPORTED;
  if (rc) return rc;
  PACE_CHECK_LAUNCH();
  return PACE_OK;
#else
  return PACE_ERR_UNSUPPORTED;
#endif
}
