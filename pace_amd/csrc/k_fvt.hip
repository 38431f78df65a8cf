// FiniteVolumeTransport for the production tilings -- the lean form of k_fvtp2d.hip (same arithmetic, same bits).
// (kernels and launchers; the device code and the design notes are in fvt_core.h)
// This source is compiled twice: as itself for the 32 x 24 tile (C96, C192, C384: N a multiple of 32 and of 24) and through
// k_fvt16.hip for a 16 x 24 tile (C48: N a multiple of 16 and of 24 only), each in a namespace of its own (the kernels' names
// carry it); the functions the rest of the library calls (kernels.h) are defined here once and try the shapes in that order.
#include "fvt_core.h"

#ifndef FVT_SHAPE
#define FVT_SHAPE 32
#endif
#define FVT_CAT_(a, b) a##b
#define FVT_CAT(a, b) FVT_CAT_(a, b)
#define FVT_FN(name) FVT_CAT(FVT_CAT(fvt, FVT_SHAPE), _##name)  // fvt32_covers, fvt16_launch_transport, ...

#if FVT_AVAILABLE
using namespace FVT_NS;
namespace FVT_NS {

#ifndef FVT_WAVES
#define FVT_WAVES 3  // workgroups per CU the register budget of the single-scalar kernel is set for (168 VGPRs: at 128 the
                     // damped instances spill, and a spilled register is 8 MB of scratch traffic per launch; measured 172 -> 128 us)
#endif
template <int MORD, int DMODE, int EPI>
__global__ void __launch_bounds__(256, FVT_WAVES) k_fvt(Geo g, FvMet m, const real* __restrict__ q, const real* __restrict__ crx,
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

#ifndef FVT_SCALARS_NT
#define FVT_SCALARS_NT 512  // threads per workgroup of the scalar-phase kernel: 512 = x-runs and y-runs in different waves
#endif
// two workgroups per CU either way (LDS: 78 KB each); 512 threads -> four waves per SIMD at <= 128 VGPRs, 256 -> two at <= 256.
// (__launch_bounds__(512, 2) alone does not hold the compiler to 128 registers -- it took 134 once, which is ONE workgroup per
// CU: 342 -> 447 us --, the waves-per-EU attribute does)
#ifdef PACE_EMU
#define FVT_SCALARS_ATTR
#elif FVT_SCALARS_NT == 512
#define FVT_SCALARS_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
#define FVT_SCALARS_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
#ifndef FVT_RESIDENT
#define FVT_RESIDENT 1  // the 512-thread form on the layout with the damping's planes resident (round 6; 0: round 5's, for A/B builds)
#endif
template <int MORD>
__global__ void __launch_bounds__(FVT_SCALARS_NT) FVT_SCALARS_ATTR k_fvt_scalars(Geo g, FvMet m, FvtScalars S) {
#if FVT_SCALARS_NT == 512 && FVT_RESIDENT
  __shared__ FvtLdsScalarsRes L;
#else
  __shared__ FvtLdsScalars L;
#endif
  const FvTile wg = fv_tile_of_workgroup();
  const int gx = g.n / TI, gy = g.n / TJ;
  const bool ex = wg.bx == 0 || wg.bx == gx - 1, ey = wg.by == 0 || wg.by == gy - 1;
#if FVT_SCALARS_NT == 512 && FVT_RESIDENT
  if (ex && ey) fvt_scalars_tile_res<MORD, true, true>(L, g, m, S, wg.bx, wg.by, wg.bz);
  else if (ex) fvt_scalars_tile_res<MORD, true, false>(L, g, m, S, wg.bx, wg.by, wg.bz);
  else if (ey) fvt_scalars_tile_res<MORD, false, true>(L, g, m, S, wg.bx, wg.by, wg.bz);
  else fvt_scalars_tile_res<MORD, false, false>(L, g, m, S, wg.bx, wg.by, wg.bz);
#elif FVT_SCALARS_NT == 512
  if (ex && ey) fvt_scalars_tile_split<MORD, true, true>(L, g, m, S, wg.bx, wg.by, wg.bz);
  else if (ex) fvt_scalars_tile_split<MORD, true, false>(L, g, m, S, wg.bx, wg.by, wg.bz);
  else if (ey) fvt_scalars_tile_split<MORD, false, true>(L, g, m, S, wg.bx, wg.by, wg.bz);
  else fvt_scalars_tile_split<MORD, false, false>(L, g, m, S, wg.bx, wg.by, wg.bz);
#else
  if (ex && ey) fvt_scalars_tile<MORD, true, true>(L, g, m, S, wg.bx, wg.by, wg.bz);
  else if (ex) fvt_scalars_tile<MORD, true, false>(L, g, m, S, wg.bx, wg.by, wg.bz);
  else if (ey) fvt_scalars_tile<MORD, false, true>(L, g, m, S, wg.bx, wg.by, wg.bz);
  else fvt_scalars_tile<MORD, false, false>(L, g, m, S, wg.bx, wg.by, wg.bz);
#endif
}

}  // namespace FVT_NS

template <int MORD>
static int fvt_launch_mode(int dmode, int epi, dim3 grid, hipStream_t st, const Geo& g, const FvMet& m, const real* q, const real* crx,
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
  if (epi == 3 && dmode == 0) FVT_GO(0, 3);  // updatedzd: transport + damping of the heights -> the height update
#undef FVT_GO
  return PACE_ERR_UNSUPPORTED;
}

#endif  // FVT_AVAILABLE

bool FVT_FN(covers)(const Geo& g, int hord) {
#if FVT_AVAILABLE
  return (hord == 5 || hord == 6) && g.n % TI == 0 && g.n % TJ == 0 && g.n >= 2 * TI && g.n >= 2 * TJ && (g.sj & 1) == 0 && (g.sk & 1) == 0;
#else
  (void)g, (void)hord;
  return false;
#endif
}

// Same contract as launch_transport (k_fvtp2d.hip), for the calls this kernel covers; PACE_ERR_UNSUPPORTED otherwise (the
// caller then takes the general kernel).  xmf / ymf resolved by the caller: xu / yu are the unit fluxes.
int FVT_FN(launch_transport)(const Geo& g, const Met& m, const real* q, const real* crx, const real* cry, const real* xfx,
                             const real* yfx, real* fx, real* fy, const real* xu, const real* yu, int hord, int nlev, int dmode,
                             int epi, const FvDamp& dp, hipStream_t st) {
#if FVT_AVAILABLE
  if (!FVT_FN(covers)(g, hord) || ((uintptr_t)q & 15) != 0) return PACE_ERR_UNSUPPORTED;  // (16-byte rows of the footprint)
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

// The scalar phase of d_sw (delp, w, q_con, pt) in one launch; see fvt_core.h.  kc: the device column block of dsw_prepare
// (NCOL arrays of nk + 1).  outs[4] = delp, pt, w, q_con outputs, distinct from the inputs.  PACE_ERR_UNSUPPORTED if the geometry /
// orders are not covered (the caller then runs the scalars one by one).
// whether the scalar-phase kernel can take the winds as its fifth pass (the 512-thread form)
bool FVT_FN(take_winds)() {
#if FVT_AVAILABLE && FVT_SCALARS_NT == 512
  return true;
#else
  return false;
#endif
}

int FVT_FN(launch_scalars)(const Geo& g, const Met& m, const real* delp, const real* pt, const real* w, const real* q_con,
                           real* const* outs, const real* crx, const real* cry, const real* xfx, const real* yfx, real* mfx,
                           real* mfy, real* dw, real* heat_s, real* diss_est, const real* kc, int hord, int nmax_v, int nmax_w,
                           int nmax_t, double dt, hipStream_t st, const DswWinds* winds) {
#if FVT_AVAILABLE
  if (!FVT_FN(covers)(g, hord) || nmax_v > 2 || nmax_w > 2 || nmax_t > 2) return PACE_ERR_UNSUPPORTED;
  if (winds && !FVT_FN(take_winds)()) return PACE_ERR_UNSUPPORTED;
  const real* ins[4] = {delp, pt, w, q_con};
  for (int n = 0; n < 4; ++n)
    if (((uintptr_t)ins[n] & 15) != 0 || outs[n] == nullptr || outs[n] == ins[n]) return PACE_ERR_UNSUPPORTED;
  const int K = g.nk + 1;
  FvtScalars S{};
  // (the order of dsw_prepare: nord_v, nord_w, nord_t, damp_vt, damp_w, damp_t, d2_divg, d_con, ke_bg, fac_vt, fac_t, fac_vt_c, fac_w_c)
  const real *nord_v = kc, *nord_w = kc + K, *nord_t = kc + 2 * K, *fac_vt = kc + 9 * K, *fac_t = kc + 10 * K, *fac_w = kc + 12 * K;
  // delp, w, q_con, pt
  S.q[0] = delp, S.q[1] = w, S.q[2] = q_con, S.q[3] = pt;
  S.qout[0] = outs[0], S.qout[1] = outs[2], S.qout[2] = outs[3], S.qout[3] = outs[1];
  S.fac[0] = fac_vt, S.fac[1] = fac_w, S.fac[2] = fac_t, S.fac[3] = fac_vt;
  S.nord[0] = nord_v, S.nord[1] = nord_w, S.nord[2] = nord_t, S.nord[3] = nord_v;
  S.nmax[0] = nmax_v, S.nmax[1] = nmax_w, S.nmax[2] = nmax_t, S.nmax[3] = nmax_v;
  S.crx = crx, S.cry = cry, S.xfx = xfx, S.yfx = yfx, S.mfx = mfx, S.mfy = mfy, S.dw = dw, S.heat_s = heat_s,
  S.diss_est = diss_est;
  S.damp_w = kc + 4 * K, S.ke_bg = kc + 8 * K;
  S.dt = dt;
  if (winds) {
    // the relative vorticity: DelnFluxNoSG with nord_v and (damp_vt * da_min_c) ^ (nord_v + 1) (d_sw.py:1187-1195)
    if (((uintptr_t)winds->rel_vort & 15) != 0 || winds->u_out == winds->u || winds->v_out == winds->v) return PACE_ERR_UNSUPPORTED;
    S.winds = 1;
    S.q[4] = winds->rel_vort, S.fac[4] = kc + 11 * K, S.nord[4] = nord_v, S.nmax[4] = nmax_v;
    S.u = winds->u, S.v = winds->v, S.u_out = winds->u_out, S.v_out = winds->v_out, S.ke = winds->ke, S.vort_b = winds->vort_b;
    S.heat_source = winds->heat_source, S.do_skeb = winds->do_skeb, S.d_con = winds->d_con, S.copy_wind_halo = winds->copy_halo;
    S.ke_plus_vort = winds->ke_plus_vort;
    S.damp_vt = kc + 3 * K, S.d_con_k = kc + 7 * K;
    S.fC = m.fC_agrid, S.rdx = m.rdx, S.rdy = m.rdy, S.rsin2 = m.rsin2, S.cosa_s = m.cosa_s;
  }
  const dim3 grid(g.n / TI, g.n / TJ, g.nk);
  if (hord == 5) hipLaunchKernelGGL(k_fvt_scalars<5>, grid, dim3(FVT_SCALARS_NT), 0, st, g, fv_met(m), S);
  else hipLaunchKernelGGL(k_fvt_scalars<6>, grid, dim3(FVT_SCALARS_NT), 0, st, g, fv_met(m), S);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
#else
  return PACE_ERR_UNSUPPORTED;
#endif
}

#if FVT_SHAPE == 32
// ---- what the rest of the library calls: the 32 x 24 tile where it tiles the domain, else the 16 x 24 one (k_fvt16.hip) ----
bool fvt16_covers(const Geo& g, int hord);
bool fvt16_take_winds();
int fvt16_launch_transport(const Geo& g, const Met& m, const real* q, const real* crx, const real* cry, const real* xfx, const real* yfx,
                           real* fx, real* fy, const real* xu, const real* yu, int hord, int nlev, int dmode, int epi, const FvDamp& dp,
                           hipStream_t st);
int fvt16_launch_scalars(const Geo& g, const Met& m, const real* delp, const real* pt, const real* w, const real* q_con, real* const* outs,
                         const real* crx, const real* cry, const real* xfx, const real* yfx, real* mfx, real* mfy, real* dw, real* heat_s,
                         real* diss_est, const real* kc, int hord, int nmax_v, int nmax_w, int nmax_t, double dt, hipStream_t st,
                         const DswWinds* winds);

bool transport_lean_covers(const Geo& g, int hord) { return fvt32_covers(g, hord) || fvt16_covers(g, hord); }
bool dsw_scalars_take_winds() { return fvt32_take_winds(); }  // (both shapes are compiled from the same source with the same form)
int launch_transport_lean(const Geo& g, const Met& m, const real* q, const real* crx, const real* cry, const real* xfx,
                          const real* yfx, real* fx, real* fy, const real* xu, const real* yu, int hord, int nlev, int dmode,
                          int epi, const FvDamp& dp, hipStream_t st) {
  if (fvt32_covers(g, hord)) return fvt32_launch_transport(g, m, q, crx, cry, xfx, yfx, fx, fy, xu, yu, hord, nlev, dmode, epi, dp, st);
  return fvt16_launch_transport(g, m, q, crx, cry, xfx, yfx, fx, fy, xu, yu, hord, nlev, dmode, epi, dp, st);
}
int launch_dsw_scalars_lean(const Geo& g, const Met& m, const real* delp, const real* pt, const real* w, const real* q_con,
                            real* const* outs, const real* crx, const real* cry, const real* xfx, const real* yfx, real* mfx,
                            real* mfy, real* dw, real* heat_s, real* diss_est, const real* kc, int hord, int nmax_v, int nmax_w,
                            int nmax_t, double dt, hipStream_t st, const DswWinds* winds) {
  if (fvt32_covers(g, hord))
    return fvt32_launch_scalars(g, m, delp, pt, w, q_con, outs, crx, cry, xfx, yfx, mfx, mfy, dw, heat_s, diss_est, kc, hord, nmax_v, nmax_w,
                                nmax_t, dt, st, winds);
  return fvt16_launch_scalars(g, m, delp, pt, w, q_con, outs, crx, cry, xfx, yfx, mfx, mfy, dw, heat_s, diss_est, kc, hord, nmax_v, nmax_w,
                              nmax_t, dt, st, winds);
}
#endif
