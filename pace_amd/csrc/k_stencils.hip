// Per-stencil device implementations of gtscript definitions that the reference's Translate tests (and a user of the L3
// boundary, dsl/pace/dsl/stencil.py:395-434) launch as stencils of their own, although the fused kernels of this library never
// do: d_sw.py flux_capacitor :33-60, heat_diss :63-103, apply_fluxes :122-145, the kinetic-energy winds ubke / vbke
// (tests/savepoint/translate/translate_d_sw.py:67-81,118-133 around d_sw.py interpolate_uc_vc_to_cell_corners), and the corner
// fills of stencils/pace/stencils/corners.py: fill_corners_2cells_{x,y}_stencil :170-177, copy_corners_{x,y}_stencil_defn
// :307-425, fill_corners_bgrid_{x,y}_defn :592-712, fill_corners_dgrid_defn :987-1151.  One entry point, pace_stencil, with a
// stencil id; every kernel honours the launch window (origin, domain) the FrozenStencil was built with -- region statements
// write only inside it, as in GT4Py.  Point kernels over the window, i fastest: HBM-bound streaming passes.
#include "common.h"
#include "kernels.h"

namespace {

struct Win {
  int i0, j0, k0, ni, nj, nk;
  __device__ __forceinline__ bool has(int i, int j) const { return i >= i0 && i < i0 + ni && j >= j0 && j < j0 + nj; }
};
#define WIN_IJK(w)                                                    \
  const int i = (w).i0 + (int)(blockIdx.x * 64 + threadIdx.x);        \
  const int j = (w).j0 + (int)(blockIdx.y * 4 + threadIdx.y);         \
  const int k = (w).k0 + (int)blockIdx.z;                             \
  if (i >= (w).i0 + (w).ni || j >= (w).j0 + (w).nj) return;
static inline dim3 win_grid(const Win& w) { return dim3((unsigned)((w.ni + 63) / 64), (unsigned)((w.nj + 3) / 4), (unsigned)w.nk); }

__global__ void k_st_flux_capacitor(Geo g, Win w, real* cx, real* cy, real* xflux, real* yflux, const real* crx, const real* cry,
                                    const real* fx, const real* fy) {
  WIN_IJK(w);
  const long c = IDX3(g, i, j, k);
  cx[c] = cx[c] + crx[c];
  cy[c] = cy[c] + cry[c];
  xflux[c] = xflux[c] + fx[c];
  yflux[c] = yflux[c] + fy[c];
}

__global__ void k_st_heat_diss(Geo g, Met m, Win w, const real* fx2, const real* fy2, const real* wv, real* heat_source, real* diss_est,
                               real* dw, const real* damp_w, const real* ke_bg, double dt) {
  WIN_IJK(w);
  const long c = IDX3(g, i, j, k);
  double hs = 0.0;
  if (damp_w[k] > 1e-5) {
    const double dd8 = ke_bg[k] * fabs(dt);
    const double d = (fx2[c] - fx2[c + 1] + fy2[c] - fy2[c + g.sj]) * m.rarea[IDX2(g, i, j)];
    dw[c] = d;
    hs = dd8 - d * (wv[c] + 0.5 * d);
  }
  heat_source[c] = hs;
  diss_est[c] = hs;
}

__global__ void k_st_apply_fluxes(Geo g, Met m, Win w, real* q, const real* delp, const real* gx, const real* gy) {
  WIN_IJK(w);
  const long c = IDX3(g, i, j, k);
  q[c] = q[c] * delp[c] + (gx[c] - gx[c + 1] + gy[c] - gy[c + g.sj]) * m.rarea[IDX2(g, i, j)];
}

// ub (WHICH 0) / vb (WHICH 1) of interpolate_uc_vc_to_cell_corners (d_sw.py) times dt = 2 dt5; `t` is the contravariant wind
// the edge forms read (ut for ub, vt for vb)
template <int WHICH>
__global__ void k_st_bke(Geo g, Met m, Win w, const real* uc, const real* vc, const real* t, real* out, double dt5) {
  WIN_IJK(w);
  const long c = IDX3(g, i, j, k);
  const long c2 = IDX2(g, i, j);
  const int sj = g.sj;
  const double dt = 2.0 * dt5;
  const double ub_cov = 0.5 * (uc[c - sj] + uc[c]);
  const double vb_cov = 0.5 * (vc[c - 1] + vc[c]);
  const bool jedge = (j == g.js || j == g.je + 1), iedge = (i == g.is || i == g.ie + 1);
  double v;
  if (WHICH == 0) {
    v = (ub_cov - vb_cov * m.cosa[c2]) * m.rsina[c2];
    if (jedge) v = 0.25 * (-t[c - 2 * sj] + 3.0 * (t[c - sj] + t[c]) - t[c + sj]);
    if (iedge) v = 0.5 * (t[c - sj] + t[c]);
  } else {
    v = (vb_cov - ub_cov * m.cosa[c2]) * m.rsina[c2];
    if (iedge) v = 0.25 * (-t[c - 2] + 3.0 * (t[c - 1] + t[c]) - t[c + 1]);
    if (jedge) v = 0.5 * (t[c - 1] + t[c]);
  }
  out[c] = v * dt;
}


// xtp_u_stencil_defn / ytp_v_stencil_defn (tests/savepoint/translate/translate_xtp_u.py:13-23, translate_ytp_v.py): the 1-D
// advection of a D-grid wind component by the (contravariant wind x dt) on the cell corners, as a stencil of its own
template <int AXIS, int MORD>
__global__ void k_st_xtp(Geo g, Met m, Win w, const real* c_dt, const real* u, real* flux) {
  WIN_IJK(w);
  const long c = IDX3(g, i, j, k);
  const long c2 = IDX2(g, i, j);
  const int sj = g.sj;
  const long d = AXIS == 0 ? 1 : sj;
  const int pos = AXIS == 0 ? i : j, s0 = AXIS == 0 ? g.is : g.js, e0 = AXIS == 0 ? g.ie : g.je;
  const int opos = AXIS == 0 ? j : i, os = AXIS == 0 ? g.js : g.is, oe = AXIS == 0 ? g.je : g.ie;
  double q6[6];
#pragma unroll
  for (int t = 0; t < 6; ++t) q6[t] = u[c + (long)(t - 3) * d];
  const double ub = c_dt[c];  // dt = 1 folded in
  const real* rd = AXIS == 0 ? m.rdx : m.rdy;
  const real* sp = (AXIS == 0 ? m.dx : m.dy) + (AXIS == 0 ? (long)j * sj : (long)i);
  const double cfl = (ub > 0.0) ? ub * 1.0 * rd[c2 - d] : ub * 1.0 * rd[c2];
  const bool zo = (opos == os || opos == oe + 1);
  auto zero = [&](int p) { return zo && (p == s0 - 1 || p == s0 || p == e0 || p == e0 + 1); };
  flux[c] = wind_flux6<MORD>(q6, ub, cfl, pos, s0, e0, [=](int p) { return (double)sp[(long)p * d]; }, zero(pos - 1), zero(pos));
}

// moist_pt_last_step (fv3core/pace/fv3core/stencils/moist_cv.py:84-118, nwat = 6): gz = the condensate, pt back to the (virtual)
// temperature with the heating of the last remapping step
__global__ void k_st_moist_pt_last_step(Geo g, Win w, const real* qv, const real* ql, const real* qr, const real* qs, const real* qi,
                                        const real* qg, real* gz, real* pt, const real* pkz, double dtmp, double r_vir) {
  WIN_IJK(w);
  const long c = IDX3(g, i, j, k);
  const double cond = ql[c] + qr[c] + qi[c] + qs[c] + qg[c];
  gz[c] = cond;
  pt[c] = (pt[c] + dtmp * pkz[c]) / ((1.0 + r_vir * qv[c]) * (1.0 - cond));
}

// moist_pkz (moist_cv.py:130-172) and the test-side stencil around moist_pt_func (moist_cv.py:48-70;
// tests/savepoint/translate/translate_moistcvpluspt_2d.py:9-37), nwat = 6: the condensate, the moist heat capacity, cappa, and pkz /
// the moist potential temperature.  Constants: util/pace/util/constants.py.
#define ST_RDGAS 287.05
#define ST_RVGAS 461.50
#define ST_GRAV 9.80665
#define ST_CP_AIR 1004.6
#define ST_CV_AIR (ST_CP_AIR - ST_RDGAS)
#define ST_RDG (-ST_RDGAS / ST_GRAV)
#define ST_CV_VAP (3.0 * ST_RVGAS)
#define ST_C_ICE 1972.0
#define ST_C_LIQ 4.1855e3
template <int PT>  // 0: moist_pkz, 1: moist_pt
__global__ void k_st_moist_cv(Geo g, Win w, const real* qv_, const real* ql_, const real* qr_, const real* qs_, const real* qi_,
                              const real* qg_, real* q_con, real* gz_, real* cvm_, real* pkz, real* pt, real* cappa_, const real* delp,
                              const real* delz, double r_vir) {
  WIN_IJK(w);
  const long c = IDX3(g, i, j, k);
  const double qv = qv_[c];
  const double ql = ql_[c] + qr_[c];
  const double qs = qi_[c] + qs_[c] + qg_[c];
  const double gz = ql + qs;
  const double cvm = (1.0 - (qv + gz)) * ST_CV_AIR + qv * ST_CV_VAP + ql * ST_C_LIQ + qs * ST_C_ICE;
  const double cappa = ST_RDGAS / (ST_RDGAS + cvm / (1.0 + r_vir * qv));
  q_con[c] = gz;
  cappa_[c] = cappa;
  if (PT == 0) {
    gz_[c] = gz;
    cvm_[c] = cvm;
    pkz[c] = exp(cappa * log(ST_RDG * delp[c] / delz[c] * pt[c]));
  } else {
    const double p = pt[c];
    pt[c] = p * exp(cappa / (1.0 - cappa) * log(ST_RDG * delp[c] / delz[c] * p));
  }
}

// ---- corner fills: one thread per destination cell of the four 3 x 3 (A-grid) / (B-grid: see below) corner blocks ----
__device__ __forceinline__ void remap_bgrid(const Geo& g, int dir, int& i, int& j) {  // corners.py:591-712 (oracle/corner_ops.py)
  const bool w_ = i < g.is, e_ = i > g.ie + 1, s_ = j < g.js, n_ = j > g.je + 1;
  if (!((w_ || e_) && (s_ || n_))) return;
  const int a = w_ ? g.is - i : i - g.ie - 1;
  const int b = s_ ? g.js - j : j - g.je - 1;
  if (dir == 0) {
    i = w_ ? g.is - b : g.ie + 1 + b;
    j = s_ ? g.js + a : g.je + 1 - a;
  } else {
    i = w_ ? g.is + b : g.ie + 1 - b;
    j = s_ ? g.js - a : g.je + 1 + a;
  }
}
// GRID 0: A-grid cells (copy_corners_*), 1: B-grid points (fill_corners_bgrid_*); DIR 0 = x, 1 = y
template <int GRID, int DIR>
__global__ void k_st_corner_copy(Geo g, Win w, const real* qin, real* qout) {
  const int t = (int)threadIdx.x;  // 36 destination cells: corner q = t / 9, (a, b) = ((t % 9) / 3, t % 3)
  const int k = w.k0 + (int)blockIdx.x;
  if (t >= 36) return;
  const int q = t / 9, a = (t % 9) / 3, b = t % 3;
  const int hi = GRID == 0 ? g.ie : g.ie + 1, hj = GRID == 0 ? g.je : g.je + 1;
  int i = (q & 1) ? hi + 1 + a : g.is - 1 - a;
  int j = (q & 2) ? hj + 1 + b : g.js - 1 - b;
  if (!w.has(i, j)) return;
  int si = i, sj_ = j;
  if (GRID == 0) {
    if (DIR == 0) remap_agrid_x(g, si, sj_);
    else remap_agrid_y(g, si, sj_);
  } else {
    remap_bgrid(g, DIR, si, sj_);
  }
  qout[IDX3(g, i, j, k)] = qin[IDX3(g, si, sj_, k)];
}

// fill_corners_dgrid_defn (corners.py:987-1151; oracle/corner_ops.py fill_corners_dgrid): x_out / y_out corner values from
// y_in / x_in of the adjacent edge halos, with `mysign` on the SW / NE corners
__global__ void k_st_fill_corners_dgrid(Geo g, Win w, const real* x_in, real* x_out, const real* y_in, real* y_out, double mysign) {
  const int t = (int)threadIdx.x;
  const int k = w.k0 + (int)blockIdx.x;
  if (t >= 72) return;
  const bool isx = t < 36;
  const int u = isx ? t : t - 36;
  const int q = u / 9, a = (u % 9) / 3 + 1, b = u % 3 + 1;  // q: 0 SW, 1 NE, 2 NW, 3 SE
  const int is_ = g.is, ie = g.ie, js = g.js, je = g.je;
  int di, dj, si, sj_;
  double s;
  if (isx) {
    if (q == 0) { di = is_ - a; dj = js - b; s = mysign; si = is_ - b; sj_ = js + a - 1; }
    else if (q == 1) { di = ie + a; dj = je + 1 + b; s = mysign; si = ie + 1 + b; sj_ = je + 1 - a; }
    else if (q == 2) { di = is_ - a; dj = je + 1 + b; s = 1.0; si = is_ - b; sj_ = je + 1 - a; }
    else { di = ie + a; dj = js - b; s = 1.0; si = ie + 1 + b; sj_ = js + a - 1; }
    if (w.has(di, dj)) x_out[IDX3(g, di, dj, k)] = s * y_in[IDX3(g, si, sj_, k)];
  } else {
    if (q == 0) { di = is_ - a; dj = js - b; s = mysign; si = is_ + b - 1; sj_ = js - a; }
    else if (q == 1) { di = ie + 1 + a; dj = je + b; s = mysign; si = ie + 1 - b; sj_ = je + 1 + a; }
    else if (q == 2) { di = is_ - a; dj = je + b; s = 1.0; si = is_ + b - 1; sj_ = je + 1 + a; }
    else { di = ie + 1 + a; dj = js - b; s = 1.0; si = ie + 1 - b; sj_ = js - a; }
    if (w.has(di, dj)) y_out[IDX3(g, di, dj, k)] = s * x_in[IDX3(g, si, sj_, k)];
  }
}

// fill_corners_2cells_{x,y}_stencil (corners.py:130-177 and its y twin): eight cells per level
template <int DIR>
__global__ void k_st_fill_2cells(Geo g, Win w, real* q_out, const real* q_in) {
  const int t = (int)threadIdx.x;
  const int k = w.k0 + (int)blockIdx.x;
  if (t >= 8) return;
  const int q = t >> 1, far = t & 1;  // q: 0 SW, 1 SE, 2 NW, 3 NE; far: the second cell away from the edge
  const int is_ = g.is, ie = g.ie, js = g.js, je = g.je;
  int di, dj, oi, oj;  // destination, offset of the source relative to it
  if (DIR == 0) {
    di = (q & 1) ? ie + 1 + far : is_ - 1 - far;
    dj = (q & 2) ? je + 1 : js - 1;
    oi = far ? ((q & 1) ? -1 : 1) : 0;
    oj = (q & 2) ? -(1 + far) : (1 + far);
  } else {
    di = (q & 1) ? ie + 1 : is_ - 1;
    dj = (q & 2) ? je + 1 + far : js - 1 - far;
    oi = (q & 1) ? -(1 + far) : (1 + far);
    oj = far ? ((q & 2) ? -1 : 1) : 0;
  }
  if (w.has(di, dj)) q_out[IDX3(g, di, dj, k)] = q_in[IDX3(g, di + oi, dj + oj, k)];
}

}  // namespace

int launch_stencil(const Geo& g, const Met& m, int id, void* const* f, int nf, const double* sc, int ns, const int* origin,
                   const int* domain, hipStream_t st) {
  const Win w{origin[0], origin[1], origin[2], domain[0], domain[1], domain[2]};
  if (w.ni < 1 || w.nj < 1 || w.nk < 1 || w.i0 < 0 || w.j0 < 0 || w.k0 < 0 || w.i0 + w.ni > g.ni || w.j0 + w.nj > g.nj ||
      w.k0 + w.nk > g.nk + 1)
    return PACE_ERR_ARG;
  auto F = [&](int n) { return (real*)f[n]; };
  const dim3 blk(64, 4);
  switch (id) {
    case PACE_ST_FLUX_CAPACITOR:
      if (nf != 8) return PACE_ERR_ARG;
      hipLaunchKernelGGL(k_st_flux_capacitor, win_grid(w), blk, 0, st, g, w, F(0), F(1), F(2), F(3), F(4), F(5), F(6), F(7));
      break;
    case PACE_ST_HEAT_DISS:
      if (nf != 8 || ns != 1) return PACE_ERR_ARG;
      // (the reads at i + 1 / j + 1 must exist)
      if (w.i0 + w.ni + 1 > g.ni || w.j0 + w.nj + 1 > g.nj) return PACE_ERR_ARG;
      hipLaunchKernelGGL(k_st_heat_diss, win_grid(w), blk, 0, st, g, m, w, F(0), F(1), F(2), F(3), F(4), F(5), F(6), F(7), sc[0]);
      break;
    case PACE_ST_APPLY_FLUXES:
      if (nf != 4) return PACE_ERR_ARG;
      if (w.i0 + w.ni + 1 > g.ni || w.j0 + w.nj + 1 > g.nj) return PACE_ERR_ARG;
      hipLaunchKernelGGL(k_st_apply_fluxes, win_grid(w), blk, 0, st, g, m, w, F(0), F(1), F(2), F(3));
      break;
    case PACE_ST_UBKE:
    case PACE_ST_VBKE:
      if (nf != 4 || ns != 1) return PACE_ERR_ARG;
      if (w.i0 < 2 || w.j0 < 2 || w.i0 + w.ni + 1 > g.ni || w.j0 + w.nj + 1 > g.nj) return PACE_ERR_ARG;
      if (id == PACE_ST_UBKE) hipLaunchKernelGGL(k_st_bke<0>, win_grid(w), blk, 0, st, g, m, w, F(0), F(1), F(2), F(3), sc[0]);
      else hipLaunchKernelGGL(k_st_bke<1>, win_grid(w), blk, 0, st, g, m, w, F(0), F(1), F(2), F(3), sc[0]);
      break;
    case PACE_ST_COPY_CORNERS_X:
      if (nf != 2) return PACE_ERR_ARG;
      hipLaunchKernelGGL((k_st_corner_copy<0, 0>), dim3((unsigned)w.nk), dim3(64), 0, st, g, w, F(0), F(1));
      break;
    case PACE_ST_COPY_CORNERS_Y:
      if (nf != 2) return PACE_ERR_ARG;
      hipLaunchKernelGGL((k_st_corner_copy<0, 1>), dim3((unsigned)w.nk), dim3(64), 0, st, g, w, F(0), F(1));
      break;
    case PACE_ST_FILL_CORNERS_BGRID_X:
      if (nf != 2) return PACE_ERR_ARG;
      hipLaunchKernelGGL((k_st_corner_copy<1, 0>), dim3((unsigned)w.nk), dim3(64), 0, st, g, w, F(0), F(1));
      break;
    case PACE_ST_FILL_CORNERS_BGRID_Y:
      if (nf != 2) return PACE_ERR_ARG;
      hipLaunchKernelGGL((k_st_corner_copy<1, 1>), dim3((unsigned)w.nk), dim3(64), 0, st, g, w, F(0), F(1));
      break;
    case PACE_ST_FILL_CORNERS_DGRID:
      if (nf != 4 || ns != 1) return PACE_ERR_ARG;
      hipLaunchKernelGGL(k_st_fill_corners_dgrid, dim3((unsigned)w.nk), dim3(128), 0, st, g, w, F(0), F(1), F(2), F(3), sc[0]);
      break;
    case PACE_ST_FILL_CORNERS_2CELLS_X:
      if (nf != 2) return PACE_ERR_ARG;
      hipLaunchKernelGGL(k_st_fill_2cells<0>, dim3((unsigned)w.nk), dim3(64), 0, st, g, w, F(0), F(1));
      break;
    case PACE_ST_FILL_CORNERS_2CELLS_Y:
      if (nf != 2) return PACE_ERR_ARG;
      hipLaunchKernelGGL(k_st_fill_2cells<1>, dim3((unsigned)w.nk), dim3(64), 0, st, g, w, F(0), F(1));
      break;
    case PACE_ST_XTP_U:
    case PACE_ST_YTP_V: {
      if (nf != 3 || ns != 1) return PACE_ERR_ARG;
      const int mord = (int)sc[0];
      if (mord != 5 && mord != 6 && mord != 7) return PACE_ERR_UNSUPPORTED;  // (ord 8: not on this path, as in d_sw)
      const bool x = id == PACE_ST_XTP_U;  // six cells along the axis: pos-3 .. pos+2
      if ((x ? w.i0 : w.j0) < 3 || (x ? w.i0 + w.ni + 2 > g.ni : w.j0 + w.nj + 2 > g.nj)) return PACE_ERR_ARG;
      if (x && mord == 5) hipLaunchKernelGGL((k_st_xtp<0, 5>), win_grid(w), blk, 0, st, g, m, w, F(0), F(1), F(2));
      else if (x) hipLaunchKernelGGL((k_st_xtp<0, 6>), win_grid(w), blk, 0, st, g, m, w, F(0), F(1), F(2));
      else if (mord == 5) hipLaunchKernelGGL((k_st_xtp<1, 5>), win_grid(w), blk, 0, st, g, m, w, F(0), F(1), F(2));
      else hipLaunchKernelGGL((k_st_xtp<1, 6>), win_grid(w), blk, 0, st, g, m, w, F(0), F(1), F(2));
      break;
    }
    case PACE_ST_MOIST_PT_LAST_STEP:
      if (nf != 9 || ns != 2) return PACE_ERR_ARG;
      hipLaunchKernelGGL(k_st_moist_pt_last_step, win_grid(w), blk, 0, st, g, w, F(0), F(1), F(2), F(3), F(4), F(5), F(6), F(7), F(8), sc[0],
                         sc[1]);
      break;
    case PACE_ST_MOIST_PKZ:
      if (nf != 14 || ns != 1) return PACE_ERR_ARG;
      hipLaunchKernelGGL(k_st_moist_cv<0>, win_grid(w), blk, 0, st, g, w, F(0), F(1), F(2), F(3), F(4), F(5), F(6), F(7), F(8), F(9), F(10), F(11),
                         F(12), F(13), sc[0]);
      break;
    case PACE_ST_MOIST_PT:
      if (nf != 11 || ns != 1) return PACE_ERR_ARG;
      // fields: qvapor, qliquid, qrain, qsnow, qice, qgraupel, q_con, pt, cappa, delp, delz
      hipLaunchKernelGGL(k_st_moist_cv<1>, win_grid(w), blk, 0, st, g, w, F(0), F(1), F(2), F(3), F(4), F(5), F(6), nullptr, nullptr, nullptr, F(7),
                         F(8), F(9), F(10), sc[0]);
      break;
    default:
      return PACE_ERR_UNSUPPORTED;
  }
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}
