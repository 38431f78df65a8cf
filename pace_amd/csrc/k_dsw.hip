// D-grid shallow-water step (Fortran d_sw) -- everything in
// fv3core/pace/fv3core/stencils/d_sw.py:33-608 and divergence_damping.py:23-632 / a2b_ord4.py that is
// not transport (k_fvtp2d.hip), del-n damping (k_delnflux.hip) or flux preparation (k_fxadv.hip).
// All kernels are one thread per (i, j, k) point with i fastest (coalesced rows); neighbour values are
// re-read through L1/L2 and multi-stage reference temporaries (u_contra_dyc, v_contra_dxc, uc/vc of the
// divergence iteration, qx/qy of a2b_ord4, ubt/vbt ...) are recomputed in registers instead of being
// stored, so each kernel is a pure streaming pass.  HBM-bound.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "common.h"
#include "kernels.h"

#define DCON_THRESHOLD 1e-5



// the results of the fused scalar phase from their workspace fields back into the caller's fields (compute domain)
__global__ void __launch_bounds__(256)
k_copy_scalars(Geo g, real* __restrict__ delp, real* __restrict__ pt, real* __restrict__ w, real* __restrict__ q_con,
               const real* __restrict__ delp_n, const real* __restrict__ pt_n, const real* __restrict__ w_n,
               const real* __restrict__ q_con_n) {
  PATCH_IJK(g);
  if (i < g.is || i > g.ie || j < g.js || j > g.je) return;
  const long c = IDX3(g, i, j, k);
  delp[c] = delp_n[c];
  pt[c] = pt_n[c];
  w[c] = w_n[c];
  q_con[c] = q_con_n[c];
}

// the final winds from their workspace fields back into the caller's u (faces [is, ie] x [js, je + 1]) and v ([is, ie + 1] x [js, je])
__global__ void __launch_bounds__(256)
k_copy_winds(Geo g, real* __restrict__ u, real* __restrict__ v, const real* __restrict__ u_n, const real* __restrict__ v_n) {
  PATCH_IJK(g);
  if (i < g.is || i > g.ie + 1 || j < g.js || j > g.je + 1) return;
  const long c = IDX3(g, i, j, k);
  if (i <= g.ie) u[c] = u_n[c];
  if (j <= g.je) v[c] = v_n[c];
}

// What the reference LEAVES in the 3 x 3 corner blocks of the halo of a transported scalar (full contract only):
// FiniteVolumeTransport copies the corners in y, then in x, IN PLACE on q (fvtp2d.py:262-345; corners.py:307-425), the del-n
// damping works on a copy, and nothing of d_sw writes the blocks afterwards -- so delp, pt, w, q_con end with copy_corners_x of
// what they held, which TranslateD_SW compares (translate_d_sw.py:36-65: the whole storage).  The kernels apply the corner
// copies as index maps on reads; this writes the blocks once, at the end: every source lies in an edge halo, which d_sw never
// writes.  One workgroup per level, 4 fields x 4 corners x 9 cells.
__global__ void __launch_bounds__(256)
k_corner_blocks_x(Geo g, real* __restrict__ f0, real* __restrict__ f1, real* __restrict__ f2, real* __restrict__ f3) {
  const int t = (int)threadIdx.x;
  if (t >= 144) return;
  const int f = t / 36, e = t - f * 36, q = e / 9, a = (e % 9) % 3, b = (e % 9) / 3;
  real* const p = (f == 0 ? f0 : f == 1 ? f1 : f == 2 ? f2 : f3) + (long)blockIdx.x * g.sk;
  const int i = (q & 1) ? g.ie + 1 + a : g.is - 1 - a, j = (q & 2) ? g.je + 1 + b : g.js - 1 - b;
  int si = i, sj_ = j;
  remap_agrid_x(g, si, sj_);
  p[IDX2(g, i, j)] = p[IDX2(g, si, sj_)];
}

bool dsw_pingpong_supported(const Geo& g, const pace_dsw_config_t* cfg) {
  return cfg->hord_dp == cfg->hord_vt && cfg->hord_dp == cfg->hord_tm && transport_lean_covers(g, cfg->hord_dp);
}
// whether a whole-d_sw call runs the winds as the fifth pass of the scalar-phase kernel (then nothing of d_sw is left to overlap
// with what follows it, and the winds can have outputs of their own)
bool dsw_winds_in_scalars(const Geo& g, const pace_dsw_config_t* cfg) {
  return dsw_pingpong_supported(g, cfg) && dsw_scalars_take_winds() && getenv("PACE_DSW_SEPARATE_WINDS") == nullptr;
}

// apply_pt_delp_fluxes (d_sw.py:148-201) + adjust_w_and_qcon (:331-350), given the flux-form updates
// pt*delp + F(pt), w*delp + F(w), q_con*delp + F(q_con) that the transport kernels' epilogues produced
__global__ void __launch_bounds__(256)
k_finish_scalars(Geo g, Met m, real* __restrict__ pt, real* __restrict__ delp, real* __restrict__ w,
                 real* __restrict__ q_con, const real* __restrict__ ptn, const real* __restrict__ wn,
                 const real* __restrict__ qn, const real* __restrict__ fx, const real* __restrict__ fy,
                 const real* __restrict__ dw, const real* __restrict__ damp_w) {
  PATCH_IJK(g);
  if (i < g.is || i > g.ie || j < g.js || j > g.je) return;
  const long c = IDX3(g, i, j, k);
  const double dn = delp[c] + (fx[c] - fx[c + 1] + fy[c] - fy[c + g.sj]) * m.rarea[IDX2(g, i, j)];
  pt[c] = ptn[c] / dn;
  delp[c] = dn;
  double wv = wn[c] / dn;
  if (damp_w[k] > 1e-5) wv = wv + dw[c];
  w[c] = wv;
  q_con[c] = qn[c] / dn;
}

// ------------------------------------------------------------------------------------------------
// compute_kinetic_energy (d_sw.py:204-298) with xtp_u / ytp_v (xtp_u.py:9-91, ytp_v.py:9-91), ord < 8
// ------------------------------------------------------------------------------------------------
// (wind_flux6 -- the 1-D advection of a wind component, xtp_u.py:9-91 -- lives in common.h: csrc/k_stencils.hip launches it as a
// stencil of its own)
// compute_vorticity (d_sw.py:301-328) at cell (i, j): the relative vorticity; the absolute one (+ fC_agrid, rel_vorticity_to_abs
// :389-402) is formed where it is transported (launch_d_sw)
__device__ __forceinline__ double rel_vorticity(const Geo& g, const Met& m, const real* __restrict__ u, const real* __restrict__ v,
                                                long c, long c2) {
  const double ra = m.rarea[c2], dx = m.dx[c2], dy = m.dy[c2];
  return (u[c] - u[c + g.sj] * m.dx[c2 + g.sj] / dx) * (ra * dx) + (v[c + 1] * m.dy[c2 + 1] / dy - v[c]) * (ra * dy);
}

// ... with the relative vorticity of the same winds (round 5: it was a kernel of its own that read u and v once more): regions
// [0, nke) are the B-grid points of the kinetic energy (each also a cell of the vorticity), the regions after them the rest of
// the vorticity's domain (the halo cells around the B-grid domain), where only the vorticity is formed.  vort == nullptr: none.
template <int MORD>
__device__ __forceinline__ void
kinetic_energy_point(const Geo& g, const Met& m, const real* __restrict__ uc, const real* __restrict__ vc, const real* __restrict__ u,
                     const real* __restrict__ v, const real* __restrict__ ut, const real* __restrict__ vt, real* __restrict__ ke, double dt,
                     real* __restrict__ vort, int i, int j, int k, bool interior, bool ke_region) {
  const long c = IDX3(g, i, j, k);
  const long c2 = IDX2(g, i, j);
  const int sj = g.sj;
  if (vort != nullptr) vort[c] = rel_vorticity(g, m, u, v, c, c2);  // (every point of every region is a cell of its domain)
  if (!ke_region) return;
  if (interior) {
    // is+3 <= i <= ie-2 and the same in j: no edge wind, no one-sided PPM interface, no zeroed reconstruction
    const double ub_cov = 0.5 * (uc[c - sj] + uc[c]);
    const double vb_cov = 0.5 * (vc[c - 1] + vc[c]);
    const double ub = (ub_cov - vb_cov * m.cosa[c2]) * m.rsina[c2];
    const double vb = (vb_cov - ub_cov * m.cosa[c2]) * m.rsina[c2];
    double q6[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) q6[t] = v[c + (long)(t - 3) * sj];
    double cfl = (vb > 0.0) ? vb * dt * m.rdy[c2 - sj] : vb * dt * m.rdy[c2];
    const double adv_v = wind_flux6<MORD, false>(q6, vb, cfl, j, g.js, g.je, [](int) { return 0.0; }, false, false);
#pragma unroll
    for (int t = 0; t < 6; ++t) q6[t] = u[c + (t - 3)];
    cfl = (ub > 0.0) ? ub * dt * m.rdx[c2 - 1] : ub * dt * m.rdx[c2];
    const double adv_u = wind_flux6<MORD, false>(q6, ub, cfl, i, g.is, g.ie, [](int) { return 0.0; }, false, false);
    ke[c] = 0.5 * dt * (ub * adv_u + vb * adv_v);
    return;
  }
  const bool jedge = (j == g.js || j == g.je + 1), iedge = (i == g.is || i == g.ie + 1);
  double kev;
  if (iedge && jedge) {
    // all_corners_ke / corner_ke (d_sw.py:259-298)
    int io1, jo1, io2;
    double vs;
    if (i == g.is && j == g.js) { io1 = 0; jo1 = 0; io2 = -1; vs = 1.0; }
    else if (i == g.ie + 1 && j == g.js) { io1 = -1; jo1 = 0; io2 = 0; vs = -1.0; }
    else if (i == g.ie + 1 && j == g.je + 1) { io1 = -1; jo1 = -1; io2 = 0; vs = 1.0; }
    else { io1 = 0; jo1 = -1; io2 = -1; vs = -1.0; }
    const double dt6 = dt / 6.0;
    const double ut0 = ut[c], utm = ut[c - sj], vt0 = vt[c], vtm = vt[c - 1];
    const double u0 = u[c], um = u[c - 1], v0 = v[c], vm = v[c - sj];
    kev = dt6 * ((ut0 + utm) * ((io1 + 1) * u0 - (io1 * um)) + (vt0 + vtm) * ((jo1 + 1) * v0 - (jo1 * vm)) +
                 (((jo1 + 1) * ut0 - (jo1 * utm)) + vs * ((io1 + 1) * vt0 - (io1 * vtm))) *
                     ((io2 + 1) * u0 - (io2 * um)));
  } else {
    const double ub_cov = 0.5 * (uc[c - sj] + uc[c]);
    const double vb_cov = 0.5 * (vc[c - 1] + vc[c]);
    double ub = (ub_cov - vb_cov * m.cosa[c2]) * m.rsina[c2];
    double vb = (vb_cov - ub_cov * m.cosa[c2]) * m.rsina[c2];
    if (jedge) ub = 0.25 * (-ut[c - 2 * sj] + 3.0 * (ut[c - sj] + ut[c]) - ut[c + sj]);
    if (iedge) ub = 0.5 * (ut[c - sj] + ut[c]);
    if (iedge) vb = 0.25 * (-vt[c - 2] + 3.0 * (vt[c - 1] + vt[c]) - vt[c + 1]);
    if (jedge) vb = 0.5 * (vt[c - 1] + vt[c]);
    double q6[6];
    // advect_v_along_y: cells (i, j-3 .. j+2); spacing passed to compute_al is dy (ytp_v.py:22)
#pragma unroll
    for (int t = 0; t < 6; ++t) q6[t] = v[c + (long)(t - 3) * sj];
    const real* dy = m.dy;
    const long col = i;
    const bool zi = (i == g.is || i == g.ie + 1);
    auto zrow = [&](int jj) { return zi && (jj == g.js - 1 || jj == g.js || jj == g.je || jj == g.je + 1); };
    double cfl = (vb > 0.0) ? vb * dt * m.rdy[c2 - sj] : vb * dt * m.rdy[c2];
    const double adv_v = wind_flux6<MORD>(q6, vb, cfl, j, g.js, g.je,
                                          [=](int p) { return dy[col + (long)p * sj]; }, zrow(j - 1), zrow(j));
#pragma unroll
    for (int t = 0; t < 6; ++t) q6[t] = u[c + (t - 3)];
    const real* dx = m.dx + (long)j * sj;
    const bool zj = (j == g.js || j == g.je + 1);
    auto zcol = [&](int ii) { return zj && (ii == g.is - 1 || ii == g.is || ii == g.ie || ii == g.ie + 1); };
    cfl = (ub > 0.0) ? ub * dt * m.rdx[c2 - 1] : ub * dt * m.rdx[c2];
    const double adv_u = wind_flux6<MORD>(q6, ub, cfl, i, g.is, g.ie, [=](int p) { return dx[p]; }, zcol(i - 1), zcol(i));
    kev = 0.5 * dt * (ub * adv_u + vb * adv_v);
  }
  ke[c] = kev;
}

template <int MORD>
__global__ void __launch_bounds__(256)
k_kinetic_energy(Geo g, Met m, const real* __restrict__ uc, const real* __restrict__ vc,
                 const real* __restrict__ u, const real* __restrict__ v, const real* __restrict__ ut,
                 const real* __restrict__ vt, real* __restrict__ ke, double dt, Regions R, real* __restrict__ vort, int nke) {
  REGION_POINT_XCD(R);  // (six rows of v per point: the j-neighbouring patches share an L2 -- 255 -> 156 MB, round 3's x17)
  kinetic_energy_point<MORD>(g, m, uc, vc, u, v, ut, vt, ke, dt, vort, i, j, k, interior, reg__ < nke);
}

// The kinetic energy and the relative vorticity in ONE launch, as two kinds of workgroups (round 6): a level's kinetic-energy blocks and,
// behind them in the same level's share of the launch, the 64 x 4 patches of its vorticity.  Nothing is shared per point (x02: the
// vorticity inside the kinetic-energy point function was slower); what is shared is the launch -- no drain / fill between the two -- and
// the XCD: a level's blocks of both kinds run on one XCD, where the vorticity finds the rows of u and v the kinetic energy just read.
// KE_CH levels per thread: 2 where the launch still has many workgroups per compute unit (the metric values of a point are loaded once
// for both levels: -7 us at C192), 1 on small tiles (C48: 12 against 15 us); 4 is no better than 1 at C192 (experiment x40).
#ifndef KE_CH2_MIN_WGS
#define KE_CH2_MIN_WGS 16384u  // workgroups the one-level launch must have for two levels per thread to pay (C192: 27 k; C96: 7.9 k)
#endif
template <int MORD, int KE_CH>
__global__ void __launch_bounds__(256)
k_ke_vorticity(Geo g, Met m, const real* __restrict__ uc, const real* __restrict__ vc, const real* __restrict__ u,
               const real* __restrict__ v, const real* __restrict__ ut, const real* __restrict__ vt, real* __restrict__ ke, double dt,
               Regions R, real* __restrict__ vort, int nbr) {
  int bx__, bz__;
#ifdef PACE_EMU
  bx__ = (int)blockIdx.x, bz__ = (int)blockIdx.z;
#else
  {  // (REGION_POINT_XCD's map: XCD x works through levels x, x + 8, ...)
    const int nbx__ = (int)gridDim.x, nlev__ = (int)gridDim.z;
    const int lin__ = (int)blockIdx.x + nbx__ * (int)blockIdx.z;
    const int full__ = (nlev__ / 8) * 8;
    if (lin__ < full__ * nbx__) {
      const int slot__ = lin__ >> 3;
      bz__ = (slot__ / nbx__) * 8 + (lin__ & 7);
      bx__ = slot__ - (slot__ / nbx__) * nbx__;
    } else {
      bz__ = lin__ / nbx__;
      bx__ = lin__ - bz__ * nbx__;
    }
  }
#endif
  const int k0 = bz__ * KE_CH;
  if (bx__ >= nbr) {
    // a patch of the vorticity (k_vorticity's enumeration)
    const int pb = bx__ - nbr, npx = (g.ni + PATCH_W - 1) / PATCH_W;
    const int i = (pb % npx) * PATCH_W + (int)threadIdx.x, j = (pb / npx) * PATCH_H + (int)threadIdx.y;
    if (i > g.ni - 2 || j > g.nj - 2) return;
#pragma unroll
    for (int t = 0; t < KE_CH; ++t) {
      const int k = k0 + t;
      if (k >= g.nk) break;
      const long c = IDX3(g, i, j, k);
      vort[c] = rel_vorticity(g, m, u, v, c, IDX2(g, i, j));
    }
    return;
  }
  int reg__ = 0;
  while (reg__ + 1 < R.n && bx__ >= R.first[reg__ + 1]) ++reg__;
  const int b__ = bx__ - R.first[reg__];
  const bool interior = reg__ < R.nplain;
  int i, j;
  if (reg__ == 0) {
    i = R.ib[0] + (b__ % R.nbx0) * 64 + (int)threadIdx.x;
    j = R.jb[0] + (b__ / R.nbx0) * 4 + (int)threadIdx.y;
    if (i > R.ie[0] || j > R.je[0]) return;
  } else {
    const int w__ = R.ie[reg__] - R.ib[reg__] + 1;
    const int p__ = b__ * 256 + (int)threadIdx.y * 64 + (int)threadIdx.x;
    j = R.jb[reg__] + p__ / w__;
    i = R.ib[reg__] + p__ % w__;
    if (j > R.je[reg__]) return;
  }
#pragma unroll
  for (int t = 0; t < KE_CH; ++t) {
    const int k = k0 + t;
    if (k >= g.nk) break;
    kinetic_energy_point<MORD>(g, m, uc, vc, u, v, ut, vt, ke, dt, (real*)nullptr, i, j, k, interior, true);
  }
}

// Separate outputs of the winds (which the caller swaps in): the output buffers get the halo the inputs have -- the storage outside
// the faces d_sw writes, u: [is, ie] x [js, je + 1], v: [is, ie + 1] x [js, je].  The frame of the plane as four strips, one thread
// per point: south rows [0, js), north rows (je, nj), and between them the west columns [0, is) and the east columns (ie, ni).
// (Inside the scalar-phase kernel the copy cost an edge tile 4.5 - 8 k cycles; riding on k_vorticity it cost that kernel 7 us.)
__global__ void __launch_bounds__(256)
k_copy_wind_halo(Geo g, const real* __restrict__ u, const real* __restrict__ v, real* __restrict__ u_out, real* __restrict__ v_out) {
  const int p = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (p >= wind_halo_points(g)) return;
  wind_halo_copy_point(g, p, (int)blockIdx.y, u, v, u_out, v_out);
}

// compute_vorticity (d_sw.py:301-328) + rel_vorticity_to_abs (:389-402), compute domain + halo 3
__global__ void __launch_bounds__(256)
k_vorticity(Geo g, Met m, const real* __restrict__ u, const real* __restrict__ v, real* __restrict__ vort) {
  PATCH_IJK(g);
  if (i > g.ni - 2 || j > g.nj - 2) return;
  const long c = IDX3(g, i, j, k);
  const long c2 = IDX2(g, i, j);
  vort[c] = rel_vorticity(g, m, u, v, c, c2);
}

// ------------------------------------------------------------------------------------------------
// DivergenceDamping, sponge levels (nord_col == 0): divergence_damping.py:30-158
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double dd_u_contra_dyc(const Geo& g, const Met& m, const real* u, const real* va,
                                                  const real* vc, long c, long c2, int j) {
  double uc_;
  if (j == g.js || j == g.je + 1) {
    uc_ = (vc[c] > 0.0) ? u[c] * m.sin_sg4[c2 - g.sj] : u[c] * m.sin_sg2[c2];
  } else {
    const double vfa = 0.5 * (va[c - g.sj] + va[c]);
    uc_ = (u[c] - vfa * m.cosa_v[c2]) * m.sina_v[c2];
  }
  return uc_ * m.dyc[c2];
}

__device__ __forceinline__ double dd_v_contra_dxc(const Geo& g, const Met& m, const real* v, const real* ua,
                                                  const real* uc, long c, long c2, int i) {
  double vc_;
  if (i == g.is || i == g.ie + 1) {
    vc_ = (uc[c] > 0.0) ? v[c] * m.sin_sg3[c2 - 1] : v[c] * m.sin_sg1[c2];
  } else {
    const double ufa = 0.5 * (ua[c - 1] + ua[c]);
    vc_ = (v[c] - ufa * m.cosa_u[c2]) * m.sina_u[c2];
  }
  return vc_ * m.dxc[c2];
}

// one B-grid point of the sponge levels: delpc, the damped vorticity and ke += it (divergence_damping.py:30-158)
__device__ __forceinline__ void divdamp_low_point(const Geo& g, const Met& m, const real* __restrict__ u, const real* __restrict__ v,
                                                  const real* __restrict__ ua, const real* __restrict__ va, const real* __restrict__ uc,
                                                  const real* __restrict__ vc, real* __restrict__ delpc, real* __restrict__ vort_b,
                                                  real* __restrict__ ke, double d2, double dddmp, double dt, int i, int j, int k) {
  const long c = IDX3(g, i, j, k);
  const long c2 = IDX2(g, i, j);
  const int sj = g.sj;
  // a = u_contra_dyc, b = v_contra_dxc (argument order at divergence_damping.py:561-566)
  const double a0 = dd_u_contra_dyc(g, m, u, va, vc, c, c2, j);
  const double am = dd_u_contra_dyc(g, m, u, va, vc, c - 1, c2 - 1, j);
  const double b0 = dd_v_contra_dxc(g, m, v, ua, uc, c, c2, i);
  const double bm = dd_v_contra_dxc(g, m, v, ua, uc, c - sj, c2 - sj, i);
  double d = bm - b0 + am - a0;
  const bool ic = (i == g.is || i == g.ie + 1);
  if (ic && j == g.js) d = d - bm;
  if (ic && j == g.je + 1) d = d + b0;
  d = m.rarea_c[c2] * d;
  delpc[c] = d;
  const double delpcdt = d * dt;
  const double damp = m.da_min_c * fmax(d2, fmin(0.2, dddmp * fabs(delpcdt)));
  const double vort = damp * d;
  vort_b[c] = vort;
  if (ke != nullptr) ke[c] = ke[c] + vort;  // (nullptr: the consumer adds the damped vorticity to the kinetic energy itself)
}

__global__ void __launch_bounds__(256)
k_divdamp_low(Geo g, Met m, const real* __restrict__ u, const real* __restrict__ v,
              const real* __restrict__ ua, const real* __restrict__ va, const real* __restrict__ uc,
              const real* __restrict__ vc, real* __restrict__ delpc, real* __restrict__ vort_b,
              real* __restrict__ ke, const real* __restrict__ d2_bg, double dddmp, double dt) {
  PLANE_IJK(g);
  if (i < g.is || i > g.ie + 1 || j < g.js || j > g.je + 1) return;
  const long c = IDX3(g, i, j, k);
  const long c2 = IDX2(g, i, j);
  const int sj = g.sj;
  // a = u_contra_dyc, b = v_contra_dxc (argument order at divergence_damping.py:561-566)
  const double a0 = dd_u_contra_dyc(g, m, u, va, vc, c, c2, j);
  const double am = dd_u_contra_dyc(g, m, u, va, vc, c - 1, c2 - 1, j);
  const double b0 = dd_v_contra_dxc(g, m, v, ua, uc, c, c2, i);
  const double bm = dd_v_contra_dxc(g, m, v, ua, uc, c - sj, c2 - sj, i);
  double d = bm - b0 + am - a0;
  const bool ic = (i == g.is || i == g.ie + 1);
  if (ic && j == g.js) d = d - bm;
  if (ic && j == g.je + 1) d = d + b0;
  d = m.rarea_c[c2] * d;
  delpc[c] = d;
  const double delpcdt = d * dt;
  const double damp = m.da_min_c * fmax(d2_bg[k], fmin(0.2, dddmp * fabs(delpcdt)));
  const double vort = damp * d;
  vort_b[c] = vort;
  ke[c] = ke[c] + vort;
}

// ------------------------------------------------------------------------------------------------
// DivergenceDamping, nord > 0 levels: one pass of
//   fill_corners_bgrid_x -> vc_from_divg -> fill_corners_bgrid_y -> uc_from_divg -> fill_corners_dgrid
//   -> redo_divg_d                                   (divergence_damping.py:579-600)
// as a 5-point update of the divergence itself; uc/vc are recomputed, never stored (they are dead
// after d_sw: d_sw.py:1032-1033, dyn_core.py:851-852).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void remap_bgrid_x(const Geo& g, int& i, int& j) {
  // corners.py:591-650; closed form in oracle/corner_ops.py.  B-grid points: tile spans is .. ie+1.
  const bool w = i < g.is, e = i > g.ie + 1, s = j < g.js, n = j > g.je + 1;
  if (!((w || e) && (s || n))) return;
  const int a = w ? g.is - i : i - g.ie - 1;
  const int b = s ? g.js - j : j - g.je - 1;
  if (a > 3 || b > 3) return;
  i = w ? g.is - b : g.ie + 1 + b;
  j = s ? g.js + a : g.je + 1 - a;
}
__device__ __forceinline__ void remap_bgrid_y(const Geo& g, int& i, int& j) {
  const bool w = i < g.is, e = i > g.ie + 1, s = j < g.js, n = j > g.je + 1;
  if (!((w || e) && (s || n))) return;
  const int a = w ? g.is - i : i - g.ie - 1;
  const int b = s ? g.js - j : j - g.je - 1;
  if (a > 3 || b > 3) return;
  i = w ? g.is + b : g.ie + 1 - b;
  j = s ? g.js - a : g.je + 1 + a;
}

// accessors of a 2-D plane by global (i, j): a level of a field in memory, or a tile's footprint staged in LDS
struct PlaneInMemory {
  const real* p;  // level base applied
  int sj;
  __device__ __forceinline__ double operator()(int i, int j) const { return p[(long)j * sj + i]; }
};
struct PlaneInLds {
  const double* s;
  int ilo, jlo, pitch;
  __device__ __forceinline__ double operator()(int i, int j) const { return s[(j - jlo) * pitch + (i - ilo)]; }
};

template <class Plane>
struct DivIterT {
  const Geo& g;
  const Met& m;
  Plane d;  // divergence of the previous iterate
  bool fill;
  __device__ __forceinline__ double dx_(int i, int j) const {  // divg with corners filled in x
    if (fill) remap_bgrid_x(g, i, j);
    return d(i, j);
  }
  __device__ __forceinline__ double dy_(int i, int j) const {
    if (fill) remap_bgrid_y(g, i, j);
    return d(i, j);
  }
  __device__ __forceinline__ double vc_raw(int i, int j) const {  // vc_from_divg :188-197
    return (dx_(i + 1, j) - dx_(i, j)) * m.divg_u[IDX2(g, i, j)];
  }
  __device__ __forceinline__ double uc_raw(int i, int j) const {  // uc_from_divg :200-209
    return (dy_(i, j + 1) - dy_(i, j)) * m.divg_v[IDX2(g, i, j)];
  }
  // fill_corners_dgrid(vc as x, uc as y, mysign = -1) (corners.py:987-1151)
  __device__ __forceinline__ double vc(int i, int j) const {
    if (fill) {
      if (i < g.is && j < g.js) { int a = g.is - i, b = g.js - j; return -uc_raw(g.is - b, g.js + a - 1); }
      if (i > g.ie && j > g.je + 1) { int a = i - g.ie, b = j - g.je - 1; return -uc_raw(g.ie + 1 + b, g.je + 1 - a); }
      if (i < g.is && j > g.je + 1) { int a = g.is - i, b = j - g.je - 1; return uc_raw(g.is - b, g.je + 1 - a); }
      if (i > g.ie && j < g.js) { int a = i - g.ie, b = g.js - j; return uc_raw(g.ie + 1 + b, g.js + a - 1); }
    }
    return vc_raw(i, j);
  }
  __device__ __forceinline__ double uc(int i, int j) const {
    if (fill) {
      if (i < g.is && j < g.js) { int a = g.is - i, b = g.js - j; return -vc_raw(g.is + b - 1, g.js - a); }
      if (i > g.ie + 1 && j > g.je) { int a = i - g.ie - 1, b = j - g.je; return -vc_raw(g.ie + 1 - b, g.je + 1 + a); }
      if (i < g.is && j > g.je) { int a = g.is - i, b = j - g.je; return vc_raw(g.is + b - 1, g.je + 1 + a); }
      if (i > g.ie + 1 && j < g.js) { int a = i - g.ie - 1, b = g.js - j; return vc_raw(g.ie + 1 - b, g.js - a); }
    }
    return uc_raw(i, j);
  }
};

__global__ void __launch_bounds__(256)
k_divdamp_iter(Geo g, Met m, const real* __restrict__ din, real* __restrict__ dout, int k0, int fill, int adjust,
               real* __restrict__ uc_out, real* __restrict__ vc_out, Regions R) {
  REGION_POINT(R);
  const int kk = k + k0;
  const long c2 = IDX2(g, i, j);
  const long c = c2 + (long)kk * g.sk;
  double d, uc_here, vc_here;
  if (interior) {
    // columns is+1 .. ie: no operand lies in a corner region and no corner adjustment applies
    const int sj = g.sj;
    const double d0 = din[c];
    const double ucm = (d0 - din[c - sj]) * m.divg_v[c2 - sj];
    const double uc0 = (din[c + sj] - d0) * m.divg_v[c2];
    const double vcm = (d0 - din[c - 1]) * m.divg_u[c2 - 1];
    const double vc0 = (din[c + 1] - d0) * m.divg_u[c2];
    d = ucm - uc0 + vcm - vc0;
    uc_here = uc0;
    vc_here = vc0;
  } else {
    DivIterT<PlaneInMemory> it{g, m, PlaneInMemory{din + (long)kk * g.sk, g.sj}, fill != 0};
    const double ucm = it.uc(i, j - 1), uc0 = it.uc(i, j), vcm = it.vc(i - 1, j), vc0 = it.vc(i, j);
    d = ucm - uc0 + vcm - vc0;  // redo_divg_d :212-240
    const bool ic = (i == g.is || i == g.ie + 1);
    if (ic && j == g.js) d = d - ucm;
    if (ic && j == g.je + 1) d = d + uc0;
    uc_here = uc0;
    vc_here = vc0;
  }
  if (adjust) d = d * m.rarea_c[c2];
  dout[c] = d;
  // The reference uses the caller's uc / vc as the work fields of this iteration (uc_from_divg / vc_from_divg,
  // divergence_damping.py:188-209) and its Translate tests compare what is left in them after d_sw on the staggered
  // compute windows (translate_d_sw.py:36-65): the values of the LAST iteration.
  if (uc_out != nullptr) {
    if (j <= g.je) uc_out[c] = uc_here;
    if (i <= g.ie) vc_out[c] = vc_here;
  }
}

static void launch_divdamp_iter(const Geo& g, const Met& m, const real* din, real* dout, int k0, int nlev, int nt, int fill,
                                real* uc_out, real* vc_out, hipStream_t st) {
  const int jb = g.js - nt, je_ = g.je + nt + 1;
  Regions r{};
  add_region(r, g.is + 1, g.ie, jb, je_);
  add_region(r, g.is - nt, g.is, jb, je_);
  add_region(r, g.ie + 1, g.ie + nt + 1, jb, je_);
  hipLaunchKernelGGL(k_divdamp_iter, regions_grid(r, nlev), dim3(64, 4), 0, st, g, m, din, dout, k0, fill, 1, uc_out, vc_out, r);
}

// ------------------------------------------------------------------------------------------------
// a2b_ord4 (a2b_ord4.py:59-506) as a point function, + the tail of DivergenceDamping.__call__
// (smagorinsky_diffusion_approx :243-251, damping_nord_highorder_stencil :161-185)
// ------------------------------------------------------------------------------------------------
template <class Plane>
__device__ __forceinline__ double a2b_interior_point(const Plane& Q, int i, int j) {
  // is+2 <= i <= ie-1 and js+2 <= j <= je-1: every qx / qy involved is the plain 4-point mean (a2b_ord4.py:329-506)
  const double a1 = 9.0 / 16.0, a2 = -1.0 / 16.0, b1 = 7.0 / 12.0, b2 = -1.0 / 12.0;
  double v[4][4];  // v[b][a] = Q(i-2+a, j-2+b)
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int a = 0; a < 4; ++a) v[b][a] = Q(i - 2 + a, j - 2 + b);
  double qx_[4], qy_[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) qx_[b] = b2 * (v[b][0] + v[b][3]) + b1 * (v[b][1] + v[b][2]);  // qx(i, j-2+b)
#pragma unroll
  for (int a = 0; a < 4; ++a) qy_[a] = b2 * (v[0][a] + v[3][a]) + b1 * (v[1][a] + v[2][a]);  // qy(i-2+a, j)
  const double qxx = a2 * (qx_[0] + qx_[3]) + a1 * (qx_[1] + qx_[2]);
  const double qyy = a2 * (qy_[0] + qy_[3]) + a1 * (qy_[1] + qy_[2]);
  return 0.5 * (qxx + qyy);
}

// where a2b_ord4's general forms take the field and the cell widths from: memory, or planes of a tile's footprint staged in LDS
// (the edge forms are chains of a dozen dependent reads: from memory, one round trip each)
struct A2BInMemory {
  const real* q;  // level base applied
  __device__ __forceinline__ double Q(const Geo& g, const Met&, int i, int j) const { return q[IDX2(g, i, j)]; }
  __device__ __forceinline__ double DXA(const Geo& g, const Met& m, int i, int j) const { return m.dxa[IDX2(g, i, j)]; }
  __device__ __forceinline__ double DYA(const Geo& g, const Met& m, int i, int j) const { return m.dya[IDX2(g, i, j)]; }
};
struct A2BInLds {
  const double *q, *xa, *ya;
  int ilo, jlo, pitch;
  __device__ __forceinline__ double Q(const Geo&, const Met&, int i, int j) const { return q[(j - jlo) * pitch + (i - ilo)]; }
  __device__ __forceinline__ double DXA(const Geo&, const Met&, int i, int j) const { return xa[(j - jlo) * pitch + (i - ilo)]; }
  __device__ __forceinline__ double DYA(const Geo&, const Met&, int i, int j) const { return ya[(j - jlo) * pitch + (i - ilo)]; }
};
template <class Src>
struct A2BT {
  const Geo& g;
  const Met& m;
  Src src;
  __device__ __forceinline__ double Q(int i, int j) const { return src.Q(g, m, i, j); }
  __device__ __forceinline__ double DXA(int i, int j) const { return src.DXA(g, m, i, j); }
  __device__ __forceinline__ double DYA(int i, int j) const { return src.DYA(g, m, i, j); }
  __device__ double qx(int i, int j) const {  // ppm_volume_mean_x :429-450
    const double b1 = 7.0 / 12.0, b2 = -1.0 / 12.0;
    if (i == g.is) {
      const double g_in = DXA(i + 1, j) / DXA(i, j), g_ou = DXA(i - 2, j) / DXA(i - 1, j);
      return 0.5 * (((2.0 + g_in) * Q(i, j) - Q(i + 1, j)) / (1.0 + g_in) + ((2.0 + g_ou) * Q(i - 1, j) - Q(i - 2, j)) / (1.0 + g_ou));
    }
    if (i == g.is + 1) {
      const double g_in = DXA(i, j) / DXA(i - 1, j), g_ou = DXA(i - 3, j) / DXA(i - 2, j);
      const double left = 0.5 * (((2.0 + g_in) * Q(i - 1, j) - Q(i, j)) / (1.0 + g_in) + ((2.0 + g_ou) * Q(i - 2, j) - Q(i - 3, j)) / (1.0 + g_ou));
      const double right = b2 * (Q(i - 1, j) + Q(i + 2, j)) + b1 * (Q(i, j) + Q(i + 1, j));
      return (3.0 * (g_in * Q(i - 1, j) + Q(i, j)) - (g_in * left + right)) / (2.0 + 2.0 * g_in);
    }
    if (i == g.ie + 1) {
      const double g_in = DXA(i - 2, j) / DXA(i - 1, j), g_ou = DXA(i + 1, j) / DXA(i, j);
      return 0.5 * (((2.0 + g_in) * Q(i - 1, j) - Q(i - 2, j)) / (1.0 + g_in) + ((2.0 + g_ou) * Q(i, j) - Q(i + 1, j)) / (1.0 + g_ou));
    }
    if (i == g.ie) {
      const double g_in = DXA(i - 1, j) / DXA(i, j), g_ou = DXA(i + 2, j) / DXA(i + 1, j);
      const double right = 0.5 * (((2.0 + g_in) * Q(i, j) - Q(i - 1, j)) / (1.0 + g_in) + ((2.0 + g_ou) * Q(i + 1, j) - Q(i + 2, j)) / (1.0 + g_ou));
      const double left = b2 * (Q(i - 3, j) + Q(i, j)) + b1 * (Q(i - 2, j) + Q(i - 1, j));
      return (3.0 * (Q(i - 1, j) + g_in * Q(i, j)) - (g_in * right + left)) / (2.0 + 2.0 * g_in);
    }
    return b2 * (Q(i - 2, j) + Q(i + 1, j)) + b1 * (Q(i - 1, j) + Q(i, j));
  }
  __device__ double qy(int i, int j) const {  // ppm_volume_mean_y :453-473
    const double b1 = 7.0 / 12.0, b2 = -1.0 / 12.0;
    if (j == g.js) {
      const double g_in = DYA(i, j + 1) / DYA(i, j), g_ou = DYA(i, j - 2) / DYA(i, j - 1);
      return 0.5 * (((2.0 + g_in) * Q(i, j) - Q(i, j + 1)) / (1.0 + g_in) + ((2.0 + g_ou) * Q(i, j - 1) - Q(i, j - 2)) / (1.0 + g_ou));
    }
    if (j == g.js + 1) {
      const double g_in = DYA(i, j) / DYA(i, j - 1), g_ou = DYA(i, j - 3) / DYA(i, j - 2);
      const double lower = 0.5 * (((2.0 + g_in) * Q(i, j - 1) - Q(i, j)) / (1.0 + g_in) + ((2.0 + g_ou) * Q(i, j - 2) - Q(i, j - 3)) / (1.0 + g_ou));
      const double upper = b2 * (Q(i, j - 1) + Q(i, j + 2)) + b1 * (Q(i, j) + Q(i, j + 1));
      return (3.0 * (g_in * Q(i, j - 1) + Q(i, j)) - (g_in * lower + upper)) / (2.0 + 2.0 * g_in);
    }
    if (j == g.je + 1) {
      const double g_in = DYA(i, j - 2) / DYA(i, j - 1), g_ou = DYA(i, j + 1) / DYA(i, j);
      return 0.5 * (((2.0 + g_in) * Q(i, j - 1) - Q(i, j - 2)) / (1.0 + g_in) + ((2.0 + g_ou) * Q(i, j) - Q(i, j + 1)) / (1.0 + g_ou));
    }
    if (j == g.je) {
      const double g_in = DYA(i, j - 1) / DYA(i, j), g_ou = DYA(i, j + 2) / DYA(i, j + 1);
      const double lower = b2 * (Q(i, j - 3) + Q(i, j)) + b1 * (Q(i, j - 2) + Q(i, j - 1));
      const double upper = 0.5 * (((2.0 + g_in) * Q(i, j) - Q(i, j - 1)) / (1.0 + g_in) + ((2.0 + g_ou) * Q(i, j + 1) - Q(i, j + 2)) / (1.0 + g_ou));
      return (3.0 * (Q(i, j - 1) + g_in * Q(i, j)) - (g_in * upper + lower)) / (2.0 + 2.0 * g_in);
    }
    return b2 * (Q(i, j - 2) + Q(i, j + 1)) + b1 * (Q(i, j - 1) + Q(i, j));
  }
  // qout_x_edge :286-304 (west/east columns), qout_y_edge :307-325 (south/north rows)
  __device__ __forceinline__ double q2(int i, int j) const {
    return (Q(i - 1, j) * DXA(i, j) + Q(i, j) * DXA(i - 1, j)) / (DXA(i - 1, j) + DXA(i, j));
  }
  __device__ __forceinline__ double q1(int i, int j) const {
    return (Q(i, j - 1) * DYA(i, j) + Q(i, j) * DYA(i, j - 1)) / (DYA(i, j - 1) + DYA(i, j));
  }
  __device__ __forceinline__ double edge_x(int i, int j) const {
    const double e = (i == g.is) ? m.edge_w[j] : m.edge_e[j];
    return e * q2(i, j - 1) + (1.0 - e) * q2(i, j);
  }
  __device__ __forceinline__ double edge_y(int i, int j) const {
    const double e = (j == g.js) ? m.edge_s[i] : m.edge_n[i];
    return e * q1(i - 1, j) + (1.0 - e) * q1(i, j);
  }
  __device__ double corner(int which, int i, int j) const {  // a2b_ord4.py:43-273
    // diagonals: 0 = (0,0)/(1,1), 1 = (-1,0)/(-2,1), 2 = (0,-1)/(1,-2), 3 = (-1,-1)/(-2,-2)
    const int set[4][3] = {{0, 1, 2}, {1, 3, 0}, {3, 2, 1}, {2, 3, 0}};
    const int o1[4][2] = {{0, 0}, {-1, 0}, {0, -1}, {-1, -1}};
    const int o2[4][2] = {{1, 1}, {-2, 1}, {1, -2}, {-2, -2}};
    double ec[3];
    for (int t = 0; t < 3; ++t) {
      const int d = set[which][t];
      const double qa = Q(i + o1[d][0], j + o1[d][1]), qb = Q(i + o2[d][0], j + o2[d][1]);
      ec[t] = qa + m.a2b_corner_w[which][t] * (qa - qb);
    }
    return (ec[0] + ec[1] + ec[2]) * (1.0 / 3.0);
  }
  __device__ __forceinline__ double point_interior(int i, int j) const { return a2b_interior_point(PlaneInMemory{src.q, g.sj}, i, j); }
  __device__ double point(int i, int j) const {  // value of qout at B-grid point (i, j), is <= i,j <= ie+1
    const double a1 = 9.0 / 16.0, a2 = -1.0 / 16.0, c1 = 2.0 / 3.0, c2 = -1.0 / 6.0;
    const bool iw = (i == g.is), ie_ = (i == g.ie + 1), js_ = (j == g.js), jn = (j == g.je + 1);
    if (iw && js_) return corner(0, i, j);
    if (ie_ && js_) return corner(1, i, j);
    if (ie_ && jn) return corner(2, i, j);
    if (iw && jn) return corner(3, i, j);
    if (iw || ie_) return edge_x(i, j);
    if (js_ || jn) return edge_y(i, j);
    double qxx, qyy;
    if (j == g.js + 1) {
      const double up = a2 * (qx(i, j - 1) + qx(i, j + 2)) + a1 * (qx(i, j) + qx(i, j + 1));
      qxx = c1 * (qx(i, j - 1) + qx(i, j)) + c2 * (edge_y(i, j - 1) + up);
    } else if (j == g.je) {
      const double lo = a2 * (qx(i, j - 3) + qx(i, j)) + a1 * (qx(i, j - 2) + qx(i, j - 1));
      qxx = c1 * (qx(i, j - 1) + qx(i, j)) + c2 * (edge_y(i, j + 1) + lo);
    } else {
      qxx = a2 * (qx(i, j - 2) + qx(i, j + 1)) + a1 * (qx(i, j - 1) + qx(i, j));
    }
    if (i == g.is + 1) {
      const double rt = a2 * (qy(i - 1, j) + qy(i + 2, j)) + a1 * (qy(i, j) + qy(i + 1, j));
      qyy = c1 * (qy(i - 1, j) + qy(i, j)) + c2 * (edge_x(i - 1, j) + rt);
    } else if (i == g.ie) {
      const double lf = a2 * (qy(i - 3, j) + qy(i, j)) + a1 * (qy(i - 2, j) + qy(i - 1, j));
      qyy = c1 * (qy(i - 1, j) + qy(i, j)) + c2 * (edge_x(i + 1, j) + lf);
    } else {
      qyy = a2 * (qy(i - 2, j) + qy(i + 1, j)) + a1 * (qy(i - 1, j) + qy(i, j));
    }
    return 0.5 * (qxx + qyy);
  }
};
typedef A2BT<A2BInMemory> A2B;

__global__ void __launch_bounds__(256)
k_a2b_ord4(Geo g, Met m, const real* __restrict__ qin, real* __restrict__ qout, int k0, Regions R) {
  REGION_POINT(R);
  const int kk = k + k0;
  A2B a{g, m, {qin + (long)kk * g.sk}};
  qout[IDX3(g, i, j, kk)] = interior ? a.point_interior(i, j) : a.point(i, j);
}

__global__ void __launch_bounds__(256) k_copy_window(Geo g, const real* __restrict__ src, real* __restrict__ dst, int k0, int i1, int j1) {
  PLANE_IJK(g);
  const int kk = k + k0;
  if (i < g.is || i > i1 || j < g.js || j > j1) return;
  const long c = IDX3(g, i, j, kk);
  dst[c] = src[c];
}

// tail of DivergenceDamping for nord > 0 levels: a2b_ord4(wk) -> smagorinsky -> damping
__global__ void __launch_bounds__(256)
k_divdamp_high_final(Geo g, Met m, const real* __restrict__ wk, const real* delpc_src, real* divgd_out,
                     real* __restrict__ delpc, const real* __restrict__ divg_d, real* __restrict__ vort_b,
                     real* __restrict__ ke, const real* __restrict__ d2_bg, double dddmp, double dd8, double absdt,
                     int k0, Regions R) {
  REGION_POINT(R);
  const int kk = k + k0;
  const long c = IDX3(g, i, j, kk);
  const double dpc = delpc_src[c];  // copy_computeplus :578
  delpc[c] = dpc;
  double vb;
  if (dddmp < 1e-5) {
    vb = 0.0;
  } else {
    A2B a{g, m, {wk + (long)kk * g.sk}};
    const double qb = interior ? a.point_interior(i, j) : a.point(i, j);
    vb = absdt * sqrt(dpc * dpc + qb * qb);
  }
  const double damp = m.da_min_c * fmax(d2_bg[kk], fmin(0.2, dddmp * fabs(vb)));
  const double dfin = divg_d[c];
  const double vort = damp * dpc + dd8 * dfin;
  vort_b[c] = vort;
  ke[c] = ke[c] + vort;
  divgd_out[c] = dfin;  // the caller's divgd ends as the iterated divergence (redo_divg_d; compared by TranslateD_SW)
}

// ------------------------------------------------------------------------------------------------
// DivergenceDamping, nord > 0 levels, as ONE kernel per (tile, level): the `nord` passes of the divergence of the gradient
// of the divergence run in LDS on the tile's footprint (halo `nord` <= 3, shrinking by one per pass like the reference's
// domains, divergence_damping.py:579-600), the last pass at the tile's own points with its uc / vc kept for the caller,
// then the tail (a2b_ord4 of the relative vorticity, Smagorinsky term, damped vorticity, ke += ...).  `din` holds the
// divergence c_sw left (full contract: a copy of it on the whole plane in a scratch field, so that this kernel may overwrite
// divg_d while neighbouring tiles still read their halo; PACE_DSW_SKIP_DEAD_OUTPUTS: divg_d itself, which is then not written).
// Replaces 3 x k_divdamp_iter + k_divdamp_high_final: 10 field passes instead of 16.
// ------------------------------------------------------------------------------------------------
#ifndef DD_TI
#define DD_TI 65  // 193 = 3 * 65 - 2 B-grid points per row at C192
#define DD_TJ 17  // 193 = 12 * 17 - 11
#endif
#ifndef DD_STAMP
#define DD_STAMP(n)  // (tools/census/dsw_prof.hip: wall-clock stamps of every workgroup of k_divdamp_fused)
#endif
#ifdef PACE_EMU
#define DD_ATTR
#elif !defined(DD_ATTR)
#define DD_ATTR __attribute__((amdgpu_waves_per_eu(3, 3)))  // three workgroups per compute unit (168 registers)
#endif
#define DD_W (DD_TI + 6)
#define DD_H (DD_TJ + 6)
#ifndef DD_NT
#define DD_NT 256  // threads per workgroup of k_divdamp_fused
#endif
#define DD_STRIP (DD_W * DD_H / 8 - 6 < 96 ? DD_W * DD_H / 8 - 6 : 96)  // longest strip of the frame (own points along the edge) per workgroup
// an LDS plane: a tile's footprint, or a strip's and, behind it, the cell widths (dxa, dya) on that footprint
#define DD_PLANE (DD_W * DD_H > 2 * (DD_STRIP + 6) * 8 ? DD_W * DD_H : 2 * (DD_STRIP + 6) * 8)

// The launch of k_divdamp_fused is a 1-D grid of three kinds of workgroups, in this order: the sponge levels (second-order
// damping, a point function of the winds: 256 points each), the strips (the frame of the B-grid domain, two points deep, of a level
// with nord > 0: the tail through the general point functions), the tiles (DD_TI x DD_TJ points of such a level, the plain
// forms of the tail only).  The first two are few, long-running workgroups: first in launch order, they run beside the tiles.  (Round 4 had
// the sponge levels last and the frame inside the edge tiles, 18 of the 24 tiles of a C192 level: 85 us, of which 12 us the
// frame alone.)
struct DdSponge {
  const real *u, *v, *ua, *va, *uc, *vc;
  real* delpc;
  double dt;
  int nlev;      // sponge levels [0, nlev) taken by this launch (0: none)
  int nblocks;   // their workgroups: nlev * ceil((n + 1)^2 / DD_NT)
  int nstrips;   // strip workgroups: strips per level * levels (0: the tiles take their frame themselves)
  int nch_row, len_row, nch_col, len_col;  // chunks per edge and their length, rows (south / north) and columns (west / east)
};

template <class Plane>
__device__ __forceinline__ void divdamp_point(const Geo& g, const Met& m, const Plane& src, int i, int j, bool fill, double& d,
                                              double& uc_here, double& vc_here) {
  const long c2 = IDX2(g, i, j);
  if (i > g.is && i <= g.ie) {
    // columns is+1 .. ie: no operand lies in a corner region and no corner adjustment applies
    const int sj = g.sj;
    const double d0 = src(i, j);
    const double ucm = (d0 - src(i, j - 1)) * m.divg_v[c2 - sj];
    const double uc0 = (src(i, j + 1) - d0) * m.divg_v[c2];
    const double vcm = (d0 - src(i - 1, j)) * m.divg_u[c2 - 1];
    const double vc0 = (src(i + 1, j) - d0) * m.divg_u[c2];
    d = ucm - uc0 + vcm - vc0;
    uc_here = uc0;
    vc_here = vc0;
  } else {
    DivIterT<Plane> it{g, m, src, fill};
    const double ucm = it.uc(i, j - 1), uc0 = it.uc(i, j), vcm = it.vc(i - 1, j), vc0 = it.vc(i, j);
    d = ucm - uc0 + vcm - vc0;  // redo_divg_d :212-240
    const bool ic = (i == g.is || i == g.ie + 1);
    if (ic && j == g.js) d = d - ucm;
    if (ic && j == g.je + 1) d = d + uc0;
    uc_here = uc0;
    vc_here = vc0;
  }
  d = d * m.rarea_c[c2];
}

// One workgroup of k_divdamp_fused on the own points [i0, i1] x [j0, j1] (at most TI x TJ) of level kk.  Every operand comes from
// memory ONCE, at the start and in one go -- the divergence and the relative vorticity on the footprint, the five metric values
// of every footprint point (the same in every pass and in the tail), for a strip the cell widths a2b_ord4's edge forms divide
// by; everything after the first barrier reads registers and LDS.  (Round 4's version fetched the metric values pass by pass
// and batch by batch and spent 16 of a workgroup's 18.5 us waiting for seven such round trips one after the other; a strip whose
// a2b_ord4 read memory took 25 us -- tools/dd_stage_times.py.)
// The plain form of a pass applies wherever no operand lies in a corner region and no corner adjustment applies: columns
// is+1 .. ie (any row) or rows js+1 .. je (any column); only the points near a tile corner take the general form.
// MODE 0: a tile -- the tail at the own points where a2b_ord4 is the 16-point mean (the rest of them belong to the strips);
// 2: a strip of the frame -- the tail at every own point, a2b_ord4's general forms on LDS planes; 1: the same from memory (a
// tile too small to have an interior: its footprint leaves no room for the planes of the cell widths).
template <int TI, int TJ, int MODE>
__device__ __forceinline__ void
divdamp_tile(double (*sbuf)[DD_PLANE], int i0, int i1, int j0, int j1, int kk, const Geo& g, const Met& m,
             const real* __restrict__ wk, const real* din, real* divg_d, real* __restrict__ vort_b,
             real* __restrict__ ke, real* __restrict__ uc_out, real* __restrict__ vc_out, const real* __restrict__ d2_bg, double dddmp,
             double dd8, double absdt, int nord, bool full) {
  // The divergence before the passes: `din` at the tile's own B-grid points [is, ie+1]^2, divg_d in the halo.  full (the
  // reference's contract): din = delpc, which the caller has set to the divergence there (copy_computeplus), and divg_d is
  // WRITTEN at the tile's points only -- no workgroup reads what another one writes.  full == false
  // (PACE_DSW_SKIP_DEAD_OUTPUTS): divg_d, uc_out, vc_out are not written; din is divg_d itself.
  constexpr int W = TI + 6, H = TJ + 6, NE = (W * H + DD_NT - 1) / DD_NT;
  static_assert((MODE == 2 ? 2 : 1) * W * H <= DD_PLANE, "the footprint (a strip's: twice) fits an LDS plane");
  static_assert(NE <= 32, "one bit per footprint point of a thread");
  const int tid = threadIdx.x;
  const int ilo = i0 - 3, jlo = j0 - 3, sj = g.sj;
  const long kb = (long)kk * g.sk;
  double dreg[NE], gvm[NE], gv0[NE], gum[NE], gu0[NE], ra[NE];
  unsigned plain = 0;  // bit t: this thread's footprint point t takes the plain form of the passes
#pragma unroll
  for (int t = 0; t < NE; ++t) {
    const int e = tid + DD_NT * t;
    const int jj = e / W, ii = e - jj * W;
    const int gi = ilo + ii, gj = jlo + jj;
    const bool ok = e < W * H && gi < g.ni && gj < g.nj;
    const long c2 = ok ? IDX2(g, gi, gj) : 0;
    const bool own_pt = gi >= g.is && gi <= g.ie + 1 && gj >= g.js && gj <= g.je + 1;
    dreg[t] = (own_pt ? din : (const real*)divg_d)[kb + c2];
    if (!ok) dreg[t] = 0.0;
    // (the widest pass reaches two points beyond the own points: nord <= 3)
    const bool pl = ok && gi >= i0 - 2 && gi <= i1 + 2 && gj >= j0 - 2 && gj <= j1 + 2 &&
                    ((gi > g.is && gi <= g.ie) || (gj > g.js && gj <= g.je));
    plain |= (pl ? 1u : 0u) << t;
    const long cm = pl ? c2 : IDX2(g, g.is + 1, g.js + 1);
    gvm[t] = m.divg_v[cm - sj];
    gv0[t] = m.divg_v[cm];
    gum[t] = m.divg_u[cm - 1];
    gu0[t] = m.divg_u[cm];
    ra[t] = m.rarea_c[cm];
  }
  {
    // the relative vorticity under the tile to a plane of its own, a strip's cell widths behind its planes of the passes
    double wreg[NE], xreg[MODE == 2 ? NE : 1], yreg[MODE == 2 ? NE : 1];
#pragma unroll
    for (int t = 0; t < NE; ++t) {
      const int e = tid + DD_NT * t;
      const int jj = e / W, ii = e - jj * W;
      const bool ok = e < W * H && ilo + ii < g.ni && jlo + jj < g.nj;
      const long c2 = ok ? IDX2(g, ilo + ii, jlo + jj) : 0;
      wreg[t] = wk[kb + c2];
      if (MODE == 2) xreg[t] = m.dxa[c2], yreg[t] = m.dya[c2];
    }
#pragma unroll
    for (int t = 0; t < NE; ++t) {
      const int e = tid + DD_NT * t;
      if (e < W * H) {
        sbuf[0][e] = dreg[t];
        sbuf[2][e] = wreg[t];
        if (MODE == 2) sbuf[0][W * H + e] = xreg[t], sbuf[1][W * H + e] = yreg[t];
      }
    }
  }
  __syncthreads();
  DD_STAMP(1);
  int cur = 0;
  for (int n = 1; n < nord; ++n) {
    const int nt = nord - n;
    const int ia = i0 - nt, ib = i1 + nt, ja = j0 - nt, jb = j1 + nt;
    const double* src = sbuf[cur];
    double* dst = sbuf[cur ^ 1];
#pragma unroll
    for (int t = 0; t < NE; ++t) {
      const int e = tid + DD_NT * t;
      const int jj = e / W, ii = e - jj * W;
      const int gi = ilo + ii, gj = jlo + jj;
      if (((plain >> t) & 1u) && gi >= ia && gi <= ib && gj >= ja && gj <= jb) {
        const double d0 = src[e];
        const double ucm = (d0 - src[e - W]) * gvm[t];
        const double uc0 = (src[e + W] - d0) * gv0[t];
        const double vcm = (d0 - src[e - 1]) * gum[t];
        const double vc0 = (src[e + 1] - d0) * gu0[t];
        double d = ucm - uc0 + vcm - vc0;
        d = d * ra[t];
        dst[e] = d;
      }
    }
    // the rest of the pass's domain: columns <= is or > ie in rows <= js or > je (a corner region among the operands, the corner
    // adjustment of redo_divg_d) -- a handful of points of the corner tiles, enumerated densely
    const int cl1 = g.is < ib ? g.is : ib, cr0 = g.ie + 1 > ia ? g.ie + 1 : ia;
    const int rb1 = g.js < jb ? g.js : jb, rt0 = g.je + 1 > ja ? g.je + 1 : ja;
    const int ncl = cl1 >= ia ? cl1 - ia + 1 : 0, ncr = ib >= cr0 ? ib - cr0 + 1 : 0;
    const int nrb = rb1 >= ja ? rb1 - ja + 1 : 0, nrt = jb >= rt0 ? jb - rt0 + 1 : 0;
    const int ncol = ncl + ncr, total = ncol * (nrb + nrt);
    if (total > 0) {  // block-uniform
      const PlaneInLds plane{src, ilo, jlo, W};
      for (int p = tid; p < total; p += DD_NT) {
        const int r = p / ncol, cx = p - r * ncol;
        const int gi = cx < ncl ? ia + cx : cr0 + (cx - ncl), gj = r < nrb ? ja + r : rt0 + (r - nrb);
        double d, u_, v_;
        divdamp_point(g, m, plane, gi, gj, true, d, u_, v_);
        dst[(gj - jlo) * W + (gi - ilo)] = d;
      }
    }
    __syncthreads();
    cur ^= 1;
  }
  DD_STAMP(2);
  const double d2 = d2_bg[kk];
  const double* src = sbuf[cur];
  const double* swk = sbuf[2];
#pragma unroll
  for (int t = 0; t < NE; ++t) {
    const int e = tid + DD_NT * t;
    const int jj = e / W, ii = e - jj * W;
    const int i = ilo + ii, j = jlo + jj;
    bool on = e < W * H && i >= i0 && i <= i1 && j >= j0 && j <= j1;
    // a tile: the own points where a2b_ord4 is the 16-point mean (there the pass is plain, too)
    if (MODE == 0) on = on && i >= g.is + 2 && i <= g.ie - 1 && j >= g.js + 2 && j <= g.je - 1;
    if (!on) continue;
    double dfin, uc0, vc0;
    if (MODE == 0 || ((plain >> t) & 1u)) {
      const double d0 = src[e];
      const double ucm = (d0 - src[e - W]) * gvm[t];
      uc0 = (src[e + W] - d0) * gv0[t];
      const double vcm = (d0 - src[e - 1]) * gum[t];
      vc0 = (src[e + 1] - d0) * gu0[t];
      dfin = ucm - uc0 + vcm - vc0;
      dfin = dfin * ra[t];
    } else {
      divdamp_point(g, m, PlaneInLds{src, ilo, jlo, W}, i, j, false, dfin, uc0, vc0);
    }
    const long c = kb + IDX2(g, i, j);
    // The reference uses the caller's uc / vc as the work fields of the passes and its Translate tests compare what is left
    // in them after d_sw on the staggered compute windows (translate_d_sw.py:36-65): the values of the LAST pass.
    if (full && j <= g.je) uc_out[c] = uc0;
    if (full && i <= g.ie) vc_out[c] = vc0;
    const double dpc = dreg[t];  // the divergence before the passes (= delpc)
    double vb;
    if (dddmp < 1e-5) {
      vb = 0.0;
    } else {
      double qb;
      if (MODE == 0) {
        qb = a2b_interior_point(PlaneInLds{swk, ilo, jlo, W}, i, j);
      } else if (MODE == 2) {
        A2BT<A2BInLds> a{g, m, A2BInLds{swk, sbuf[0] + W * H, sbuf[1] + W * H, ilo, jlo, W}};
        qb = a.point(i, j);
      } else {
        A2B a{g, m, {wk + kb}};
        qb = a.point(i, j);
      }
      vb = absdt * sqrt(dpc * dpc + qb * qb);
    }
    const double damp = m.da_min_c * fmax(d2, fmin(0.2, dddmp * fabs(vb)));
    const double vort = damp * dpc + dd8 * dfin;
    vort_b[c] = vort;
    if (ke != nullptr) ke[c] = ke[c] + vort;
    if (full) divg_d[c] = dfin;  // the caller's divgd ends as the iterated divergence (redo_divg_d; compared by TranslateD_SW)
  }
  DD_STAMP(3);
}

__global__ void __launch_bounds__(DD_NT) DD_ATTR
k_divdamp_fused(Geo g, Met m, const real* __restrict__ wk, const real* din, real* divg_d,
                real* __restrict__ vort_b, real* __restrict__ ke, real* __restrict__ uc_out, real* __restrict__ vc_out,
                const real* __restrict__ d2_bg, double dddmp, double dd8, double absdt, int k0, int nord, int ntx, int ntiles, int full_,
                DdSponge sp) {
  __shared__ double sbuf[3][DD_PLANE];  // two planes of the passes, one of the relative vorticity
  int b = (int)blockIdx.x;
  const bool full = full_ != 0;
  DD_STAMP(0);
  if (b < sp.nblocks) {
    // a sponge level: 256 B-grid points of it
    const int per = sp.nblocks / sp.nlev, kk = b / per, w = g.n + 1;
    const int q = (b - kk * per) * DD_NT + (int)threadIdx.x;
    if (q < w * w) {
      const int jj = q / w;
      divdamp_low_point(g, m, sp.u, sp.v, sp.ua, sp.va, sp.uc, sp.vc, sp.delpc, vort_b, ke, d2_bg[kk], dddmp, sp.dt, g.is + (q - jj * w),
                        g.js + jj, kk);
    }
    DD_STAMP(7);
    return;
  }
  b -= sp.nblocks;
  if (b < sp.nstrips) {
    const int per = 2 * (sp.nch_row + sp.nch_col);
    const int lev = b / per, id = b - lev * per;
    if (id < 2 * sp.nch_row) {  // south, north rows: the full width
      const int side = id / sp.nch_row, ch = id - side * sp.nch_row;
      const int i0 = g.is + ch * sp.len_row, j0 = side == 0 ? g.js : g.je;
      const int i1 = i0 + sp.len_row - 1 < g.ie + 1 ? i0 + sp.len_row - 1 : g.ie + 1;
      divdamp_tile<DD_STRIP, 2, 2>(sbuf, i0, i1, j0, j0 + 1, lev + k0, g, m, wk, din, divg_d, vort_b, ke, uc_out, vc_out, d2_bg, dddmp, dd8,
                                      absdt, nord, full);
    } else {  // west, east columns between them
      const int id2 = id - 2 * sp.nch_row;
      const int side = id2 / sp.nch_col, ch = id2 - side * sp.nch_col;
      const int j0 = g.js + 2 + ch * sp.len_col, i0 = side == 0 ? g.is : g.ie;
      const int j1 = j0 + sp.len_col - 1 < g.je - 1 ? j0 + sp.len_col - 1 : g.je - 1;
      divdamp_tile<2, DD_STRIP, 2>(sbuf, i0, i0 + 1, j0, j1, lev + k0, g, m, wk, din, divg_d, vort_b, ke, uc_out, vc_out, d2_bg, dddmp, dd8,
                                      absdt, nord, full);
    }
    DD_STAMP(7);
    return;
  }
  b -= sp.nstrips;
  const int zblock = b / ntiles, tile = b - zblock * ntiles;
  const int by = tile / ntx, bx = tile - by * ntx;
  const int i0 = g.is + bx * DD_TI, j0 = g.js + by * DD_TJ;
  const int i1 = i0 + DD_TI - 1 < g.ie + 1 ? i0 + DD_TI - 1 : g.ie + 1;
  const int j1 = j0 + DD_TJ - 1 < g.je + 1 ? j0 + DD_TJ - 1 : g.je + 1;
  if (sp.nstrips > 0)
    divdamp_tile<DD_TI, DD_TJ, 0>(sbuf, i0, i1, j0, j1, zblock + k0, g, m, wk, din, divg_d, vort_b, ke, uc_out, vc_out, d2_bg, dddmp, dd8,
                                      absdt, nord, full);
  else  // a tile too small for a frame and an interior
    divdamp_tile<DD_TI, DD_TJ, 1>(sbuf, i0, i1, j0, j1, zblock + k0, g, m, wk, din, divg_d, vort_b, ke, uc_out, vc_out, d2_bg, dddmp, dd8,
                                     absdt, nord, full);
  DD_STAMP(7);
}

// ------------------------------------------------------------------------------------------------
// tail of d_sw: u_and_v_from_ke (:439-477), vort_differencing (:353-380) +
// heat_source_from_vorticity_damping (:493-577), update_u_and_v (:582-608)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_uv_from_ke(Geo g, Met m, real* __restrict__ u, real* __restrict__ v, const real* __restrict__ ke,
             const real* __restrict__ fx, const real* __restrict__ fy) {
  PATCH_IJK(g);
  const long c = IDX3(g, i, j, k);
  const long c2 = IDX2(g, i, j);
  if (i >= g.is && i <= g.ie && j >= g.js && j <= g.je + 1) u[c] = u[c] * m.dx[c2] + ke[c] - ke[c + 1] + fy[c];
  if (i >= g.is && i <= g.ie + 1 && j >= g.js && j <= g.je) v[c] = v[c] * m.dy[c2] + ke[c] - ke[c + g.sj] - fx[c];
}

struct HeatPt {
  double ubt, vbt, fy, fx, gy, gx;
};

__device__ __forceinline__ HeatPt heat_point(const Geo& g, const Met& m, const real* u, const real* v,
                                             const real* vort_b, const real* ut2, const real* vt2, bool don,
                                             long c, long c2, int i, int j) {
  // vort_x_delta / vort_y_delta are only written where d_con > threshold and inside their regions
  // (d_sw.py:373-380); elsewhere the reference reads its zero-initialised persistent temporaries.
  double vxd = 0.0, vyd = 0.0;
  if (don) {
    if (i >= g.is && i <= g.ie && j >= g.js && j <= g.je + 1) vxd = vort_b[c] - vort_b[c + 1];
    if (i >= g.is && i <= g.ie + 1 && j >= g.js && j <= g.je) vyd = vort_b[c] - vort_b[c + g.sj];
  }
  HeatPt h;
  h.ubt = (vxd + vt2[c]) * m.rdx[c2];
  h.fy = u[c] * m.rdx[c2];
  h.gy = h.fy * h.ubt;
  h.vbt = (vyd - ut2[c]) * m.rdy[c2];
  h.fx = v[c] * m.rdy[c2];
  h.gx = h.fx * h.vbt;
  return h;
}

// heat_source_from_vorticity_damping (d_sw.py:493-577) and update_u_and_v (:582-608) in one pass: the winds after
// u_and_v_from_ke are read from the workspace copy the transport kernel's store phase left (umid / vmid, on the windows
// (n, n+1) / (n+1, n)), the final winds are written to u / v -- no thread reads what another one writes.
__global__ void __launch_bounds__(256)
k_heat_source(Geo g, Met m, const real* __restrict__ umid, const real* __restrict__ vmid,
              const real* __restrict__ vort_b, const real* __restrict__ ut2, const real* __restrict__ vt2,
              const real* __restrict__ delp, const real* __restrict__ heat_s, real* __restrict__ heat_source,
              real* __restrict__ diss_est, const real* __restrict__ d_con_k, double d_con, int do_skeb,
              real* __restrict__ u, real* __restrict__ v, const real* __restrict__ damp_vt) {
  PATCH_IJK(g);
  if (i < g.is || i > g.ie + 1 || j < g.js || j > g.je + 1) return;
  const long c = IDX3(g, i, j, k);
  const long c2 = IDX2(g, i, j);
  const bool upd = damp_vt[k] > 1e-5;
  if (i <= g.ie) u[c] = upd ? umid[c] + vt2[c] : umid[c];
  if (j <= g.je) v[c] = upd ? vmid[c] - ut2[c] : vmid[c];
  if (i > g.ie || j > g.je) return;  // (the reference forms heat_s on compute + 1; only the compute domain of it is ever read)
  const double dck = d_con_k[k];
  const bool don = dck > DCON_THRESHOLD;
  if ((dck > DCON_THRESHOLD) || do_skeb) {
    const HeatPt p0 = heat_point(g, m, umid, vmid, vort_b, ut2, vt2, don, c, c2, i, j);
    const HeatPt pj = heat_point(g, m, umid, vmid, vort_b, ut2, vt2, don, c + g.sj, c2 + g.sj, i, j + 1);
    const HeatPt pi = heat_point(g, m, umid, vmid, vort_b, ut2, vt2, don, c + 1, c2 + 1, i + 1, j);
    const double u2 = p0.fy + pj.fy, du2 = p0.ubt + pj.ubt, v2 = p0.fx + pi.fx, dv2 = p0.vbt + pi.vbt;
    const double dampterm = m.rsin2[c2] * 0.25 *
                            ((p0.ubt * p0.ubt + pj.ubt * pj.ubt + p0.vbt * p0.vbt + pi.vbt * pi.vbt) +
                             2.0 * (p0.gy + pj.gy + p0.gx + pi.gx) - m.cosa_s[c2] * (u2 * dv2 + v2 * du2 + du2 * dv2));
    const double hs = delp[c] * (heat_s[c] - dck * dampterm);  // (the reference's temporary: consumed right here, not stored)
    if (d_con > DCON_THRESHOLD || do_skeb) {
      heat_source[c] = heat_source[c] + hs;
      if (do_skeb) diss_est[c] = diss_est[c] - dampterm;
    }
  } else if (d_con > DCON_THRESHOLD || do_skeb) {
    heat_source[c] = heat_source[c] + heat_s[c];
  }
}

// =================================================================================================
static Regions a2b_regions(const Geo& g) {
  if (g.n >= 8) return bgrid_regions(g, 2);
  Regions r{};  // tiny tiles: no interior; one (empty-interior) strip covering everything
  r.n = 1; r.ib[0] = 0; r.ie[0] = -1; r.jb[0] = 0; r.je[0] = -1; r.nbx0 = 1; r.first[1] = 0;
  add_region(r, g.is, g.ie + 1, g.js, g.je + 1);
  return r;
}

int launch_a2b_ord4(const Geo& g, const Met& m, real* qin, real* qout, int k0, int k1, int replace, hipStream_t st) {
  const int nlev = k1 - k0;
  const Regions r = a2b_regions(g);
  hipLaunchKernelGGL(k_a2b_ord4, regions_grid(r, nlev), dim3(64, 4), 0, st, g, m, qin, qout, k0, r);
  if (replace) hipLaunchKernelGGL(k_copy_window, plane_grid(g, nlev), dim3(256), 0, st, g, qout, qin, k0, g.ie + 1, g.je + 1);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}

__global__ void __launch_bounds__(256) k_copy_levels(Geo g, const real* __restrict__ src, real* __restrict__ dst, int k0) {
  PLANE_IJK(g);
  const long c = IDX3(g, i, j, k + k0);
  dst[c] = src[c];
}

// the two plane-wise preliminaries of the divergence damping in ONE launch: second-order damping on the sponge levels
// [0, kstart) (k_divdamp_low's arithmetic) and delpc = divg_d on the levels below (k_copy_levels) -- two tiny kernels were two
// launch latencies on the critical path of the wind phase
__global__ void __launch_bounds__(256)
k_divdamp_low_and_copy(Geo g, Met m, const real* __restrict__ u, const real* __restrict__ v, const real* __restrict__ ua,
                       const real* __restrict__ va, const real* __restrict__ uc, const real* __restrict__ vc,
                       real* __restrict__ delpc, real* __restrict__ vort_b, real* __restrict__ ke, const real* __restrict__ d2_bg,
                       double dddmp, double dt, const real* __restrict__ divg_d, int kstart) {
  PLANE_IJK(g);
  const long c = IDX3(g, i, j, k);
  const bool own = i >= g.is && i <= g.ie + 1 && j >= g.js && j <= g.je + 1;
  if (k >= kstart) {
    // copy_computeplus (divergence_damping.py:578): delpc = divg_d on the B-grid points of the tile -- delpc's halo stays what c_sw
    // left there (TranslateD_SW compares it on the whole storage).  The fused kernel then takes the divergence before the passes from
    // delpc at the tile's points and from divg_d in the halo: it writes divg_d at the tile's points only, so nobody reads what
    // another workgroup writes
    if (own) delpc[c] = divg_d[c];
    return;
  }
  if (!own) return;
  const long c2 = IDX2(g, i, j);
  const int sj = g.sj;
  const double a0 = dd_u_contra_dyc(g, m, u, va, vc, c, c2, j);
  const double am = dd_u_contra_dyc(g, m, u, va, vc, c - 1, c2 - 1, j);
  const double b0 = dd_v_contra_dxc(g, m, v, ua, uc, c, c2, i);
  const double bm = dd_v_contra_dxc(g, m, v, ua, uc, c - sj, c2 - sj, i);
  double d = bm - b0 + am - a0;
  const bool ic = (i == g.is || i == g.ie + 1);
  if (ic && j == g.js) d = d - bm;
  if (ic && j == g.je + 1) d = d + b0;
  d = m.rarea_c[c2] * d;
  delpc[c] = d;
  const double delpcdt = d * dt;
  const double damp = m.da_min_c * fmax(d2_bg[k], fmin(0.2, dddmp * fabs(delpcdt)));
  const double vort = damp * d;
  vort_b[c] = vort;
  ke[c] = ke[c] + vort;
}

// PACE_LEGACY_DIVERGENCE_DAMPING=1: the round-1 sequence (one kernel per pass + the tail kernel), kept for A/B measurements
static bool legacy_divergence_damping() {
  static const bool v = getenv("PACE_LEGACY_DIVERGENCE_DAMPING") != nullptr;
  return v;
}

// ------------------------------------------------------------------------------------------------
// a2b_ord4 with the input tile staged in LDS: one workgroup per (64 x 8 box of B-grid points inside is+2 .. ie-1, level).
// The 16-point mean of a point reads a 4 x 4 neighbourhood of cells: from global memory that is 16 loads per point through the
// texture path (34.5 us per field at C192 x 79, 1.5 TB/s of algorithmic traffic), staged it is 1.4 loads per point.  The
// frame of the tile domain (two points deep: edge and corner forms) goes through the general point function in a second,
// small launch.  qout must not alias qin.
// ------------------------------------------------------------------------------------------------
#ifndef AB_TI
#define AB_TI 64
#define AB_TJ 8
#endif
#define AB_W (AB_TI + 3)
#define AB_H (AB_TJ + 3)
#define AB_NE ((AB_W * AB_H + 255) / 256)
#define AB_NP ((AB_TI * AB_TJ + 255) / 256)

struct A2BBatch {  // up to four fields in one launch (blockIdx.y): nh_p_grad interpolates pp, pk3, gz and delp together
  const real* in[4];
  real* out[4];
  int k0[4], k1[4];
};

__global__ void __launch_bounds__(256)
k_a2b_interior_tiled(Geo g, Met m, A2BBatch job, int kmin, int ntx, Regions R) {
  // One launch for the interior tiles AND the frame (blocks [0, R.first[R.n]): every region of the frame through the general point
  // function -- long, divergent code on few points; as a launch of its own it took 32 us after the 46 us of the tiles, here it
  // runs beside them, first in launch order).  Launch with dim3(64, 4) threads.
  __shared__ double sq[AB_W * AB_H];
  const int f = (int)blockIdx.y;
  const int nfr = R.first[R.n];
  if ((int)blockIdx.x < nfr) {
    REGION_POINT(R);
    (void)interior;
    const int kf = k + kmin;
    if (kf < job.k0[f] || kf >= job.k1[f]) return;
    A2B a{g, m, {job.in[f] + (long)kf * g.sk}};
    job.out[f][IDX3(g, i, j, kf)] = a.point(i, j);
    return;
  }
  const int kk = (int)blockIdx.z + kmin;
  if (kk < job.k0[f] || kk >= job.k1[f]) return;  // block-uniform
  const real* __restrict__ qin = job.in[f];
  real* __restrict__ qout = job.out[f];
  const int tid = (int)threadIdx.y * 64 + (int)threadIdx.x;
  const int tile = (int)blockIdx.x - nfr;
  const int bx = tile % ntx, by = tile / ntx;
  const int i0 = g.is + 2 + bx * AB_TI, j0 = g.js + 2 + by * AB_TJ;  // first B-grid point of the tile
  const long kb = (long)kk * g.sk;
  double v[AB_NE];
#pragma unroll
  for (int t = 0; t < AB_NE; ++t) {  // cells i0-2 .. i0+TI, j0-2 .. j0+TJ
    const int e = tid + 256 * t;
    const int jj = e / AB_W, ii = e - jj * AB_W;
    const int gi = i0 - 2 + ii, gj = j0 - 2 + jj;
    const bool ok = e < AB_W * AB_H && gi < g.ni && gj < g.nj;
    v[t] = qin[kb + (ok ? IDX2(g, gi, gj) : 0)];
  }
#pragma unroll
  for (int t = 0; t < AB_NE; ++t) {
    const int e = tid + 256 * t;
    if (e < AB_W * AB_H) sq[e] = v[t];
  }
  __syncthreads();
  const PlaneInLds Q{sq, i0 - 2, j0 - 2, AB_W};
#pragma unroll
  for (int t = 0; t < AB_NP; ++t) {
    const int q = tid + 256 * t;
    const int jj = q / AB_TI, ii = q - jj * AB_TI;
    const int i = i0 + ii, j = j0 + jj;
    if (q < AB_TI * AB_TJ && i <= g.ie - 1 && j <= g.je - 1) qout[kb + IDX2(g, i, j)] = a2b_interior_point(Q, i, j);
  }
}

int launch_a2b_ord4_batch(const Geo& g, const Met& m, const real* const* qin, real* const* qout, const int* k0, const int* k1,
                          int nfields, hipStream_t st) {
  if (nfields < 1 || nfields > 4) return PACE_ERR_ARG;
  if (g.n < 8) {  // no interior box
    for (int f = 0; f < nfields; ++f) {
      const int rc = launch_a2b_ord4(g, m, const_cast<real*>(qin[f]), qout[f], k0[f], k1[f], 0, st);
      if (rc) return rc;
    }
    return PACE_OK;
  }
  A2BBatch job{};
  int kmin = k0[0], kmax = k1[0];
  for (int f = 0; f < nfields; ++f) {
    job.in[f] = qin[f]; job.out[f] = qout[f]; job.k0[f] = k0[f]; job.k1[f] = k1[f];
    kmin = k0[f] < kmin ? k0[f] : kmin;
    kmax = k1[f] > kmax ? k1[f] : kmax;
  }
  const int nlev = kmax - kmin;
  const int nbox = g.n - 3;  // points is+2 .. ie-1
  const int ntx = (nbox + AB_TI - 1) / AB_TI, nty = (nbox + AB_TJ - 1) / AB_TJ;
  Regions r{};  // the frame (region 0 is enumerated in 64 x 4 patches: a row strip suits that, a column strip would not)
  add_region(r, g.is + 2, g.ie - 1, g.js, g.js + 1);
  add_region(r, g.is + 2, g.ie - 1, g.je, g.je + 1);
  add_region(r, g.is, g.is + 1, g.js, g.je + 1);
  add_region(r, g.ie, g.ie + 1, g.js, g.je + 1);
  hipLaunchKernelGGL(k_a2b_interior_tiled, dim3((unsigned)(r.first[r.n] + ntx * nty), (unsigned)nfields, (unsigned)nlev), dim3(64, 4), 0, st,
                     g, m, job, kmin, ntx, r);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}

int launch_a2b_ord4_tiled(const Geo& g, const Met& m, const real* qin, real* qout, int k0, int k1, hipStream_t st) {
  return launch_a2b_ord4_batch(g, m, &qin, &qout, &k0, &k1, 1, st);
}

// ------------------------------------------------------------------------------------------------
// What the reference LEAVES in the halo of its work fields (full contract only: pace_dsw_config_t.flags without
// PACE_DSW_SKIP_DEAD_OUTPUTS).  DivergenceDamping iterates in place on the caller's divg_d, uc, vc over domains that reach
// nt = nord - 1, ..., 0 points into the halo and fills the corner points of all three between the steps
// (divergence_damping.py:579-600; corners.py:591-712, 987-1151), and TranslateD_SW compares uc, vc, divgd over the whole
// storage (translate_d_sw.py:36-65).  The fused kernel above produces the tile's own points only; this kernel -- one workgroup
// per level with nord > 0 -- replays the iteration LITERALLY on a band around the tile's edges (the whole halo and DDH_M points
// inside: what the halo's final values depend on lies within two points of the edge), in the reference's order:
//   A  vc on its domain from the divergence with corners filled in x, uc on its domain with corners filled in y
//   B  fill_corners_dgrid(vc, uc, -1) -- and the y fill is what stays in the divergence's corner points
//   C  the divergence on its domain from uc, vc (redo_divg_d)
// with a barrier after each.  uc / vc are the caller's arrays, written OUTSIDE their compute windows only (inside, the fused
// kernel has put the last pass's values, and a value needed there is formed again from the divergence: the window lies inside
// every pass's domain and touches no corner point); the divergence ping-pongs between two scratch planes, `d0` holding the
// divergence before the passes on the whole plane.  The band's innermost points go wrong by one point per pass (they read past
// the band); nothing the halo depends on reaches them.  ~5 k points per level.
// ------------------------------------------------------------------------------------------------
// Two forms.  LDS (tiles up to C200): the band's divergence (two planes) and uc / vc live in LDS for the whole kernel; a thread owns
// at most four band points and keeps, per point, the LDS places of its neighbours -- plain and with the corner fills applied as
// index maps --, its five metric values and a word of flags (which pass's domains it lies in, which windows) in registers, so a
// step is straight-line code on LDS: ~12 us at C192 x 76 levels (the first version, a literal walk through memory: 68 us).
// Memory form (larger tiles; PACE_DDH_MEM=1): the planes are the two scratch fields, uc / vc the caller's arrays.
#define DDH_M 2                     // band depth inside the tile's edge, LDS form (the memory form keeps 4)
#define DDH_MM 4
#define DDH_NT 1024
#define DDH_PP 4                    // points per thread (LDS form)
#define DDH_SLOTS (DDH_PP * DDH_NT)
struct DdhBand {  // the band as four rectangles: south rows, north rows, west and east columns between them
  int ni, nj, lo, hi, nsouth, nnorth, nmid, nwest, neast, total;
  __host__ __device__ DdhBand(const Geo& g, int depth) {
    ni = g.ni, nj = g.nj;
    lo = g.is + depth, hi = g.ie + 1 - depth;
    const bool all = hi - lo < 1;  // a tile too small to have an inside: the whole plane
    nsouth = all ? g.nj : lo, nnorth = all ? 0 : g.nj - 1 - hi, nmid = all ? 0 : hi - lo + 1;
    nwest = all ? 0 : lo, neast = all ? 0 : g.ni - 1 - hi;
    total = (nsouth + nnorth) * g.ni + nmid * (nwest + neast);
  }
  __device__ __forceinline__ void point(int p, int& i, int& j) const {
    if (p < nsouth * ni) {
      j = p / ni, i = p - j * ni;
    } else if (p < (nsouth + nnorth) * ni) {
      p -= nsouth * ni;
      const int r = p / ni;
      j = hi + 1 + r, i = p - r * ni;
    } else {
      p -= (nsouth + nnorth) * ni;
      const int w = nwest + neast, r = p / w, c = p - r * w;
      j = lo + r, i = c < nwest ? c : hi + 1 + (c - nwest);
    }
  }
  // the slot of (i, j); `none` outside the band (or the storage)
  __device__ __forceinline__ int slot(int i, int j, int none) const {
    if (i < 0 || i >= ni || j < 0 || j >= nj) return none;
    if (j < nsouth) return j * ni + i;
    if (j > hi) return (nsouth + (j - hi - 1)) * ni + i;
    if (i < nwest) return (nsouth + nnorth) * ni + (j - lo) * (nwest + neast) + i;
    if (i > hi) return (nsouth + nnorth) * ni + (j - lo) * (nwest + neast) + nwest + (i - hi - 1);
    return none;
  }
};
// fill_corners_dgrid(x = vc, y = uc, mysign = -1), entry e of 72 (corners.py:987-1151; oracle/corner_ops.py): destination, source, sign;
// e < 36: vc <- sgn * uc, else uc <- sgn * vc.  Every source lies outside the corner blocks and outside the windows.
__device__ __forceinline__ void ddh_dgrid_entry(const Geo& g, int e72, int& di, int& dj, int& si, int& sj_, double& sgn) {
  const int e = e72 % 36, q = e / 9, a = 1 + (e % 9) % 3, b = 1 + (e % 9) / 3;
  if (e72 < 36) {
    if (q == 0) di = g.is - a, dj = g.js - b, sgn = -1.0, si = g.is - b, sj_ = g.js + a - 1;                   // SW
    else if (q == 1) di = g.ie + a, dj = g.je + 1 + b, sgn = -1.0, si = g.ie + 1 + b, sj_ = g.je + 1 - a;      // NE
    else if (q == 2) di = g.is - a, dj = g.je + 1 + b, sgn = 1.0, si = g.is - b, sj_ = g.je + 1 - a;           // NW
    else di = g.ie + a, dj = g.js - b, sgn = 1.0, si = g.ie + 1 + b, sj_ = g.js + a - 1;                       // SE
  } else {
    if (q == 0) di = g.is - a, dj = g.js - b, sgn = -1.0, si = g.is + b - 1, sj_ = g.js - a;                   // SW
    else if (q == 1) di = g.ie + 1 + a, dj = g.je + b, sgn = -1.0, si = g.ie + 1 - b, sj_ = g.je + 1 + a;      // NE
    else if (q == 2) di = g.is - a, dj = g.je + b, sgn = 1.0, si = g.is + b - 1, sj_ = g.je + 1 + a;           // NW
    else di = g.ie + 1 + a, dj = g.js - b, sgn = 1.0, si = g.ie + 1 - b, sj_ = g.js - a;                       // SE
  }
}

// what a thread of k_divdamp_halo_state_lds knows of a band point -- the same at every level and in every call on a geometry:
// [0] its offset in a plane, [1..4] the LDS places of its east / west / north / south neighbours, [5..8] those of the operands of
// vc_from_divg / uc_from_divg with the corner fills applied as index maps (x fill: east, centre; y fill: north, centre), [9] flags
enum { DDH_F_VD = 0, DDH_F_UD = 3, DDH_F_DD = 6, DDH_F_VW = 9, DDH_F_VWM = 10, DDH_F_UW = 11, DDH_F_UWM = 12, DDH_F_CS = 13, DDH_F_CN = 14,
       DDH_F_OK = 15, DDH_F_OWN = 16 };
#define DDH_TAB 10
__device__ __forceinline__ void ddh_point_table(const Geo& g, const DdhBand& band, int p, int* t) {
  constexpr int NONE = DDH_SLOTS;
  int i = 0, j = 0;
  const bool ok = p < band.total;
  if (ok) band.point(p, i, j);
  t[0] = (int)IDX2(g, i, j);
  t[1] = band.slot(i + 1, j, NONE), t[2] = band.slot(i - 1, j, NONE), t[3] = band.slot(i, j + 1, NONE), t[4] = band.slot(i, j - 1, NONE);
  int a = i + 1, b = j;
  remap_bgrid_x(g, a, b);
  t[5] = band.slot(a, b, NONE);
  a = i, b = j;
  remap_bgrid_x(g, a, b);
  t[6] = band.slot(a, b, NONE);
  a = i, b = j + 1;
  remap_bgrid_y(g, a, b);
  t[7] = band.slot(a, b, NONE);
  a = i, b = j;
  remap_bgrid_y(g, a, b);
  t[8] = band.slot(a, b, NONE);
  auto in_uwin = [&](int x, int y) { return x >= g.is && x <= g.ie + 1 && y >= g.js && y <= g.je; };
  auto in_vwin = [&](int x, int y) { return x >= g.is && x <= g.ie && y >= g.js && y <= g.je + 1; };
  unsigned f = 0;
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) {
    f |= (unsigned)(i >= g.is - nt - 1 && i <= g.ie + nt + 1 && j >= g.js - nt && j <= g.je + nt + 1) << (DDH_F_VD + nt);
    f |= (unsigned)(i >= g.is - nt && i <= g.ie + nt + 1 && j >= g.js - nt - 1 && j <= g.je + nt + 1) << (DDH_F_UD + nt);
    f |= (unsigned)(i >= g.is - nt && i <= g.ie + nt + 1 && j >= g.js - nt && j <= g.je + nt + 1) << (DDH_F_DD + nt);
  }
  const bool ic = (i == g.is || i == g.ie + 1);
  const bool own = i >= g.is && i <= g.ie + 1 && j >= g.js && j <= g.je + 1;
  f |= (unsigned)in_vwin(i, j) << DDH_F_VW | (unsigned)in_vwin(i - 1, j) << DDH_F_VWM | (unsigned)in_uwin(i, j) << DDH_F_UW |
       (unsigned)in_uwin(i, j - 1) << DDH_F_UWM;
  f |= (unsigned)(ic && j == g.js) << DDH_F_CS | (unsigned)(ic && j == g.je + 1) << DDH_F_CN | (unsigned)ok << DDH_F_OK | (unsigned)own << DDH_F_OWN;
  t[9] = (int)f;
}
// the tables of a geometry, [DDH_TAB][DDH_SLOTS] ints, built once per object (dsw_prepare) into the caller's workspace: the kernel
// below then loads a thread's 4 x 10 words instead of deriving them at every level (~60 % of its 37 us)
__global__ void __launch_bounds__(DDH_NT) k_ddh_tables(Geo g, int* __restrict__ tab) {
  const DdhBand band(g, DDH_M);
  for (int q = 0; q < DDH_PP; ++q) {
    const int p = (int)threadIdx.x + DDH_NT * q;
    int t[DDH_TAB];
    ddh_point_table(g, band, p, t);
#pragma unroll
    for (int w = 0; w < DDH_TAB; ++w) tab[w * DDH_SLOTS + p] = t[w];
  }
}

__global__ void __launch_bounds__(DDH_NT)
k_divdamp_halo_state_lds(Geo g, Met m, const real* __restrict__ delpc, real* divg_d, real* __restrict__ uc, real* __restrict__ vc, int k0,
                         int nord, const int* __restrict__ tab) {
  __shared__ double sda[DDH_SLOTS + 1], sdb[DDH_SLOTS + 1], su[DDH_SLOTS + 1], sv[DDH_SLOTS + 1];  // (+ 1: the place of "no such point")
  constexpr int NONE = DDH_SLOTS;
  const int kk = (int)blockIdx.x + k0;
  const long kb = (long)kk * g.sk;
  const int tid = (int)threadIdx.x;
  const DdhBand band(g, DDH_M);
  // what is stored in a field is rounded to the field's type (float32 build), as the passes through memory do
  auto stored = [](double x) { return (double)(real)x; };
  enum { F_VD = DDH_F_VD, F_UD = DDH_F_UD, F_DD = DDH_F_DD, F_VW = DDH_F_VW, F_VWM = DDH_F_VWM, F_UW = DDH_F_UW, F_UWM = DDH_F_UWM,
         F_CS = DDH_F_CS, F_CN = DDH_F_CN, F_OK = DDH_F_OK, F_OWN = DDH_F_OWN };
  int c2_[DDH_PP], sE_[DDH_PP], sW_[DDH_PP], sN_[DDH_PP], sS_[DDH_PP], xE_[DDH_PP], xC_[DDH_PP], yN_[DDH_PP], yC_[DDH_PP];
  unsigned fl_[DDH_PP];
  double gu0_[DDH_PP], gum_[DDH_PP], gv0_[DDH_PP], gvm_[DDH_PP], ra_[DDH_PP];
  if (tid == 0) sda[NONE] = sdb[NONE] = su[NONE] = sv[NONE] = 0.0;
#pragma unroll
  for (int t = 0; t < DDH_PP; ++t) {
    const int p = tid + DDH_NT * t;
    int w[DDH_TAB];
    if (tab != nullptr) {  // (block-uniform)
#pragma unroll
      for (int q = 0; q < DDH_TAB; ++q) w[q] = tab[q * DDH_SLOTS + p];
    } else {
      ddh_point_table(g, band, p, w);
    }
    const int c2 = w[0];
    c2_[t] = c2, sE_[t] = w[1], sW_[t] = w[2], sN_[t] = w[3], sS_[t] = w[4], xE_[t] = w[5], xC_[t] = w[6], yN_[t] = w[7], yC_[t] = w[8];
    const unsigned f = (unsigned)w[9];
    fl_[t] = f;
    const bool ok = (f >> F_OK) & 1u, own = (f >> F_OWN) & 1u;
    gu0_[t] = m.divg_u[c2], gv0_[t] = m.divg_v[c2], ra_[t] = m.rarea_c[c2];
    // (the neighbours' values: used only where the flags say the neighbour lies in a window; any valid address otherwise)
    gum_[t] = m.divg_u[c2 > 0 ? c2 - 1 : c2], gvm_[t] = m.divg_v[c2 >= g.sj ? c2 - g.sj : c2];
    if (ok) {
      // the divergence before the passes: delpc at the tile's points (copy_computeplus), the caller's divg_d in the halo (written by
      // this kernel at its very end only); uc / vc: what the arrays hold (inside the windows: never read from here)
      sda[p] = (double)(own ? delpc : (const real*)divg_d)[kb + c2];
      su[p] = (double)uc[kb + c2];
      sv[p] = (double)vc[kb + c2];
    }
  }
  __syncthreads();
  double* src = sda;
  double* dst = sdb;
  for (int it = 0; it < nord; ++it) {
    const int nt = nord - (it + 1);
    const bool fillc = it + 1 != nord;
    // ---- A: vc_from_divg (corners filled in x), uc_from_divg (corners filled in y), outside the windows
#pragma unroll
    for (int t = 0; t < DDH_PP; ++t) {
      const unsigned f = fl_[t];
      const int p = tid + DDH_NT * t;
      if (!((f >> F_OK) & 1u)) continue;
      if (((f >> (F_VD + nt)) & 1u) && !((f >> F_VW) & 1u)) sv[p] = stored((src[fillc ? xE_[t] : sE_[t]] - src[fillc ? xC_[t] : p]) * gu0_[t]);
      if (((f >> (F_UD + nt)) & 1u) && !((f >> F_UW) & 1u)) su[p] = stored((src[fillc ? yN_[t] : sN_[t]] - src[fillc ? yC_[t] : p]) * gv0_[t]);
    }
    __syncthreads();
    // ---- B: fill_corners_dgrid
    if (fillc) {
      if (tid < 72) {
        int di, dj, si, sj_;
        double sgn;
        ddh_dgrid_entry(g, tid, di, dj, si, sj_, sgn);
        const int d_ = band.slot(di, dj, NONE), s_ = band.slot(si, sj_, NONE);
        if (tid < 36) sv[d_] = sgn * su[s_];
        else su[d_] = sgn * sv[s_];
      }
      __syncthreads();
    }
    // ---- C: redo_divg_d on its domain; elsewhere the plane carries the state on (corner points: the y fill)
#pragma unroll
    for (int t = 0; t < DDH_PP; ++t) {
      const unsigned f = fl_[t];
      const int p = tid + DDH_NT * t;
      if (!((f >> F_OK) & 1u)) continue;
      double d;
      if ((f >> (F_DD + nt)) & 1u) {
        const double dc = src[p];
        // (the band's innermost points read one point past it: in the first pass that is the divergence before the passes -- delpc,
        // a point of the tile --, afterwards nothing the halo depends on)
        auto nb = [&](int sl, int off) -> double { return (it == 0 && sl == NONE) ? (double)delpc[kb + c2_[t] + off] : src[sl]; };
        const double dS = nb(sS_[t], -g.sj), dN = nb(sN_[t], g.sj), dW = nb(sW_[t], -1), dE = nb(sE_[t], 1);
        const double ucm = ((f >> F_UWM) & 1u) ? (dc - dS) * gvm_[t] : su[sS_[t]];
        const double uc0 = ((f >> F_UW) & 1u) ? (dN - dc) * gv0_[t] : su[p];
        const double vcm = ((f >> F_VWM) & 1u) ? (dc - dW) * gum_[t] : sv[sW_[t]];
        const double vc0 = ((f >> F_VW) & 1u) ? (dE - dc) * gu0_[t] : sv[p];
        d = ucm - uc0 + vcm - vc0;
        if ((f >> F_CS) & 1u) d = d - ucm;
        if ((f >> F_CN) & 1u) d = d + uc0;
        d = d * ra_[t];
      } else {
        d = src[fillc ? yC_[t] : p];
      }
      dst[p] = stored(d);
    }
    __syncthreads();
    double* const tmp = src;
    src = dst;
    dst = tmp;
  }
  // the halo of the caller's divergence, uc and vc
#pragma unroll
  for (int t = 0; t < DDH_PP; ++t) {
    const unsigned f = fl_[t];
    const int p = tid + DDH_NT * t;
    if (!((f >> F_OK) & 1u)) continue;
    if (!((f >> F_OWN) & 1u)) divg_d[kb + c2_[t]] = (real)src[p];
    if (!((f >> F_UW) & 1u)) uc[kb + c2_[t]] = (real)su[p];
    if (!((f >> F_VW) & 1u)) vc[kb + c2_[t]] = (real)sv[p];
  }
}

__global__ void __launch_bounds__(DDH_NT)
k_divdamp_halo_state_mem(Geo g, Met m, const real* __restrict__ delpc, real* __restrict__ d0, real* __restrict__ d1, real* divg_d,
                         real* __restrict__ uc, real* __restrict__ vc, int k0, int nord) {
  const int kk = (int)blockIdx.x + k0;
  const long kb = (long)kk * g.sk;
  const int tid = (int)threadIdx.x;
  const DdhBand band(g, DDH_MM);
  const int total = band.total;
  real* U = uc + kb;
  real* V = vc + kb;
  // the divergence before the passes: delpc at the tile's points (copy_computeplus), the caller's divg_d in the halo -- which this
  // kernel writes at its very end only
  auto orig = [&](int i, int j) -> double {
    const bool own = i >= g.is && i <= g.ie + 1 && j >= g.js && j <= g.je + 1;
    return (double)(own ? delpc : (const real*)divg_d)[kb + IDX2(g, i, j)];
  };
  // the two planes of the passes: plane `cur` holds the state before a pass (first pass: the original), the other takes the state
  // after it.  The band's innermost points go wrong by one point per pass (they read past the band); nothing the halo depends on
  // reaches them.
  real* Gm[2] = {d0 + kb, d1 + kb};
  int cur = 0;
  bool first = true;
  auto rd = [&](int i, int j) -> double { return first ? orig(i, j) : (double)Gm[cur][IDX2(g, i, j)]; };
  auto in_uwin = [&](int i, int j) { return i >= g.is && i <= g.ie + 1 && j >= g.js && j <= g.je; };
  auto in_vwin = [&](int i, int j) { return i >= g.is && i <= g.ie && j >= g.js && j <= g.je + 1; };
  for (int it = 0; it < nord; ++it) {
    const int nt = nord - (it + 1);
    const bool fillc = it + 1 != nord;
    auto sx = [&](int i, int j) -> double {
      if (fillc) remap_bgrid_x(g, i, j);
      return rd(i, j);
    };
    auto sy = [&](int i, int j) -> double {
      if (fillc) remap_bgrid_y(g, i, j);
      return rd(i, j);
    };
    // ---- A: vc_from_divg on [is-nt-1, ie+nt+1] x [js-nt, je+nt+1], uc_from_divg on [is-nt, ie+nt+1] x [js-nt-1, je+nt+1]
    for (int p = tid; p < total; p += DDH_NT) {
      int i, j;
      band.point(p, i, j);
      const long c2 = IDX2(g, i, j);
      if (i >= g.is - nt - 1 && i <= g.ie + nt + 1 && j >= g.js - nt && j <= g.je + nt + 1 && !in_vwin(i, j))
        V[c2] = (real)((sx(i + 1, j) - sx(i, j)) * m.divg_u[c2]);
      if (i >= g.is - nt && i <= g.ie + nt + 1 && j >= g.js - nt - 1 && j <= g.je + nt + 1 && !in_uwin(i, j))
        U[c2] = (real)((sy(i, j + 1) - sy(i, j)) * m.divg_v[c2]);
    }
    __syncthreads();
    // ---- B: fill_corners_dgrid(x = vc, y = uc, mysign = -1)
    if (fillc && tid < 72) {
      int di, dj, si, sj_;
      double sgn;
      ddh_dgrid_entry(g, tid, di, dj, si, sj_, sgn);
      if (tid < 36) V[IDX2(g, di, dj)] = (real)(sgn * (double)U[IDX2(g, si, sj_)]);
      else U[IDX2(g, di, dj)] = (real)(sgn * (double)V[IDX2(g, si, sj_)]);
    }
    __syncthreads();
    // ---- C: redo_divg_d on [is-nt, ie+nt+1]^2; elsewhere the plane carries the state on (corner points: the y fill)
    for (int p = tid; p < total; p += DDH_NT) {
      int i, j;
      band.point(p, i, j);
      const long c2 = IDX2(g, i, j);
      double d;
      if (i >= g.is - nt && i <= g.ie + nt + 1 && j >= g.js - nt && j <= g.je + nt + 1) {
        const double dc = rd(i, j);
        const double ucm = in_uwin(i, j - 1) ? (dc - rd(i, j - 1)) * m.divg_v[c2 - g.sj] : (double)U[c2 - g.sj];
        const double uc0 = in_uwin(i, j) ? (rd(i, j + 1) - dc) * m.divg_v[c2] : (double)U[c2];
        const double vcm = in_vwin(i - 1, j) ? (dc - rd(i - 1, j)) * m.divg_u[c2 - 1] : (double)V[c2 - 1];
        const double vc0 = in_vwin(i, j) ? (rd(i + 1, j) - dc) * m.divg_u[c2] : (double)V[c2];
        d = ucm - uc0 + vcm - vc0;
        const bool ic = (i == g.is || i == g.ie + 1);
        if (ic && j == g.js) d = d - ucm;
        if (ic && j == g.je + 1) d = d + uc0;
        d = d * m.rarea_c[c2];
      } else {
        d = sy(i, j);
      }
      Gm[cur ^ 1][c2] = (real)d;
    }
    __syncthreads();
    cur ^= 1;
    first = false;
  }
  // the halo of the caller's divergence
  for (int p = tid; p < total; p += DDH_NT) {
    int i, j;
    band.point(p, i, j);
    if (i >= g.is && i <= g.ie + 1 && j >= g.js && j <= g.je + 1) continue;
    divg_d[kb + IDX2(g, i, j)] = Gm[cur][IDX2(g, i, j)];
  }
}

// DivergenceDamping.__call__ (divergence_damping.py:482-632): second-order damping on the levels above `kstart` (the sponge
// layers, nord = 0 there), `nonzero_nord` iterations of the divergence of the gradient of the divergence below, then
// a2b_ord4 of the relative vorticity, the Smagorinsky term and the damped vorticity; ke += damping.  uc, vc and divg_d end as
// the reference leaves them (the last iteration's work values on the staggered compute windows).  da / db: two scratch fields.
int launch_divergence_damping(const Geo& g, const Met& m, const real* u, const real* v, const real* va, real* vort_b,
                              const real* ua, real* divg_d, real* vc, real* uc, real* delpc, real* ke,
                              const real* rel_vort_agrid, double dt, const real* d2_bg_dev, int kstart, int nonzero_nord,
                              double dddmp, double d4_bg, real* da, real* db, hipStream_t st, bool skip_dead, bool ke_by_consumer,
                              const int* ddh_tab) {
  // ke_by_consumer (with skip_dead only): `ke += damped vorticity` is left to the kernel that reads both (the fused scalar + wind
  // kernel forms ke + vort_b, the same single addition): this operator then neither reads nor writes ke
  const int nk = g.nk;
  const int nhigh = nk - kstart;
  const bool fused = nhigh > 0 && !legacy_divergence_damping();
  skip_dead = skip_dead && fused;
  if (ke_by_consumer && !skip_dead) return PACE_ERR_ARG;
  if (ke_by_consumer) ke = nullptr;
  if (fused && !skip_dead) {
    // sponge levels + delpc = divg_d below them (copy_computeplus :578; whole planes, so that the fused kernel can take its
    // footprint from delpc), one launch
    hipLaunchKernelGGL(k_divdamp_low_and_copy, plane_grid(g, nk), dim3(256), 0, st, g, m, u, v, ua, va, uc, vc, delpc, vort_b, ke,
                       d2_bg_dev, dddmp, dt, divg_d, kstart);
  } else if (fused) {
    // the work fields are dead after d_sw (PACE_DSW_SKIP_DEAD_OUTPUTS): no copy -- the fused kernel reads the divergence where it
    // is and writes neither it nor uc / vc -- and the sponge levels are extra workgroups of the fused launch
  } else if (kstart > 0) {
    hipLaunchKernelGGL(k_divdamp_low, plane_grid(g, kstart), dim3(256), 0, st, g, m, u, v, ua, va, uc, vc, delpc, vort_b, ke,
                       d2_bg_dev, dddmp, dt);
  }
  if (fused) {
    const double dd8 = pow(m.da_min_c * d4_bg, (double)(nonzero_nord + 1));
    const int ntx = (g.n + 1 + DD_TI - 1) / DD_TI, nty = (g.n + 1 + DD_TJ - 1) / DD_TJ;
    DdSponge sp{u, v, ua, va, uc, vc, delpc, dt, 0, 0, 0, 1, 1, 1, 1};
    if (skip_dead && kstart > 0) {
      sp.nlev = kstart;
      sp.nblocks = kstart * (((g.n + 1) * (g.n + 1) + DD_NT - 1) / DD_NT);
    }
    if (g.n >= 8) {
      // chunks of at most DD_STRIP points along the edge (footprint (len + 6) x 8 <= a tile's: the strips share its two LDS planes)
      const int nrow = g.n + 1, ncol = g.n - 3;
      sp.nch_row = (nrow + DD_STRIP - 1) / DD_STRIP;
      sp.len_row = (nrow + sp.nch_row - 1) / sp.nch_row;
      sp.nch_col = (ncol + DD_STRIP - 1) / DD_STRIP;
      sp.len_col = (ncol + sp.nch_col - 1) / sp.nch_col;
      sp.nstrips = 2 * (sp.nch_row + sp.nch_col) * nhigh;
    }
    hipLaunchKernelGGL(k_divdamp_fused, dim3((unsigned)(sp.nblocks + sp.nstrips + ntx * nty * nhigh)), dim3(DD_NT), 0, st, g, m,
                       rel_vort_agrid, skip_dead ? divg_d : delpc, divg_d, vort_b, ke, uc, vc, d2_bg_dev, dddmp, dd8, fabs(dt), kstart,
                       nonzero_nord, ntx, ntx * nty, skip_dead ? 0 : 1, sp);
    // the full contract: the halo of divg_d, uc, vc as the reference's in-place passes leave it
    if (!skip_dead && nonzero_nord > 0) {
      const bool mem = getenv("PACE_DDH_MEM") != nullptr;  // (tests: the memory form on a small tile; read at every call)
      if (DdhBand(g, DDH_M).total <= DDH_SLOTS && !mem)
        hipLaunchKernelGGL(k_divdamp_halo_state_lds, dim3((unsigned)nhigh), dim3(DDH_NT), 0, st, g, m, delpc, divg_d, uc, vc, kstart, nonzero_nord,
                           ddh_tab);
      else
        hipLaunchKernelGGL(k_divdamp_halo_state_mem, dim3((unsigned)nhigh), dim3(DDH_NT), 0, st, g, m, delpc, da, db, divg_d, uc, vc, kstart,
                           nonzero_nord);
    }
  } else if (nhigh > 0) {
    const real* src = divg_d;
    real* bufs[2] = {da, db};
    for (int n = 0; n < nonzero_nord; ++n) {
      const int nt = nonzero_nord - (n + 1);
      const int fill = (n + 1 != nonzero_nord) ? 1 : 0;
      real* dst = bufs[n & 1];
      const bool last = (n + 1 == nonzero_nord);  // nt == 0: its region is exactly the (n+1) x (n+1) corner points
      launch_divdamp_iter(g, m, src, dst, kstart, nhigh, nt, fill, last ? uc : nullptr, last ? vc : nullptr, st);
      src = dst;
    }
    const double dd8 = pow(m.da_min_c * d4_bg, (double)(nonzero_nord + 1));
    const Regions r = a2b_regions(g);
    hipLaunchKernelGGL(k_divdamp_high_final, regions_grid(r, nhigh), dim3(64, 4), 0, st, g, m, rel_vort_agrid, divg_d, divg_d,
                       delpc, src, vort_b, ke, d2_bg_dev, dddmp, dd8, fabs(dt), kstart, r);
  }
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}

struct DswWork {
  real *ut, *vt, *fx, *fy, *gx, *gy, *fx2, *fy2, *dw, *heat_s, *ke, *wk, *abs_vort, *vort_b, *ut2, *vt2, *da, *db, *fyv, *umid, *vmid, *wtmp;
  real* kcol;  // device copy of the column namelist: 12 arrays of (nk+1)
  int* ddh;    // the halo-state kernel's tables of this geometry (k_ddh_tables), or null where its LDS form does not apply
};
#define DSW_NFIELDS 22

int64_t dsw_workspace_bytes(const Geo& g) {
  const int64_t field = (int64_t)g.sk * (g.nk + 1) * (int64_t)sizeof(real);
  return field * DSW_NFIELDS + 16 * (int64_t)(g.nk + 1) * (int64_t)sizeof(real) + 256 + (int64_t)sizeof(int) * DDH_TAB * DDH_SLOTS;
}

static DswWork carve(const Geo& g, void* ws) {
  DswWork w;
  real* p = (real*)ws;
  const long field = g.sk * (g.nk + 1);
  real** f = &w.ut;
  for (int n = 0; n < DSW_NFIELDS; ++n) f[n] = p + (long)n * field;
  w.kcol = p + (long)DSW_NFIELDS * field;
  // (behind the column block: 16 (nk + 1) elements + padding to 256 bytes)
  char* after = (char*)(w.kcol + 16 * (long)(g.nk + 1));
  after += (256 - ((uintptr_t)after & 255)) & 255;
  w.ddh = (DdhBand(g, DDH_M).total <= DDH_SLOTS) ? (int*)after : nullptr;
  return w;
}

#ifdef PACE_EMU
#include <cstring>
static void upload(real* dst, const real* src, size_t n, hipStream_t) { memcpy(dst, src, n * sizeof(real)); }
#else
static void upload(real* dst, const real* src, size_t n, hipStream_t st) {
  (void)hipMemcpyAsync(dst, src, n * sizeof(real), hipMemcpyHostToDevice, st);
  (void)hipStreamSynchronize(st);
}
#endif

#define NCOL 13
// Upload the column namelist once per object (the reference derives these in __init__, d_sw.py:785,924-933).
int dsw_prepare(const Geo& g, const pace_column_t* col, void* ws, hipStream_t st) {
  DswWork W = carve(g, ws);
  const int K = g.nk + 1;
  std::vector<real> h((size_t)NCOL * K, (real)0.0);
  const double* src[NCOL] = {col->nord_v, col->nord_w, col->nord_t, col->damp_vt, col->damp_w, col->damp_t, col->d2_divg,
                             col->d_con,  col->ke_bg,  col->fac_vt, col->fac_t,   col->fac_vt_c, col->fac_w_c};
  for (int a = 0; a < NCOL; ++a)
    for (int k = 0; k < g.nk; ++k) h[(size_t)a * K + k] = (real)src[a][k];
  upload(W.kcol, h.data(), h.size(), st);
  if (W.ddh != nullptr) hipLaunchKernelGGL(k_ddh_tables, dim3(1), dim3(DDH_NT), 0, st, g, W.ddh);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}

// phases (bit mask): 1 = flux preparation (fxadv), 2 = transport of delp, w, q_con, pt (everything riem_solver3 /
// updatedzd depend on), 4 = winds A (kinetic energy, vorticity, divergence damping, vorticity transport, u/v from ke,
// vorticity damping fluxes), 8 = winds B (dissipative heating, final u/v update).  15 = the whole of d_sw.
// Dependencies: 2 and 4 need only 1 (and use disjoint workspace fields); 8 needs 2 and 4.  A caller may therefore run
// 4 (then 8) on a second stream concurrently with 2 and with whatever follows d_sw on the first stream -- the vertical
// solver -- see pace_amd/fv3core/stencils/d_sw.py.
int launch_d_sw(const Geo& g, const Met& m, const pace_column_t* col, const pace_dsw_config_t* cfg, void* ws,
                real* delpc, real* delp, real* pt, real* u, real* v, real* w, real* uc, real* vc,
                const real* ua, const real* va, real* divgd, real* mfx, real* mfy, real* cx, real* cy,
                real* crx, real* cry, real* xfx, real* yfx, real* q_con, const real* zh,
                real* heat_source, real* diss_est, double dt, int phases, hipStream_t st) {
  (void)zh;
  // optional separate outputs of the four transported scalars (pace_dsw_config_t): delp, pt, w, q_con
  real* scalar_outs[4] = {cfg->delp_out, cfg->pt_out, cfg->w_out, cfg->q_con_out};
  const bool pingpong = scalar_outs[0] != nullptr;
  if (pingpong && !dsw_pingpong_supported(g, cfg)) return PACE_ERR_UNSUPPORTED;
  const int nk = g.nk;
  DswWork W = carve(g, ws);
  const int K = nk + 1;
  real* kc = W.kcol;  // filled by dsw_prepare
  real *d_nord_v = kc, *d_nord_w = kc + K, *d_nord_t = kc + 2 * K, *d_damp_vt_c = kc + 3 * K, *d_damp_w_c = kc + 4 * K,
         *d_d2 = kc + 6 * K, *d_dcon = kc + 7 * K, *d_kebg = kc + 8 * K, *d_dampfac_vt = kc + 9 * K,
         *d_dampfac_t = kc + 10 * K, *d_dampfac_vt_c = kc + 11 * K, *d_dampfac_w_c = kc + 12 * K;
  int nmax_v = 0, nmax_w = 0, nmax_t = 0, kstart = 0, nonzero_nord = cfg->nord;
  bool found = false;
  for (int k = 0; k < nk; ++k) {
    nmax_v = std::max(nmax_v, (int)col->nord_v[k]);
    nmax_w = std::max(nmax_w, (int)col->nord_w[k]);
    nmax_t = std::max(nmax_t, (int)col->nord_t[k]);
    if (!found && col->nord[k] > 0) {
      found = true;
      kstart = k;
      nonzero_nord = (int)col->nord[k];
    }
  }
  int rc;
  // Where the scalar-phase kernel can take the winds (the production tilings, one order for all transports, the 512-thread form)
  // and the call asks for scalars and winds together, the vorticity transport, the wind update and the dissipative heating are
  // the FIFTH PASS of that kernel (fvt_core.h): k_fvt<.., 0, 0> and k_heat_source, and the fields between them, disappear.
  // The kinetic energy, the vorticity and the divergence damping then have to run BEFORE the scalars.
  static const bool separate_winds = getenv("PACE_DSW_SEPARATE_WINDS") != nullptr;  // (A/B measurements)
  const bool lean_scalars = cfg->hord_dp == cfg->hord_vt && cfg->hord_dp == cfg->hord_tm && transport_lean_covers(g, cfg->hord_dp);
  // (phases 256, a measurement aid: that kernel ALONE, on the kinetic energy / vorticity / damped vorticity a previous call left in
  // the workspace)
  const bool winds_in_scalars = lean_scalars && (((phases & 2) && (phases & 4) && (phases & 8)) || (phases & 256)) && dsw_scalars_take_winds() &&
                                !separate_winds && nmax_v <= 2 && nmax_w <= 2 && nmax_t <= 2 && ((uintptr_t)W.wk & 15) == 0;
  // (separate wind outputs exist in that form only; a call that runs neither the scalars nor the heating does not touch them)
  if ((cfg->u_out != nullptr) && !winds_in_scalars && (phases & (2 | 8))) return PACE_ERR_UNSUPPORTED;
  const bool skip_dead = (cfg->flags & PACE_DSW_SKIP_DEAD_OUTPUTS) != 0;
  // ... and where the work fields are not asked for either, the divergence damping leaves `ke += damped vorticity` to that kernel
  // (it holds both at the tile's B-grid points): the damping then does not touch ke, 47 MB less, and it no longer depends on the
  // kinetic-energy kernel
  // (the same predicate as launch_divergence_damping's `fused`: with the legacy A/B switch the damping adds to ke itself)
  const bool ke_by_consumer = winds_in_scalars && skip_dead && nk - kstart > 0 && !legacy_divergence_damping();
  // (separate wind outputs: the halo of the output buffers is a copy of the inputs' -- taken along by the flux preparation's frame
  // workgroups when this call also runs the wind phase, else by a launch of its own below)
  FxWindHalo wind_halo{u, v, cfg->u_out, cfg->v_out, false};
  FxWindHalo* const wh = (winds_in_scalars && cfg->u_out != nullptr && (phases & (4 | 64))) ? &wind_halo : nullptr;
  if (phases & 1) {
  if ((rc = launch_fxadv(g, m, uc, vc, crx, cry, xfx, yfx, W.ut, W.vt, dt, cx, cy, st, 0, 0, wh))) return rc;
  } else {  // the same in two parts around a halo exchange: 16 = interior box, 32 = the rest
    if ((phases & 16) && (rc = launch_fxadv(g, m, uc, vc, crx, cry, xfx, yfx, W.ut, W.vt, dt, cx, cy, st, 1))) return rc;
    if ((phases & 32) && (rc = launch_fxadv(g, m, uc, vc, crx, cry, xfx, yfx, W.ut, W.vt, dt, cx, cy, st, 2, 0, wh))) return rc;
  }
  auto scalar_phase = [&]() -> int {
    // The production tilings with one order for all four: ONE kernel (k_fvt.hip launch_dsw_scalars_lean) takes a tile through
    // delp, w, q_con, pt and the division by the new delp; its results go to the caller's separate outputs, or to workspace
    // fields that are copied back (the in-place contract of pace_d_sw).
    bool fused = false;
    if (lean_scalars) {
      real* ws_outs[4] = {W.gx, W.fx2, W.wtmp, W.gy};
      DswWinds wd{};
      if (winds_in_scalars) {
        wd.rel_vort = W.wk, wd.u = u, wd.v = v, wd.ke = W.ke, wd.vort_b = W.vort_b, wd.heat_source = heat_source;
        wd.u_out = cfg->u_out ? cfg->u_out : W.umid, wd.v_out = cfg->v_out ? cfg->v_out : W.vmid;
        // (the halo of separate wind outputs: copied by k_copy_wind_halo when this call runs the wind phase, by the kernel's edge tiles when the
        // kernel is launched alone -- phases 256, a measurement aid)
        wd.copy_halo = cfg->u_out != nullptr && !(phases & (4 | 64)), wd.do_skeb = cfg->do_skeb, wd.d_con = cfg->d_con;
        wd.ke_plus_vort = ke_by_consumer;
      }
      rc = launch_dsw_scalars_lean(g, m, delp, pt, w, q_con, pingpong ? scalar_outs : ws_outs, crx, cry, xfx, yfx, mfx, mfy, W.dw,
                                   W.heat_s, diss_est, kc, cfg->hord_dp, nmax_v, nmax_w, nmax_t, dt, st, winds_in_scalars ? &wd : nullptr);
      if (rc == PACE_OK) {
        fused = true;
        if (!pingpong) hipLaunchKernelGGL(k_copy_scalars, patch_grid(g, nk), PATCH_BLOCK, 0, st, g, delp, pt, w, q_con, W.gx, W.fx2, W.wtmp, W.gy);
        if (winds_in_scalars && !cfg->u_out) hipLaunchKernelGGL(k_copy_winds, patch_grid(g, nk), PATCH_BLOCK, 0, st, g, u, v, W.umid, W.vmid);
      } else if (rc != PACE_ERR_UNSUPPORTED || winds_in_scalars) {
        return rc;
      }
    }
    if (!fused) {
    if (pingpong) return PACE_ERR_UNSUPPORTED;
    const int nl = nk;
    FvDamp dp{};
    // delp: transport + del-n damping of the mass fluxes -> fx, fy
    dp.damp_k = d_dampfac_vt; dp.nord_k = d_nord_v; dp.nmax = nmax_v; dp.mass_given = 0;
    dp.accx = mfx; dp.accy = mfy;  // flux_capacitor (d_sw.py:33-60); its Courant-number half sits in fxadv
    if ((rc = launch_transport(g, m, delp, crx, cry, xfx, yfx, W.fx, W.fy, nullptr, nullptr, cfg->hord_dp, nl, 1, 0, dp, st))) return rc;
    // w: transport with the mass fluxes, del-n damping fluxes -> heat_diss, flux-form update -> W.gx (= w*delp + F(w))
    dp = FvDamp{};
    dp.damp_k = d_dampfac_w_c; dp.nord_k = d_nord_w; dp.nmax = nmax_w; dp.mass_given = 0;
    dp.qout = W.gx; dp.amass = delp; dp.dw = W.dw; dp.heat_s = W.heat_s; dp.diss_est = diss_est;
    dp.damp_w_k = d_damp_w_c; dp.ke_bg_k = d_kebg; dp.dt = dt;
    const FvDamp dpw = dp;
    // q_con -> W.gy
    dp = FvDamp{};
    dp.damp_k = d_dampfac_t; dp.nord_k = d_nord_t; dp.nmax = nmax_t; dp.mass_given = 1; dp.mass = delp;
    dp.qout = W.gy; dp.amass = delp;
    // pt -> W.fx2
    FvDamp dp2 = dp;
    dp2.damp_k = d_dampfac_vt; dp2.nord_k = d_nord_v; dp2.nmax = nmax_v; dp2.qout = W.fx2;
    // General tilings with ord 6 for all three: ONE launch (k_fvtp2d_scalars3: a grid three tile planes high)
    bool done3 = false;
    if (!transport_lean_covers(g, 6) && cfg->hord_vt == 6 && cfg->hord_dp == 6 && cfg->hord_tm == 6) {
      rc = launch_transport_scalars3(g, m, w, q_con, pt, crx, cry, xfx, yfx, W.fx, W.fy, nl, dpw, dp, dp2, st);
      if (rc == PACE_OK) done3 = true;
      else if (rc != PACE_ERR_UNSUPPORTED) return rc;
    }
    if (!done3) {
      if ((rc = launch_transport(g, m, w, crx, cry, xfx, yfx, nullptr, nullptr, W.fx, W.fy, cfg->hord_vt, nl, 0, 2, dpw, st))) return rc;
      if ((rc = launch_transport(g, m, q_con, crx, cry, xfx, yfx, nullptr, nullptr, W.fx, W.fy, cfg->hord_dp, nl, 2, 1, dp, st))) return rc;
      if ((rc = launch_transport(g, m, pt, crx, cry, xfx, yfx, nullptr, nullptr, W.fx, W.fy, cfg->hord_tm, nl, 2, 1, dp2, st))) return rc;
    }
    hipLaunchKernelGGL(k_finish_scalars, patch_grid(g, nk), PATCH_BLOCK, 0, st, g, m, pt, delp, w, q_con, W.fx2, W.gx, W.gy, W.fx, W.fy, W.dw, d_damp_w_c);
    }
    // the full contract: the corner blocks of the four scalars as the transport's in-place corner copies leave them
    // (separate outputs: the scalar-phase kernel's corner tiles have written them with the halo, fvt_core.h place_footprint)
    if (!skip_dead && !pingpong) hipLaunchKernelGGL(k_corner_blocks_x, dim3((unsigned)nk), dim3(256), 0, st, g, delp, pt, w, q_con);
    return PACE_OK;
  };
  if ((phases & 2) && !winds_in_scalars && (rc = scalar_phase())) return rc;
  if (phases & (4 | 64)) {
  // winds A1: kinetic energy and relative vorticity (need only the flux preparation)
  // (two point kernels.  The vorticity inside the kinetic-energy kernel: no faster, x02; both from LDS tiles of u and v: twice as
  // slow, x06 -- profiles/r05_experiments)
  const Regions rke = (g.n >= 8) ? bgrid_regions(g, 3) : a2b_regions(g);
  if (cfg->hord_mt != 5 && cfg->hord_mt != 6) return PACE_ERR_UNSUPPORTED;
  const bool two_launches = getenv("PACE_KE_VORT_SPLIT") != nullptr;  // (read per call; A/B measurements, tests: round 5's two launches)
  if (!two_launches && PATCH_W == 64) {
    const int nbr = rke.first[rke.n];
    const dim3 pg = patch_grid(g, 1);
    const unsigned nbl = (unsigned)nbr + pg.x * pg.y;
    const char* ke_ch_env = getenv("PACE_KE_LEVELS");  // (read per call; A/B measurements, tests: 1 or 2)
    const int ke_ch = ke_ch_env ? (ke_ch_env[0] == '2' ? 2 : 1) : (nbl * (unsigned)nk >= KE_CH2_MIN_WGS ? 2 : 1);
#define KE_GO(MORD, CH)                                                                                                                   \
  hipLaunchKernelGGL((k_ke_vorticity<MORD, CH>), dim3(nbl, 1, (unsigned)((nk + CH - 1) / CH)), dim3(64, 4), 0, st, g, m, uc, vc, u, v, W.ut, W.vt, \
                     W.ke, dt, rke, W.wk, nbr)
    if (cfg->hord_mt == 5) {
      if (ke_ch == 2) KE_GO(5, 2); else KE_GO(5, 1);
    } else {
      if (ke_ch == 2) KE_GO(6, 2); else KE_GO(6, 1);
    }
#undef KE_GO
  } else {
  if (cfg->hord_mt == 5) {
    hipLaunchKernelGGL(k_kinetic_energy<5>, regions_grid(rke, nk), dim3(64, 4), 0, st, g, m, uc, vc, u, v, W.ut, W.vt, W.ke, dt, rke, (real*)nullptr, rke.n);
  } else {
    hipLaunchKernelGGL(k_kinetic_energy<6>, regions_grid(rke, nk), dim3(64, 4), 0, st, g, m, uc, vc, u, v, W.ut, W.vt, W.ke, dt, rke, (real*)nullptr, rke.n);
  }
  hipLaunchKernelGGL(k_vorticity, patch_grid(g, nk), PATCH_BLOCK, 0, st, g, m, u, v, W.wk);
  }
  // (with separate wind outputs the halo of the output buffers is copied here, by a launch of its own over the frame of the plane)
  if (winds_in_scalars && cfg->u_out != nullptr && !wind_halo.done) {
    const int frame = wind_halo_points(g);
    hipLaunchKernelGGL(k_copy_wind_halo, dim3((unsigned)((frame + 255) / 256), (unsigned)nk), dim3(256), 0, st, g, u, v, cfg->u_out, cfg->v_out);
  }
  }
  if (phases & (4 | 128)) {
  // winds A2: divergence damping
  if (nonzero_nord < 0 || nonzero_nord > 3) return PACE_ERR_UNSUPPORTED;  // (as pace_divergence_damping: halo 3)
  if ((rc = launch_divergence_damping(g, m, u, v, va, W.vort_b, ua, divgd, vc, uc, delpc, W.ke, W.wk, dt, d_d2, kstart, nonzero_nord,
                                      cfg->dddmp, cfg->d4_bg, W.da, W.db, st, skip_dead, ke_by_consumer, W.ddh)))
    return rc;
  // vorticity transport
  // vorticity: transport of the absolute vorticity (wk + fC_agrid) -> W.fy2, W.fyv (own flux buffers: the mass fluxes
  // in W.fx / W.fy may still be in use by phase 2 on another stream) and the del-n damping fluxes of the relative
  // vorticity -> ut2, vt2 (DelnFluxNoSG, d_sw.py:1187-1195), one kernel
  if (!winds_in_scalars) {
    FvDamp dp{};
    dp.damp_k = d_dampfac_vt_c; dp.nord_k = d_nord_v; dp.nmax = nmax_v; dp.mass_given = 0;
    dp.fx2o = W.ut2; dp.fy2o = W.vt2; dp.add2d = m.fC_agrid;
    // ... and u_and_v_from_ke finished in the kernel's store phase: the vorticity fluxes never reach memory
    dp.u_upd = u; dp.v_upd = v; dp.ke = W.ke;
    dp.u_out = W.umid; dp.v_out = W.vmid;  // (read by k_heat_source, which writes the final winds to u, v)
    if ((rc = launch_transport(g, m, W.wk, crx, cry, xfx, yfx, W.fy2, W.fyv, nullptr, nullptr, cfg->hord_vt, nk, 0, 0, dp, st))) return rc;
  }
  }
  if ((phases & 256) && !winds_in_scalars) return PACE_ERR_UNSUPPORTED;
  if (winds_in_scalars && (rc = scalar_phase())) return rc;  // scalars + winds, after the kinetic energy and the divergence damping
  if ((phases & 8) && !winds_in_scalars) {
  hipLaunchKernelGGL(k_heat_source, patch_grid(g, nk), PATCH_BLOCK, 0, st, g, m, W.umid, W.vmid, W.vort_b, W.ut2, W.vt2, pingpong ? scalar_outs[0] : delp, W.heat_s, heat_source,
                     diss_est, d_dcon, cfg->d_con, cfg->do_skeb, u, v, d_damp_vt_c);
  }
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}
