// Host-side launch functions (one per reference class on the hot path).
#pragma once
#include "common.h"

// d_sw's separate wind outputs get the halo of the inputs (k_dsw.hip k_copy_wind_halo): a copy the frame workgroups of the one-launch
// form take along (their stage A waits for a round trip anyway); `done` tells the caller whether this launch took it
struct FxWindHalo {
  const real *u, *v;
  real *u_out, *v_out;
  bool done;
};
int launch_fxadv(const Geo& g, const Met& m, const real* uc, const real* vc, real* crx, real* cry,
                 real* xfx, real* yfx, real* ut, real* vt, double dt, real* cx_acc, real* cy_acc,
                 hipStream_t st, int part = 0, int contra_out = 0, FxWindHalo* wind_halo = nullptr);
// the frame of the plane outside the faces d_sw's winds are written on, point p of its level (k_copy_wind_halo's enumeration)
__device__ __forceinline__ void wind_halo_copy_point(const Geo& g, int p, int k, const real* __restrict__ u, const real* __restrict__ v,
                                                     real* __restrict__ u_out, real* __restrict__ v_out) {
  const int nsouth = g.js, nnorth = g.nj - 1 - g.je, nmid = g.je - g.js + 1, nwest = g.is, neast = g.ni - 1 - g.ie;
  const int rows = (nsouth + nnorth) * g.ni;
  int i, j;
  if (p < rows) {
    const int r = p / g.ni;
    i = p - r * g.ni, j = r < nsouth ? r : g.je + 1 + (r - nsouth);
  } else {
    p -= rows;
    const int w = nwest + neast, r = p / w, c = p - r * w;
    if (r >= nmid) return;
    j = g.js + r, i = c < nwest ? c : g.ie + 1 + (c - nwest);
  }
  const long ch = IDX3(g, i, j, k);
  const bool in_i = i >= g.is && i <= g.ie, in_j = j >= g.js && j <= g.je;
  if (!(in_i && (in_j || j == g.je + 1))) u_out[ch] = u[ch];
  if (!((in_i || i == g.ie + 1) && in_j)) v_out[ch] = v[ch];
}
__host__ __device__ __forceinline__ int wind_halo_points(const Geo& g) {
  return (g.js + g.nj - 1 - g.je) * g.ni + (g.je - g.js + 1) * (g.is + g.ni - 1 - g.ie);
}
int launch_fvtp2d(const Geo& g, const Met& m, const real* q, const real* crx, const real* cry,
                  const real* xfx, const real* yfx, real* fx, real* fy, const real* xmf,
                  const real* ymf, int hord, int nlev, hipStream_t st);
// Optional fused del-n damping of the same scalar (FiniteVolumeTransport calls DelnFlux on q right after the transport,
// fvtp2d.py:338-345; d_sw also needs DelnFluxNoSG(w) next to the transport of w).  DMODE -1: none; 0: damping fluxes
// written to fx2o / fy2o; 1: added to fx / fy; 2: added mass-weighted (delnflux.py:318-328).
struct FvDamp {
  const real* damp_k;
  const real* nord_k;
  const real* mass;
  real* fx2o;
  real* fy2o;
  int nmax, mass_given;
  // epilogue (EPI > 0): qout = q * amass + flux_increment(fx, fy) (apply_fluxes, d_sw.py:122-145) is written instead of
  // the fluxes; EPI == 2 additionally turns the damping fluxes into dw / heat_s / diss_est (heat_diss, d_sw.py:63-103)
  real* qout;
  const real* amass;
  real* dw;
  real* heat_s;
  real* diss_est;
  const real* damp_w_k;
  const real* ke_bg_k;
  double dt;
  // optional accumulators of the final fluxes (EPI == 0): the mass-flux half of flux_capacitor, mfx += fx, mfy += fy
  real* accx;
  real* accy;
  // optional 2-D field added to q AFTER the fused damping has read it: the transported scalar is q + add2d while the damped
  // one is q (d_sw: absolute vorticity = relative vorticity + fC_agrid, d_sw.py:389-402, damping acts on the relative one)
  const real* add2d;
  // optional (EPI == 0): instead of storing the fluxes, finish u_and_v_from_ke (d_sw.py:406-477) with them -- the transported
  // scalar is then the absolute vorticity: u = u * dx + ke - ke[i+1] + fy, v = v * dy + ke - ke[j+1] - fx, each on the face the
  // flux lives on, by the workgroup that owns the face
  real* u_upd;
  real* v_upd;
  const real* ke;
  // optional: where the results of that go instead of u_upd / v_upd themselves (launch_d_sw: a workspace copy, so that the kernel
  // that follows -- heating + final wind update in one -- reads the intermediate winds of its neighbours race-free)
  real* u_out;
  real* v_out;
};

int launch_transport(const Geo& g, const Met& m, const real* q, const real* crx, const real* cry, const real* xfx,
                     const real* yfx, real* fx, real* fy, const real* xmf, const real* ymf, int hord, int nlev,
                     int dmode, int epi, const FvDamp& dp, hipStream_t st);
int launch_transport_scalars3(const Geo& g, const Met& m, const real* w, const real* q_con, const real* pt, const real* crx,
                              const real* cry, const real* xfx, const real* yfx, const real* xmf, const real* ymf, int nlev,
                              const FvDamp& dpw, const FvDamp& dpq, const FvDamp& dpt, hipStream_t st);
// k_fvt.hip: the lean transport kernel of the production tilings (every tile edge on a workgroup-tile boundary); same contract
// as launch_transport with the unit fluxes resolved (xu = xmf or xfx); PACE_ERR_UNSUPPORTED for the calls it does not cover
int launch_transport_lean(const Geo& g, const Met& m, const real* q, const real* crx, const real* cry, const real* xfx,
                          const real* yfx, real* fx, real* fy, const real* xu, const real* yu, int hord, int nlev, int dmode,
                          int epi, const FvDamp& dp, hipStream_t st);

// the winds of d_sw as the fifth pass of the scalar-phase kernel (fvt_core.h): the vorticity transport (u_and_v_from_ke), the
// vorticity damping, the dissipative heating and the final winds.  u_out / v_out: buffers of their own; copy_halo: they also get
// the halo u / v have (the caller swaps the buffers).
struct DswWinds {
  const real *rel_vort, *u, *v, *ke, *vort_b;
  real *u_out, *v_out, *heat_source;
  int do_skeb, copy_halo;
  int ke_plus_vort;  // `ke` is the plain kinetic energy: the kernel adds the damped vorticity to it (divergence_damping.py:161-185)
  double d_con;
};
bool dsw_scalars_take_winds();
int launch_dsw_scalars_lean(const Geo& g, const Met& m, const real* delp, const real* pt, const real* w, const real* q_con,
                            real* const* outs, const real* crx, const real* cry, const real* xfx, const real* yfx, real* mfx,
                            real* mfy, real* dw, real* heat_s, real* diss_est, const real* kc, int hord, int nmax_v, int nmax_w,
                            int nmax_t, double dt, hipStream_t st, const DswWinds* winds = nullptr);
// whether launch_transport_lean takes this geometry with this order (fp64 build, tiling, row alignment)
bool transport_lean_covers(const Geo& g, int hord);

// The metric fields the transport kernels read (kernel arguments live in SGPRs: the full pace_metrics_t is 43 pointers, and the
// register allocator answered with ~300 SGPR spills -- v_readlane / v_writelane traffic on the vector pipe)
struct FvMet {
  const real *area, *rarea, *dxa, *dya, *dx, *dy, *del6_u, *del6_v;
};
static inline FvMet fv_met(const Met& m) { return FvMet{m.area, m.rarea, m.dxa, m.dya, m.dx, m.dy, m.del6_u, m.del6_v}; }

// Workgroup -> (tile, level) of the transport kernels.  Workgroups are handed to the eight XCDs round-robin in launch order, and
// every XCD has its own 4 MB L2: with the plain (x, y, z) order, neighbouring tiles of a level land on DIFFERENT XCDs and every
// line of their overlapping footprints is fetched from memory once per XCD (measured: 1.9 x the algorithmic bytes).  Here a level
// belongs to ONE XCD: XCD x works through levels x, x + 8, x + 16, ... tile by tile, so the halo lines shared by neighbouring
// tiles are L2 hits.  (Affinity only: nothing depends on where a workgroup really runs.)
struct FvTile {
  int bx, by, bz;
};
__device__ __forceinline__ FvTile fv_tile_of_workgroup() {
#ifdef PACE_EMU
  return FvTile{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
#else
  const int gx = gridDim.x, gy = gridDim.y, nlev = gridDim.z;
  const int tpl = gx * gy;
  const int b = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  const int full = (nlev / 8) * 8;  // levels that can be dealt out eight at a time
  int lev, t;
  if (b < full * tpl) {
    const int xcd = b & 7, slot = b >> 3;
    lev = (slot / tpl) * 8 + xcd;
    t = slot - (slot / tpl) * tpl;
  } else {
    lev = b / tpl;
    t = b - lev * tpl;
  }
  // within a level: the four corner tiles first, then the edge tiles, then the interior ones -- the corner and edge forms take
  // 1.2 - 2 x as long as the straight-line interior code, and the workgroups that start last should not be the longest ones
  if (gx >= 3 && gy >= 3) {
    if (t < 4) return FvTile{(t & 1) ? gx - 1 : 0, (t & 2) ? gy - 1 : 0, lev};
    t -= 4;
    const int nsn = 2 * (gx - 2), nwe = 2 * (gy - 2);
    if (t < nsn) return FvTile{1 + (t >> 1), (t & 1) ? gy - 1 : 0, lev};
    t -= nsn;
    if (t < nwe) return FvTile{(t & 1) ? gx - 1 : 0, 1 + (t >> 1), lev};
    t -= nwe;
    return FvTile{1 + t % (gx - 2), 1 + t / (gx - 2), lev};
  }
  return FvTile{t % gx, t / gx, lev};
#endif
}
int launch_delnflux(const Geo& g, const Met& m, int mode, const real* q, real* fx, real* fy,
                    const real* mass, const real* damp_k, const real* nord_k, int nmax, int mass_given,
                    int nlev, hipStream_t st);
int launch_a2b_ord4(const Geo& g, const Met& m, real* qin, real* qout, int k0, int k1, int replace, hipStream_t st);
// the same with the input tile staged in LDS (qout must not alias qin)
int launch_a2b_ord4_tiled(const Geo& g, const Met& m, const real* qin, real* qout, int k0, int k1, hipStream_t st);
// up to four fields (each with its own level range) in one pair of launches
int launch_a2b_ord4_batch(const Geo& g, const Met& m, const real* const* qin, real* const* qout, const int* k0, const int* k1,
                          int nfields, hipStream_t st);
int64_t dsw_workspace_bytes(const Geo& g);
// whether launch_d_sw can write the four transported scalars to separate buffers (pace_dsw_config_t::delp_out ...)
bool dsw_pingpong_supported(const Geo& g, const pace_dsw_config_t* cfg);
bool dsw_winds_in_scalars(const Geo& g, const pace_dsw_config_t* cfg);
int dsw_prepare(const Geo& g, const pace_column_t* col, void* ws, hipStream_t st);
int launch_d_sw(const Geo& g, const Met& m, const pace_column_t* col, const pace_dsw_config_t* cfg, void* ws,
                real* delpc, real* delp, real* pt, real* u, real* v, real* w, real* uc, real* vc,
                const real* ua, const real* va, real* divgd, real* mfx, real* mfy, real* cx, real* cy,
                real* crx, real* cry, real* xfx, real* yfx, real* q_con, const real* zh,
                real* heat_source, real* diss_est, double dt, int phases, hipStream_t st);
int64_t riem3_workspace_bytes(const Geo& g);
int launch_riem_solver3(const Geo& g, void* ws, int last_call, double dt, const real* cappa, double ptop,
                        const real* zs, const real* wsd, real* delz, const real* q_con, const real* delp,
                        const real* pt, real* zh, real* pe, real* ppe, real* pk3, real* pk, real* peln,
                        real* w, double p_fac, hipStream_t st);
// k_ppm.hip: XPiecewiseParabolic / YPiecewiseParabolic on a window (axis 0 = x, 1 = y)
int launch_ppm1d(const Geo& g, const Met& m, int axis, int iord, const real* q, const real* c, real* out, int i0, int j0,
                 int k0, int ni, int nj, int nk, hipStream_t st);
// k_dsw.hip: DivergenceDamping.__call__
int launch_divergence_damping(const Geo& g, const Met& m, const real* u, const real* v, const real* va, real* vort_b,
                              const real* ua, real* divg_d, real* vc, real* uc, real* delpc, real* ke,
                              const real* rel_vort_agrid, double dt, const real* d2_bg_dev, int kstart, int nonzero_nord,
                              double dddmp, double d4_bg, real* da, real* db, hipStream_t st, bool skip_dead = false,
                              bool ke_by_consumer = false, const int* ddh_tab = nullptr);
// k_riem3f.hip: both column solvers as one k-cooperative kernel (16 lanes per column), no workspace
bool riem_column_supported(const Geo& g);
int launch_riem_solver3_column(const Geo& g, int last_call, double dt, const real* cappa, double ptop, const real* zs,
                               const real* wsd, real* delz, const real* q_con, const real* delp, const real* pt,
                               real* zh, real* pe, real* ppe, real* pk3, real* pk, real* peln, real* w,
                               double p_fac, hipStream_t st);
int launch_riem_solver_c_column(const Geo& g, double dt2, const real* cappa, double ptop, const real* hs, const real* ws3,
                                const real* ptc, const real* q_con, const real* delpc, real* gz, real* pef,
                                const real* w3, double p_fac, hipStream_t st);
// k_sim1.hip: Sim1Solver as a class of its own (not the hot path)
int64_t sim1_workspace_bytes(const Geo& g);
int launch_sim1_solver(const Geo& g, void* ws, int n_halo, double dt, double p_fac, const real* gamma, const real* cp3,
                       real* pe, const real* delta_mass, const real* pm, const real* pem, real* w, real* dz,
                       const real* pt, const real* ws2d, hipStream_t st);
// k_csw.hip
int64_t csw_workspace_bytes(const Geo& g);
int launch_d2a2c_vect(const Geo& g, const Met& m, void* ws, real* uc, real* vc, const real* u, const real* v,
                      real* ua, real* va, real* utc, real* vtc, hipStream_t st);
int launch_c_sw(const Geo& g, const Met& m, void* ws, real* delpc, real* ptc, real* delp, real* pt,
                const real* u, const real* v, real* w, real* uc, real* vc, real* ua, real* va,
                real* ut, real* vt, real* divgd, real* omga, double dt2, int nord, hipStream_t st, int part = 0);
// k_riem3.hip (C-grid solver)
int64_t riemc_workspace_bytes(const Geo& g);
int launch_riem_solver_c(const Geo& g, void* ws, double dt2, const real* cappa, double ptop, const real* hs,
                         const real* ws3, const real* ptc, const real* q_con, const real* delpc, real* gz,
                         real* pef, const real* w3, double p_fac, hipStream_t st);
// k_acoustic.hip
int64_t updatedzc_workspace_bytes(const Geo& g);
int launch_updatedzc(const Geo& g, const Met& m, void* ws_, const real* dp_ref, const real* zs, const real* ut,
                     const real* vt, real* gz, real* ws, double dt, hipStream_t st);
int64_t updatedzd_workspace_bytes(const Geo& g);
int launch_updatedzd(const Geo& g, const Met& m, void* ws_, const pace_updatedzd_k_t* kc, const real* zs, real* zh,
                     const real* crx, const real* cry, const real* xfx, const real* yfx, real* wsd, double dt,
                     int hord_tm, hipStream_t st);
int launch_gz_from_surface(const Geo& g, const real* zs, const real* delz, real* gz, hipStream_t st);
int launch_scale_copy(const Geo& g, const real* src, real* dst, double factor, int scale, int halo, int nlev,
                      hipStream_t st);
int launch_p_grad_c(const Geo& g, const Met& m, real* uc, real* vc, const real* delpc, const real* pkc,
                    const real* gz, double dt2, hipStream_t st);
int64_t nh_p_grad_workspace_bytes(const Geo& g);
int launch_nh_p_grad(const Geo& g, const Met& m, void* ws_, real* u, real* v, real* pp, real* gz, real* pk3,
                     real* delp, double dt, double ptop, double akap, hipStream_t st);
int launch_edge_pe(const Geo& g, real* pe, const real* delp, double ptop, hipStream_t st);
int launch_pk3_halo(const Geo& g, real* pk3, const real* delp, double ptop, double akap, hipStream_t st);
int launch_ray_fast(const Geo& g, real* u, real* v, real* w, const double* dp, const double* pfull, double dt,
                    double ptop, double rf_cutoff, double tau, int hydrostatic, hipStream_t st);
int64_t del2cubed_workspace_bytes(const Geo& g);
int launch_del2cubed(const Geo& g, const Met& m, void* ws_, real* qdel, double cd, int nmax, hipStream_t st);
int launch_diffusive_heating(const Geo& g, const real* delp, const real* delz, const real* cappa,
                             const real* heat_source, real* pt, double delt_time_factor, int nlev, hipStream_t st);
// k_halo.hip
int launch_halo_copy(const Geo& g, const pace_halo_desc_t* descs, int ndesc, int unpack, hipStream_t st);
// k_tracer.hip
int launch_tracer_flux_compute(const Geo& g, const Met& m, const real* cx, const real* cy, real* xfx, real* yfx,
                               hipStream_t st);
int launch_tracer_divide(const Geo& g, real* cxd, real* xfx, real* mfxd, real* cyd, real* yfx, real* mfyd,
                         int n_split, hipStream_t st);
int launch_apply_mass_flux(const Geo& g, const Met& m, const real* dp1, const real* mfx, const real* mfy, real* dp2,
                           hipStream_t st);
int launch_apply_tracer_flux(const Geo& g, const Met& m, real* q, const real* dp1, const real* fx, const real* fy,
                             const real* dp2, hipStream_t st);
int64_t map_single_workspace_bytes(const Geo& g, int nq);
int launch_map_fields(const Geo& g, void* ws, real* const* q, int nq, const real* pe1, const real* pe2, const real* qs,
                      double qmin, int kord, int iv, int xstag, int ystag, hipStream_t st);
int launch_fillz(const Geo& g, real* const* q, int nq, const real* dp, hipStream_t st);
int launch_l2e_prepare(const Geo& g, const real* const* water, real* q_con, real* pt, real* cappa, real* delp,
                       real* delz, const real* pe, real* pe1, real* pe2, const real* ak, const real* bk, real* dp2,
                       real* ps, real* pn2, const real* peln, real* pk, double ptop, double akap, double r_vir,
                       hipStream_t st);
int launch_l2e_post(const Geo& g, const real* const* water, real* q_con, real* pkz, const real* pt, real* cappa,
                    const real* delp, real* delz, real* peln, real* pe0, const real* pn2, double r_vir, hipStream_t st);
int launch_l2e_pressures(const Geo& g, int dir, const real* pe, const real* pe1, const real* ak, const real* bk,
                         real* pe0, real* pe3, hipStream_t st);
int launch_l2e_finish(const Geo& g, const real* const* water, real* pe, const real* pe2, real* pt, const real* pkz,
                      double r_vir, int last_step, hipStream_t st);
int launch_fv_setup_pt(const Geo& g, real* const* water, real* q_con, real* pkz, real* pt, real* cappa,
                       const real* delp, const real* delz, real* dp1, hipStream_t st);
int launch_omega_from_w(const Geo& g, const real* delp, const real* delz, const real* w, real* omga, hipStream_t st);
int launch_neg_adj3(const Geo& g, real* const* water, real* qcld, real* pt, const real* delp, hipStream_t st);
int launch_c2l(const Geo& g, const Met& m, int order, const real* u, const real* v, const real* a11, const real* a12,
               const real* a21, const real* a22, real* ua, real* va, hipStream_t st);
// k_stencils.hip: per-stencil device implementations (pace_stencil)
int launch_stencil(const Geo& g, const Met& m, int id, void* const* fields, int nfields, const double* scalars, int nscalars,
                   const int* origin, const int* domain, hipStream_t st);
int launch_swap_dp(const Geo& g, real* dp1, real* dp2, hipStream_t st);
int launch_zero_data(const Geo& g, real* mfxd, real* mfyd, real* cxd, real* cyd, real* heat_source, real* diss_estd,
                     int first_timestep, hipStream_t st);
int launch_interface_pressure(const Geo& g, const real* delp, real* pem, double ptop, hipStream_t st);
