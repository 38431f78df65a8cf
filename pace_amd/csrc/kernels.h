// Host-side launch functions (one per reference class on the hot path).
#pragma once
#include "common.h"

int launch_fxadv(const Geo& g, const Met& m, const double* uc, const double* vc, double* crx, double* cry,
                 double* xfx, double* yfx, double* ut, double* vt, double dt, double* cx_acc, double* cy_acc,
                 hipStream_t st);
int launch_fvtp2d(const Geo& g, const Met& m, const double* q, const double* crx, const double* cry,
                  const double* xfx, const double* yfx, double* fx, double* fy, const double* xmf,
                  const double* ymf, int hord, int nlev, hipStream_t st);
// Optional fused del-n damping of the same scalar (FiniteVolumeTransport calls DelnFlux on q right after the transport,
// fvtp2d.py:338-345; d_sw also needs DelnFluxNoSG(w) next to the transport of w).  DMODE -1: none; 0: damping fluxes
// written to fx2o / fy2o; 1: added to fx / fy; 2: added mass-weighted (delnflux.py:318-328).
struct FvDamp {
  const double* damp_k;
  const double* nord_k;
  const double* mass;
  double* fx2o;
  double* fy2o;
  int nmax, mass_given;
  // epilogue (EPI > 0): qout = q * amass + flux_increment(fx, fy) (apply_fluxes, d_sw.py:122-145) is written instead of
  // the fluxes; EPI == 2 additionally turns the damping fluxes into dw / heat_s / diss_est (heat_diss, d_sw.py:63-103)
  double* qout;
  const double* amass;
  double* dw;
  double* heat_s;
  double* diss_est;
  const double* damp_w_k;
  const double* ke_bg_k;
  double dt;
  // optional accumulators of the final fluxes (EPI == 0): the mass-flux half of flux_capacitor, mfx += fx, mfy += fy
  double* accx;
  double* accy;
  // optional 2-D field added to q AFTER the fused damping has read it: the transported scalar is q + add2d while the damped
  // one is q (d_sw: absolute vorticity = relative vorticity + fC_agrid, d_sw.py:389-402, damping acts on the relative one)
  const double* add2d;
  // optional (EPI == 0): instead of storing the fluxes, finish u_and_v_from_ke (d_sw.py:406-477) with them -- the transported
  // scalar is then the absolute vorticity: u = u * dx + ke - ke[i+1] + fy, v = v * dy + ke - ke[j+1] - fx, each on the face the
  // flux lives on, by the workgroup that owns the face
  double* u_upd;
  double* v_upd;
  const double* ke;
};

int launch_transport(const Geo& g, const Met& m, const double* q, const double* crx, const double* cry, const double* xfx,
                     const double* yfx, double* fx, double* fy, const double* xmf, const double* ymf, int hord, int nlev,
                     int dmode, int epi, const FvDamp& dp, hipStream_t st);
int launch_delnflux(const Geo& g, const Met& m, int mode, const double* q, double* fx, double* fy,
                    const double* mass, const double* damp_k, const double* nord_k, int nmax, int mass_given,
                    int nlev, hipStream_t st);
int launch_a2b_ord4(const Geo& g, const Met& m, double* qin, double* qout, int k0, int k1, int replace, hipStream_t st);
int64_t dsw_workspace_bytes(const Geo& g);
int dsw_prepare(const Geo& g, const pace_column_t* col, void* ws, hipStream_t st);
int launch_d_sw(const Geo& g, const Met& m, const pace_column_t* col, const pace_dsw_config_t* cfg, void* ws,
                double* delpc, double* delp, double* pt, double* u, double* v, double* w, double* uc, double* vc,
                const double* ua, const double* va, double* divgd, double* mfx, double* mfy, double* cx, double* cy,
                double* crx, double* cry, double* xfx, double* yfx, double* q_con, const double* zh,
                double* heat_source, double* diss_est, double dt, int phases, hipStream_t st);
int64_t riem3_workspace_bytes(const Geo& g);
int launch_riem_solver3(const Geo& g, void* ws, int last_call, double dt, const double* cappa, double ptop,
                        const double* zs, const double* wsd, double* delz, const double* q_con, const double* delp,
                        const double* pt, double* zh, double* pe, double* ppe, double* pk3, double* pk, double* peln,
                        double* w, double p_fac, hipStream_t st);
// k_ppm.hip: XPiecewiseParabolic / YPiecewiseParabolic on a window (axis 0 = x, 1 = y)
int launch_ppm1d(const Geo& g, const Met& m, int axis, int iord, const double* q, const double* c, double* out, int i0, int j0,
                 int k0, int ni, int nj, int nk, hipStream_t st);
// k_dsw.hip: DivergenceDamping.__call__
int launch_divergence_damping(const Geo& g, const Met& m, const double* u, const double* v, const double* va, double* vort_b,
                              const double* ua, double* divg_d, double* vc, double* uc, double* delpc, double* ke,
                              const double* rel_vort_agrid, double dt, const double* d2_bg_dev, int kstart, int nonzero_nord,
                              double dddmp, double d4_bg, double* da, double* db, hipStream_t st);
// k_riem3f.hip: both column solvers as one k-cooperative kernel (16 lanes per column), no workspace
bool riem_column_supported(const Geo& g);
int launch_riem_solver3_column(const Geo& g, int last_call, double dt, const double* cappa, double ptop, const double* zs,
                               const double* wsd, double* delz, const double* q_con, const double* delp, const double* pt,
                               double* zh, double* pe, double* ppe, double* pk3, double* pk, double* peln, double* w,
                               double p_fac, hipStream_t st);
int launch_riem_solver_c_column(const Geo& g, double dt2, const double* cappa, double ptop, const double* hs, const double* ws3,
                                const double* ptc, const double* q_con, const double* delpc, double* gz, double* pef,
                                const double* w3, double p_fac, hipStream_t st);
// k_sim1.hip: Sim1Solver as a class of its own (not the hot path)
int64_t sim1_workspace_bytes(const Geo& g);
int launch_sim1_solver(const Geo& g, void* ws, int n_halo, double dt, double p_fac, const double* gamma, const double* cp3,
                       double* pe, const double* delta_mass, const double* pm, const double* pem, double* w, double* dz,
                       const double* pt, const double* ws2d, hipStream_t st);
// k_csw.hip
int64_t csw_workspace_bytes(const Geo& g);
int launch_d2a2c_vect(const Geo& g, const Met& m, void* ws, double* uc, double* vc, const double* u, const double* v,
                      double* ua, double* va, double* utc, double* vtc, hipStream_t st);
int launch_c_sw(const Geo& g, const Met& m, void* ws, double* delpc, double* ptc, const double* delp, const double* pt,
                const double* u, const double* v, const double* w, double* uc, double* vc, double* ua, double* va,
                double* ut, double* vt, double* divgd, double* omga, double dt2, int nord, hipStream_t st);
// k_riem3.hip (C-grid solver)
int64_t riemc_workspace_bytes(const Geo& g);
int launch_riem_solver_c(const Geo& g, void* ws, double dt2, const double* cappa, double ptop, const double* hs,
                         const double* ws3, const double* ptc, const double* q_con, const double* delpc, double* gz,
                         double* pef, const double* w3, double p_fac, hipStream_t st);
// k_acoustic.hip
int64_t updatedzc_workspace_bytes(const Geo& g);
int launch_updatedzc(const Geo& g, const Met& m, void* ws_, const double* dp_ref, const double* zs, const double* ut,
                     const double* vt, double* gz, double* ws, double dt, hipStream_t st);
int64_t updatedzd_workspace_bytes(const Geo& g);
int launch_updatedzd(const Geo& g, const Met& m, void* ws_, const pace_updatedzd_k_t* kc, const double* zs, double* zh,
                     const double* crx, const double* cry, const double* xfx, const double* yfx, double* wsd, double dt,
                     int hord_tm, hipStream_t st);
int launch_gz_from_surface(const Geo& g, const double* zs, const double* delz, double* gz, hipStream_t st);
int launch_scale_copy(const Geo& g, const double* src, double* dst, double factor, int scale, int halo, int nlev,
                      hipStream_t st);
int launch_p_grad_c(const Geo& g, const Met& m, double* uc, double* vc, const double* delpc, const double* pkc,
                    const double* gz, double dt2, hipStream_t st);
int64_t nh_p_grad_workspace_bytes(const Geo& g);
int launch_nh_p_grad(const Geo& g, const Met& m, void* ws_, double* u, double* v, double* pp, double* gz, double* pk3,
                     double* delp, double dt, double ptop, double akap, hipStream_t st);
int launch_edge_pe(const Geo& g, double* pe, const double* delp, double ptop, hipStream_t st);
int launch_pk3_halo(const Geo& g, double* pk3, const double* delp, double ptop, double akap, hipStream_t st);
int launch_ray_fast(const Geo& g, double* u, double* v, double* w, const double* dp, const double* pfull, double dt,
                    double ptop, double rf_cutoff, double tau, int hydrostatic, hipStream_t st);
int64_t del2cubed_workspace_bytes(const Geo& g);
int launch_del2cubed(const Geo& g, const Met& m, void* ws_, double* qdel, double cd, int nmax, hipStream_t st);
int launch_diffusive_heating(const Geo& g, const double* delp, const double* delz, const double* cappa,
                             const double* heat_source, double* pt, double delt_time_factor, int nlev, hipStream_t st);
// k_halo.hip
int launch_halo_copy(const Geo& g, const pace_halo_desc_t* descs, int ndesc, int unpack, hipStream_t st);
// k_tracer.hip
int launch_tracer_flux_compute(const Geo& g, const Met& m, const double* cx, const double* cy, double* xfx, double* yfx,
                               hipStream_t st);
int launch_tracer_divide(const Geo& g, double* cxd, double* xfx, double* mfxd, double* cyd, double* yfx, double* mfyd,
                         int n_split, hipStream_t st);
int launch_apply_mass_flux(const Geo& g, const Met& m, const double* dp1, const double* mfx, const double* mfy, double* dp2,
                           hipStream_t st);
int launch_apply_tracer_flux(const Geo& g, const Met& m, double* q, const double* dp1, const double* fx, const double* fy,
                             const double* dp2, hipStream_t st);
int64_t map_single_workspace_bytes(const Geo& g, int nq);
int launch_map_fields(const Geo& g, void* ws, double* const* q, int nq, const double* pe1, const double* pe2, const double* qs,
                      double qmin, int kord, int iv, int xstag, int ystag, hipStream_t st);
int launch_fillz(const Geo& g, double* const* q, int nq, const double* dp, hipStream_t st);
int launch_l2e_prepare(const Geo& g, const double* const* water, double* q_con, double* pt, double* cappa, double* delp,
                       double* delz, const double* pe, double* pe1, double* pe2, const double* ak, const double* bk, double* dp2,
                       double* ps, double* pn2, const double* peln, double* pk, double ptop, double akap, double r_vir,
                       hipStream_t st);
int launch_l2e_post(const Geo& g, const double* const* water, double* q_con, double* pkz, const double* pt, double* cappa,
                    const double* delp, double* delz, double* peln, double* pe0, const double* pn2, double r_vir, hipStream_t st);
int launch_l2e_pressures(const Geo& g, int dir, const double* pe, const double* pe1, const double* ak, const double* bk,
                         double* pe0, double* pe3, hipStream_t st);
int launch_l2e_finish(const Geo& g, const double* const* water, double* pe, const double* pe2, double* pt, const double* pkz,
                      double r_vir, int last_step, hipStream_t st);
int launch_fv_setup_pt(const Geo& g, double* const* water, double* q_con, double* pkz, double* pt, double* cappa,
                       const double* delp, const double* delz, double* dp1, hipStream_t st);
int launch_omega_from_w(const Geo& g, const double* delp, const double* delz, const double* w, double* omga, hipStream_t st);
int launch_neg_adj3(const Geo& g, double* const* water, double* qcld, double* pt, const double* delp, hipStream_t st);
int launch_c2l(const Geo& g, const Met& m, int order, const double* u, const double* v, const double* a11, const double* a12,
               const double* a21, const double* a22, double* ua, double* va, hipStream_t st);
int launch_fvtp2d_march(const Geo& g, const Met& m, const double* q, const double* crx, const double* cry, const double* xfx,
                        const double* yfx, double* fx, double* fy, int ib, int nx, int jb, int ny, int nlev, hipStream_t st);
int launch_swap_dp(const Geo& g, double* dp1, double* dp2, hipStream_t st);
int launch_zero_data(const Geo& g, double* mfxd, double* mfyd, double* cxd, double* cyd, double* heat_source, double* diss_estd,
                     int first_timestep, hipStream_t st);
int launch_interface_pressure(const Geo& g, const double* delp, double* pem, double ptop, hipStream_t st);
