/* pace_hip.h -- C ABI of libpace_hip.so: MI355X (gfx950) kernels for the FV3 acoustic substep.
 *
 * This is the drop-in boundary: each entry point replaces what the reference executes *under*
 * one of its Python classes (a chain of GT4Py FrozenStencil launches, dsl/pace/dsl/stencil.py:395-434).
 * Plain pointers and sizes only; no allocation, no global state, re-entrant per stream.  All
 * `pace_real_t*` arguments are DEVICE pointers unless stated otherwise.  Every function returns
 * PACE_OK (0) or a negative PACE_ERR_* code; nothing is written on error.
 *
 * Field layout: 3-D fields are [k][j][i], i fastest, logical shape (N+7, N+7, nz+1) with origin
 * (3,3,0) exactly as the reference allocates them (util/pace/util/initialization/sizer.py:132-155);
 * row stride `sj` (>= N+7) and level stride `sk` (>= sj*(N+7)) are given in pace_geom_t.  2-D metric
 * fields share the row stride.  K-fields are dense arrays of nk (or nk+1) doubles.
 */
#ifndef PACE_HIP_H
#define PACE_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Storage type of every DEVICE field, metric and K-array: double in libpace_hip.so, float in libpace_hip_f32.so (the same
 * sources compiled with -DPACE_REAL_FLOAT; dsl/pace/dsl/typing.py:24 -- the reference's PACE_FLOAT_PRECISION switch).  Strides
 * in pace_geom_t count elements of this type.  Scalars passed by value, HOST arrays and the arithmetic inside the kernels stay
 * double in both builds.  pace_real_bytes() tells which build a library is. */
#ifdef PACE_REAL_FLOAT
typedef float pace_real_t;
#else
typedef double pace_real_t;
#endif

#define PACE_OK 0
#define PACE_ERR_ARG (-1)
#define PACE_ERR_LAUNCH (-2)
#define PACE_ERR_UNSUPPORTED (-3)

typedef struct {
  int32_t n;   /* cells per tile edge (C<n>) */
  int32_t nk;  /* number of layers nz */
  int32_t sj;  /* row stride in doubles */
  int32_t pad_;
  int64_t sk;  /* level stride in doubles */
} pace_geom_t;

/* Read-only grid metrics, i.e. pace.util.grid.GridData / DampingCoefficients
 * (util/pace/util/grid/helper.py:21-45,306-530).  2-D device arrays with row stride sj. */
typedef struct {
  const pace_real_t *area, *rarea, *rarea_c;
  const pace_real_t *dx, *dy, *dxa, *dya, *dxc, *dyc;
  const pace_real_t *rdx, *rdy, *rdxa, *rdya, *rdxc, *rdyc;
  const pace_real_t *cosa, *rsina, *cosa_u, *cosa_v, *cosa_s;
  const pace_real_t *sina_u, *sina_v, *rsin_u, *rsin_v, *rsin2;
  const pace_real_t *sin_sg1, *sin_sg2, *sin_sg3, *sin_sg4;
  const pace_real_t *cos_sg1, *cos_sg2, *cos_sg3, *cos_sg4;
  const pace_real_t *del6_u, *del6_v, *divg_u, *divg_v;
  const pace_real_t *fC, *fC_agrid;
  const pace_real_t *edge_w, *edge_e; /* length nj, indexed by j */
  const pace_real_t *edge_s, *edge_n; /* length ni, indexed by i */
  /* a2b_ord4 corner extrapolation weights x1/(x2-x1) (a2b_ord4.py:43-56) for the corner
   * points sw(is,js), nw-stencil(ie+1,js), ne(ie+1,je+1), se-stencil(is,je+1), three diagonals
   * each, precomputed on the host from lon/lat (HOST values, copied by value). */
  double a2b_corner_w[4][3];
  double da_min, da_min_c;
} pace_metrics_t;

/* Column namelist (d_sw.get_column_namelist, fv3core/pace/fv3core/stencils/d_sw.py:633-683)
 * expanded to one value per level; HOST arrays of length nk. */
typedef struct {
  const double *nord, *nord_v, *nord_w, *nord_t;
  const double *damp_vt, *damp_w, *damp_t;
  const double *d2_divg, *d_con, *ke_bg;
  /* calc_damp (delnflux.py:21-38) = (damp_c * da)^(nord+1), evaluated by the host exactly where the
   * reference evaluates it: fac_vt = (damp_vt, da_min, nord_v), fac_t = (damp_t, da_min, nord_t) in
   * DelnFlux.__init__ (delnflux.py:1001-1003); fac_vt_c = (damp_vt, da_min_c, nord_v),
   * fac_w_c = (damp_w, da_min_c, nord_w) in d_sw.py:924-933. */
  const double *fac_vt, *fac_t, *fac_vt_c, *fac_w_c;
} pace_column_t;

/* flags of pace_dsw_config_t */
#define PACE_DSW_SKIP_DEAD_OUTPUTS 1 /* delpc, divgd, uc, vc are not brought to the state the reference leaves them in: they are
                                      * work fields of DivergenceDamping (divergence_damping.py:561-600) that c_sw recomputes before
                                      * anything reads them again (d_sw.py:1032-1033, dyn_core.py:720-852).  Their contents are
                                      * unspecified after the call, and so are the 3 x 3 corner blocks of the halo of delp, pt, w,
                                      * q_con (which the next halo update overwrites); every other value of every other argument
                                      * is unaffected, bit for bit.  This is what the acoustic loop asks for.
                                      * The default (0) is the reference's FULL contract: every argument over the whole storage as
                                      * TranslateD_SW compares it (translate_d_sw.py:36-65) -- the halo of divgd / uc / vc in the
                                      * state the damping's in-place passes and corner fills leave it (divergence_damping.py:579-600),
                                      * delpc's halo untouched, the corner blocks of the four scalars as FiniteVolumeTransport's
                                      * in-place corner copies leave them (fvtp2d.py:262-345). */

typedef struct {
  /* sizeof(pace_dsw_config_t) of the header the caller was built with: a mismatch is refused (PACE_ERR_ARG) instead of
   * reading fields the caller never set.  Zero-initialise the struct, then fill it. */
  int32_t struct_bytes;
  int32_t flags; /* PACE_DSW_* */
  int32_t hord_dp, hord_tm, hord_vt, hord_mt;
  int32_t nord;
  int32_t do_skeb;
  double dddmp, d4_bg, d_con;
  /* Optional separate outputs of the four scalars d_sw transports (all four or none; NULL = the reference's in-place update).
   * d_sw.py:148-201,331-350 overwrite delp, pt, w, q_con cell by cell while the transport of the neighbouring cells still reads
   * them -- the reference gets away with it through full-field temporaries; here one kernel does transport and update, so the
   * new values need a buffer of their own.  With outputs given (distinct from the inputs, same layout, 16-byte aligned) the
   * compute domain AND the halo of each input are written there and the inputs are left as they were: the caller swaps the
   * buffers (pace_amd: Quantity.swap_storage).  Without them the library writes to its workspace and copies back.
   * Only where pace_d_sw_pingpong_supported() says so; otherwise PACE_ERR_UNSUPPORTED. */
  pace_real_t *delp_out, *pt_out, *w_out, *q_con_out;
  /* Optional separate outputs of the D-grid winds, both or none, only together with the four above and only in calls that run
   * the whole of d_sw (pace_d_sw, pace_d_sw_overlapped, pace_d_sw_phases with 2, 4 and 8 set): u and v are then left as they were
   * and the updated winds (d_sw.py:406-477,582-608) are written to u_out / v_out, halo included -- the caller swaps the buffers.
   * The winds are updated by the kernel that transports the scalars, tile by tile, and a tile reads the old wind on the face its
   * neighbour writes.  Without them that kernel writes to the workspace and the winds are copied back. */
  pace_real_t *u_out, *v_out;
} pace_dsw_config_t;

/* ---- FiniteVolumeFluxPrep.__call__ (fv3core/pace/fv3core/stencils/fxadv.py:565-661) ---- */
int pace_fxadv(const pace_geom_t* geom, const pace_metrics_t* met, const pace_real_t* uc, const pace_real_t* vc,
               pace_real_t* crx, pace_real_t* cry, pace_real_t* x_area_flux, pace_real_t* y_area_flux,
               pace_real_t* uc_contra, pace_real_t* vc_contra, double dt, void* stream);

/* ---- XPiecewiseParabolic / YPiecewiseParabolic.__call__ (fv3core/pace/fv3core/stencils/xppm.py:290-355, yppm.py:290-355):
 * mean value of q_in advected through the x- (axis 0) or y- (axis 1) interfaces of the window origin (i0, j0, k0), domain
 * (ni, nj, nk) -- the origin / domain the reference class is constructed with.  iord in {5, 6, 8} (the sign is ignored, as
 * `mord = abs(iord)`).  Corner halos of q_in are the caller's business, as in the reference. */
int pace_ppm(const pace_geom_t* geom, const pace_metrics_t* met, int axis, int iord, const pace_real_t* q_in,
             const pace_real_t* c, pace_real_t* q_mean_advected, int i0, int j0, int k0, int ni, int nj, int nk,
             void* stream);

/* ---- DivergenceDamping.__call__ (fv3core/pace/fv3core/stencils/divergence_damping.py:482-632).  workspace: two fields
 * (pace_divergence_damping_workspace_bytes).  nord_col: HOST array of nk values (the class's nord_col K-field; the column is
 * split at its first positive entry, divergence_damping.py:307-331); d2_bg: DEVICE array of nk values.  In place, as the
 * reference: divg_d, uc, vc are work fields and end as the last iteration leaves them, delpc and damped_rel_vort_bgrid are
 * outputs, ke += damping. */
int64_t pace_divergence_damping_workspace_bytes(const pace_geom_t* geom);
int pace_divergence_damping(const pace_geom_t* geom, const pace_metrics_t* met, void* workspace, const pace_real_t* u,
                            const pace_real_t* v, const pace_real_t* va, pace_real_t* damped_rel_vort_bgrid,
                            const pace_real_t* ua, pace_real_t* divg_d, pace_real_t* vc, pace_real_t* uc,
                            pace_real_t* delpc, pace_real_t* ke, const pace_real_t* rel_vort_agrid, double dt,
                            const double* nord_col_host, const pace_real_t* d2_bg_dev, double dddmp, double d4_bg,
                            int nord, void* stream);

/* ---- FiniteVolumeTransport.__call__ without damping (fvtp2d.py:262-345).  x/y_mass_flux may be
 * NULL (area fluxes are used as unit fluxes).  hord in {5, 6, 8}.  nlev = number of levels
 * processed (nk, or nk+1 for interface fields).  q's corner halos are NOT rewritten: corner reads
 * go through the copy_corners index map, which yields identical fluxes. */
int pace_fvtp2d(const pace_geom_t* geom, const pace_metrics_t* met, const pace_real_t* q, const pace_real_t* crx,
                const pace_real_t* cry, const pace_real_t* x_area_flux, const pace_real_t* y_area_flux,
                pace_real_t* q_x_flux, pace_real_t* q_y_flux, const pace_real_t* x_mass_flux,
                const pace_real_t* y_mass_flux, int hord, int nlev, void* stream);

/* ---- The fused form d_sw uses for q_con and pt (d_sw.py:1075-1117): FiniteVolumeTransport with mass fluxes AND its
 * DelnFlux(mass = delp) (fvtp2d.py:262-345), followed by apply_fluxes (d_sw.py:122-145):
 *   qout = q * delp + flux_increment(q_x_flux, q_y_flux) * rarea      on the compute domain,
 * in ONE kernel; the flux fields never reach memory.  damp_k / nord_k as for pace_delnflux.  qout must not alias q. */
int pace_fvtp2d_update(const pace_geom_t* geom, const pace_metrics_t* met, const pace_real_t* q,
                       const pace_real_t* crx, const pace_real_t* cry, const pace_real_t* x_area_flux,
                       const pace_real_t* y_area_flux, const pace_real_t* x_mass_flux, const pace_real_t* y_mass_flux,
                       const pace_real_t* delp, const pace_real_t* damp_k, const pace_real_t* nord_k, int nmax,
                       pace_real_t* qout, int hord, int nlev, void* stream);

/* ---- DelnFluxNoSG.__call__ (delnflux.py:1050-1261): damping fluxes fx2, fy2 of q.
 * nord_k, damp_k: DEVICE arrays, one entry per level (see DESIGN.md for how the reference's
 * nord0..nord3 externals map to per-level values).  If mass_given != 0, d2 starts from q
 * (copy_stencil_interval) instead of damp*q.  nmax = max(nord_k). */
int pace_delnflux_nosg(const pace_geom_t* geom, const pace_metrics_t* met, const pace_real_t* q, pace_real_t* fx2,
                       pace_real_t* fy2, const pace_real_t* damp_k, const pace_real_t* nord_k, int nmax,
                       int mass_given, int nlev, void* stream);

/* ---- DelnFlux.__call__ (delnflux.py:945-1047): fx, fy += damping flux (mass-weighted if
 * mass != NULL).  damp_k = (damp_c*da_min)^(nord+1) per level (calc_damp, delnflux.py:21-38). */
int pace_delnflux(const pace_geom_t* geom, const pace_metrics_t* met, const pace_real_t* q, pace_real_t* fx,
                  pace_real_t* fy, const pace_real_t* mass, const pace_real_t* damp_k, const pace_real_t* nord_k,
                  int nmax, int nlev, void* stream);

/* ---- AGrid2BGridFourthOrder.__call__ (a2b_ord4.py:668-761) on levels [k0, k1). ---- */
int pace_a2b_ord4(const pace_geom_t* geom, const pace_metrics_t* met, pace_real_t* qin, pace_real_t* qout, int k0,
                  int k1, int replace, void* stream);

/* ---- DGridShallowWaterLagrangianDynamics.__call__ (d_sw.py:935-1237).
 * workspace: DEVICE scratch of pace_d_sw_workspace_bytes() bytes, owned by the caller for the
 * object's lifetime (the reference allocates its temporaries in __init__, d_sw.py:765-784);
 * it also carries uc_contra / vc_contra between calls.  col holds HOST arrays. */
int64_t pace_d_sw_workspace_bytes(const pace_geom_t* geom);
/* 1 if pace_d_sw* accept separate outputs (pace_dsw_config_t::delp_out ...) for this geometry and these orders, else 0. */
int pace_d_sw_pingpong_supported(const pace_geom_t* geom, const pace_dsw_config_t* cfg);
/* 1 if whole-d_sw calls also accept separate outputs of the winds (pace_dsw_config_t::u_out, v_out) -- the library then updates
 * the winds in the kernel that transports the scalars, and pace_d_sw_overlapped has nothing left for its side stream --, else 0.
 * (The damping orders nord_v, nord_w, nord_t of the column namelist must be <= 2, as get_column_namelist makes them,
 * d_sw.py:633-683; otherwise the call returns PACE_ERR_UNSUPPORTED.) */
int pace_d_sw_wind_outputs_supported(const pace_geom_t* geom, const pace_dsw_config_t* cfg);
/* The two queries above AND the condition on the column namelist in one call (what pace_d_sw* really accepts for this object):
 * 0 = no separate outputs, 1 = the four scalars', 3 = the scalars' and the winds'.  col: the HOST arrays given to pace_d_sw_prepare
 * (get_column_namelist, d_sw.py:633-683). */
int pace_d_sw_outputs_supported(const pace_geom_t* geom, const pace_column_t* col, const pace_dsw_config_t* cfg);
/* Once per object, after zero-filling the workspace: uploads the column namelist (synchronises). */
int pace_d_sw_prepare(const pace_geom_t* geom, const pace_column_t* col, void* workspace, void* stream);
int pace_d_sw(const pace_geom_t* geom, const pace_metrics_t* met, const pace_column_t* col,
              const pace_dsw_config_t* cfg, void* workspace, pace_real_t* delpc, pace_real_t* delp, pace_real_t* pt,
              pace_real_t* u, pace_real_t* v, pace_real_t* w, pace_real_t* uc, pace_real_t* vc, const pace_real_t* ua,
              const pace_real_t* va, pace_real_t* divgd, pace_real_t* mfx, pace_real_t* mfy, pace_real_t* cx,
              pace_real_t* cy, pace_real_t* crx, pace_real_t* cry, pace_real_t* xfx, pace_real_t* yfx,
              pace_real_t* q_con, const pace_real_t* zh, pace_real_t* heat_source, pace_real_t* diss_est, double dt,
              void* stream);

/* The two halves of pace_d_sw, same arguments.  pace_d_sw_transport: flux preparation and the transport of delp, w,
 * q_con, pt (d_sw.py:935-1117) -- everything updatedzd / riem_solver3 read.  pace_d_sw_winds: the rest
 * (d_sw.py:1119-1237); it reads only what the first half left behind, so it may be launched on a second stream
 * and run concurrently with the vertical solver.  pace_d_sw == transport followed by winds on one stream. */
int pace_d_sw_transport(const pace_geom_t* geom, const pace_metrics_t* met, const pace_column_t* col,
                        const pace_dsw_config_t* cfg, void* workspace, pace_real_t* delpc, pace_real_t* delp,
                        pace_real_t* pt, pace_real_t* u, pace_real_t* v, pace_real_t* w, pace_real_t* uc,
                        pace_real_t* vc, const pace_real_t* ua, const pace_real_t* va, pace_real_t* divgd,
                        pace_real_t* mfx, pace_real_t* mfy, pace_real_t* cx, pace_real_t* cy, pace_real_t* crx,
                        pace_real_t* cry, pace_real_t* xfx, pace_real_t* yfx, pace_real_t* q_con,
                        const pace_real_t* zh, pace_real_t* heat_source, pace_real_t* diss_est, double dt,
                        void* stream);
int pace_d_sw_winds(const pace_geom_t* geom, const pace_metrics_t* met, const pace_column_t* col,
                    const pace_dsw_config_t* cfg, void* workspace, pace_real_t* delpc, pace_real_t* delp,
                    pace_real_t* pt, pace_real_t* u, pace_real_t* v, pace_real_t* w, pace_real_t* uc, pace_real_t* vc,
                    const pace_real_t* ua, const pace_real_t* va, pace_real_t* divgd, pace_real_t* mfx,
                    pace_real_t* mfy, pace_real_t* cx, pace_real_t* cy, pace_real_t* crx, pace_real_t* cry,
                    pace_real_t* xfx, pace_real_t* yfx, pace_real_t* q_con, const pace_real_t* zh,
                    pace_real_t* heat_source, pace_real_t* diss_est, double dt, void* stream);

/* Finer split for callers that overlap on two streams (same arguments after `phases`).  Bit mask: 1 = flux preparation
 * (fxadv), 2 = transport of delp, w, q_con, pt, 4 = winds A (kinetic energy ... vorticity damping fluxes), 8 = winds B
 * (dissipative heating, final u/v update).  2 and 4 depend only on 1 and use disjoint workspace fields; 8 needs 2 and
 * 4.  pace_d_sw_transport == phases 3, pace_d_sw_winds == phases 12, pace_d_sw == 15.  Instead of 1: 16 = the part of the flux
 * preparation that reads no halo value of uc / vc (the box [is+2, ie-1] x [js+2, je-1]) -- it may run while the uc / vc halo
 * exchange is in flight (dyn_core.py:817-820) --, 32 = the rest of it, after the exchange.  Instead of 4: 64 = kinetic energy and
 * relative vorticity (they need only the flux preparation), 128 = the rest of winds A.
 * Where pace_d_sw_wind_outputs_supported() and the call runs 2, 4 and 8 together, the library orders the work differently: kinetic
 * energy, vorticity and divergence damping first, then ONE kernel that transports the scalars and the vorticity, updates the
 * winds and forms the dissipative heating.  256 (alone; a measurement aid): only that kernel, on the kinetic energy /
 * vorticities a previous call left in the workspace. */
int pace_d_sw_phases(int phases, const pace_geom_t* geom, const pace_metrics_t* met, const pace_column_t* col,
                     const pace_dsw_config_t* cfg, void* workspace, pace_real_t* delpc, pace_real_t* delp,
                     pace_real_t* pt, pace_real_t* u, pace_real_t* v, pace_real_t* w, pace_real_t* uc,
                     pace_real_t* vc, const pace_real_t* ua, const pace_real_t* va, pace_real_t* divgd,
                     pace_real_t* mfx, pace_real_t* mfy, pace_real_t* cx, pace_real_t* cy, pace_real_t* crx,
                     pace_real_t* cry, pace_real_t* xfx, pace_real_t* yfx, pace_real_t* q_con, const pace_real_t* zh,
                     pace_real_t* heat_source, pace_real_t* diss_est, double dt, void* stream);

/* The same four phases in ONE call, the wind half on `side_stream` (prep: 1 = the whole flux preparation, 32 = its frame after
 * phases 16): flux preparation and scalars on `stream`, winds A on the side stream after the preparation, winds B there after the
 * scalars.  The three events are the caller's (hipEvent_t; e.g. torch.cuda.Event.cuda_event); ev_done is recorded on the side
 * stream at the end -- the caller's stream must wait for it before u, v, uc, vc, heat_source, diss_est, delpc or divgd are used. */
int pace_d_sw_overlapped(int prep, const pace_geom_t* geom, const pace_metrics_t* met, const pace_column_t* col,
                     const pace_dsw_config_t* cfg, void* workspace, pace_real_t* delpc, pace_real_t* delp,
                     pace_real_t* pt, pace_real_t* u, pace_real_t* v, pace_real_t* w, pace_real_t* uc,
                     pace_real_t* vc, const pace_real_t* ua, const pace_real_t* va, pace_real_t* divgd,
                     pace_real_t* mfx, pace_real_t* mfy, pace_real_t* cx, pace_real_t* cy, pace_real_t* crx,
                     pace_real_t* cry, pace_real_t* xfx, pace_real_t* yfx, pace_real_t* q_con, const pace_real_t* zh,
                     pace_real_t* heat_source, pace_real_t* diss_est, double dt, void* stream,
                         void* side_stream, void* ev_prep, void* ev_scalars, void* ev_done);

/* ---- NonhydrostaticVerticalSolver.__call__ (riem_solver3.py:208-321), compute domain.
 * zs, ws: 2-D.  workspace: pace_riem_solver3_workspace_bytes() bytes of DEVICE scratch. */
int64_t pace_riem_solver3_workspace_bytes(const pace_geom_t* geom);
int pace_riem_solver3(const pace_geom_t* geom, void* workspace, int last_call, double dt, const pace_real_t* cappa,
                      double ptop, const pace_real_t* zs, const pace_real_t* ws, pace_real_t* delz,
                      const pace_real_t* q_con, const pace_real_t* delp, const pace_real_t* pt, pace_real_t* zh,
                      pace_real_t* pe, pace_real_t* ppe, pace_real_t* pk3, pace_real_t* pk, pace_real_t* peln,
                      pace_real_t* w, double p_fac, void* stream);

/* ---- CGridShallowWaterDynamics.__call__ (fv3core/pace/fv3core/stencils/c_sw.py:599-766), including
 * DGrid2AGrid2CGridVectors (d2a2c_vect.py:529-655).  delpc / ptc are the class attributes the reference
 * exposes (c_sw.py:497-502, read by dyn_core.py:795-800).  workspace: pace_c_sw_workspace_bytes().
 * delp, pt, w are inputs -- except the two cells next to each corner of the tile's halo, which end as the reference's in-place
 * corner fills of the C-grid transport leave them (c_sw.py:483-600; TranslateC_SW compares the three over the whole storage). */
int64_t pace_c_sw_workspace_bytes(const pace_geom_t* geom);
int pace_c_sw(const pace_geom_t* geom, const pace_metrics_t* met, void* workspace, pace_real_t* delpc,
              pace_real_t* ptc, pace_real_t* delp, pace_real_t* pt, const pace_real_t* u,
              const pace_real_t* v, pace_real_t* w, pace_real_t* uc, pace_real_t* vc, pace_real_t* ua,
              pace_real_t* va, pace_real_t* ut, pace_real_t* vt, pace_real_t* divgd, pace_real_t* omga, double dt2,
              int nord, void* stream);
/* The same in two parts around the u / v halo exchange in front of c_sw (dyn_core.py:744-745; an extension for overlapping
 * that exchange with compute): part 1 = the points of its first pass that read no halo value of u / v (the box
 * [is+1, ie-1] x [js+1, je-1]); part 2 = everything else, after the exchange; part 0 = pace_c_sw.  Same arguments. */
int pace_c_sw_part(int part, const pace_geom_t* geom, const pace_metrics_t* met, void* workspace, pace_real_t* delpc,
                   pace_real_t* ptc, pace_real_t* delp, pace_real_t* pt, const pace_real_t* u,
                   const pace_real_t* v, pace_real_t* w, pace_real_t* uc, pace_real_t* vc, pace_real_t* ua,
                   pace_real_t* va, pace_real_t* ut, pace_real_t* vt, pace_real_t* divgd, pace_real_t* omga,
                   double dt2, int nord, void* stream);
/* ---- DGrid2AGrid2CGridVectors.__call__ alone (d2a2c_vect.py:529-655), dord4 = True; same workspace. */
int pace_d2a2c_vect(const pace_geom_t* geom, const pace_metrics_t* met, void* workspace, pace_real_t* uc,
                    pace_real_t* vc, const pace_real_t* u, const pace_real_t* v, pace_real_t* ua, pace_real_t* va,
                    pace_real_t* utc, pace_real_t* vtc, void* stream);

/* ---- Sim1Solver.__call__ (sim1_solver.py:144-219) as a class of its own: the semi-implicit vertical solver on the compute
 * domain widened by n_halo (the reference builds it with n_halo = 0 for riem_solver3 and 1 for riem_solver_c).  gamma, cp3,
 * delta_mass, pm, pem, potential_temperature in; pe out (nk + 1 interfaces); w, dz inout; ws 2-D.  workspace:
 * pace_sim1_solver_workspace_bytes.  Inside pace_riem_solver3 / pace_riem_solver_c the same arithmetic runs fused. */
int64_t pace_sim1_solver_workspace_bytes(const pace_geom_t* geom);
int pace_sim1_solver(const pace_geom_t* geom, void* workspace, int n_halo, double dt, double p_fac,
                     const pace_real_t* gamma, const pace_real_t* cp3, pace_real_t* pe, const pace_real_t* delta_mass,
                     const pace_real_t* pm, const pace_real_t* pem, pace_real_t* w, pace_real_t* dz,
                     const pace_real_t* potential_temperature, const pace_real_t* ws, void* stream);

/* ---- NonhydrostaticVerticalSolverCGrid.__call__ (riem_solver_c.py:160-250), compute domain +- 1. */
int64_t pace_riem_solver_c_workspace_bytes(const pace_geom_t* geom);
int pace_riem_solver_c(const pace_geom_t* geom, void* workspace, double dt2, const pace_real_t* cappa, double ptop,
                       const pace_real_t* hs, const pace_real_t* ws, const pace_real_t* ptc, const pace_real_t* q_con,
                       const pace_real_t* delpc, pace_real_t* gz, pace_real_t* pef, const pace_real_t* w3,
                       double p_fac, void* stream);

/* ---- UpdateGeopotentialHeightOnCGrid.__call__ (updatedzc.py:172-207).  dp_ref: DEVICE K-array (nk). */
int64_t pace_updatedzc_workspace_bytes(const pace_geom_t* geom);
int pace_updatedzc(const pace_geom_t* geom, const pace_metrics_t* met, void* workspace, const pace_real_t* dp_ref,
                   const pace_real_t* zs, const pace_real_t* ut, const pace_real_t* vt, pace_real_t* gz,
                   pace_real_t* ws, double dt, void* stream);

/* ---- UpdateHeightOnDGrid.__call__ (updatedzd.py:281-356).  K-dependent constants: gk/beta/gamma from
 * cubic_spline_interpolation_constants (updatedzd.py:129-154) as DEVICE arrays of nk, the four scalars the
 * interpolation stencil derives from them (:180-192), and the DelnFluxNoSG column arguments on nk+1 levels
 * (damp = column_namelist["damp_vt"], nord = nord_v expanded per level), DEVICE arrays. */
typedef struct {
  const pace_real_t *gk, *beta, *gamma;
  double xt1_top, a_bot, xt1_bot, xt2_bot;
  const pace_real_t *damp, *nord;
  int32_t nmax, pad_;
} pace_updatedzd_k_t;
int64_t pace_updatedzd_workspace_bytes(const pace_geom_t* geom);
int pace_updatedzd(const pace_geom_t* geom, const pace_metrics_t* met, void* workspace, const pace_updatedzd_k_t* kc,
                   const pace_real_t* surface_height, pace_real_t* height, const pace_real_t* courant_number_x,
                   const pace_real_t* courant_number_y, const pace_real_t* x_area_flux,
                   const pace_real_t* y_area_flux, pace_real_t* ws, double dt, int hord_tm, void* stream);

/* ---- dyn_core.py stencils: gz_from_surface_height_and_thicknesses (:83-96, compute domain),
 * compute_geopotential (:115-117, halo 2, nk+1 levels), basic.copy_defn as used at dyn_core.py:773-781
 * (full domain, nk+1 levels), p_grad_c_stencil (:120-171, hydrostatic = False). */
int pace_zero_data(const pace_geom_t* geom, pace_real_t* mfxd, pace_real_t* mfyd, pace_real_t* cxd, pace_real_t* cyd,
                   pace_real_t* heat_source, pace_real_t* diss_estd, int first_timestep, void* stream);  /* dyn_core.py:51-80 */
int pace_interface_pressure_from_toa_pressure_and_thickness(const pace_geom_t* geom, const pace_real_t* delp,
                                                            pace_real_t* pem, double ptop, void* stream);  /* :99-112 */
int pace_gz_from_surface_height_and_thicknesses(const pace_geom_t* geom, const pace_real_t* zs,
                                                const pace_real_t* delz, pace_real_t* gz, void* stream);
int pace_compute_geopotential(const pace_geom_t* geom, const pace_real_t* zh, pace_real_t* gz, void* stream);
int pace_copy(const pace_geom_t* geom, const pace_real_t* src, pace_real_t* dst, void* stream);
int pace_p_grad_c(const pace_geom_t* geom, const pace_metrics_t* met, pace_real_t* uc, pace_real_t* vc,
                  const pace_real_t* delpc, const pace_real_t* pkc, const pace_real_t* gz, double dt2, void* stream);

/* ---- NonHydrostaticPressureGradient.__call__ (nh_p_grad.py:187-255). */
int64_t pace_nh_p_grad_workspace_bytes(const pace_geom_t* geom);
int pace_nh_p_grad(const pace_geom_t* geom, const pace_metrics_t* met, void* workspace, pace_real_t* u,
                   pace_real_t* v, pace_real_t* pp, pace_real_t* gz, pace_real_t* pk3, pace_real_t* delp, double dt,
                   double ptop, double akap, void* stream);

/* ---- pe_halo.edge_pe (pe_halo.py:6-34) and PK3Halo.__call__ (pk3_halo.py:55-69). */
int pace_edge_pe(const pace_geom_t* geom, pace_real_t* pe, const pace_real_t* delp, double ptop, void* stream);
int pace_pk3_halo(const pace_geom_t* geom, pace_real_t* pk3, const pace_real_t* delp, double ptop, double akap,
                  void* stream);

/* ---- RayleighDamping.__call__ (ray_fast.py:186-206).  dp, pfull: HOST K-arrays (nk). */
int pace_ray_fast(const pace_geom_t* geom, pace_real_t* u, pace_real_t* v, pace_real_t* w, const double* dp,
                  const double* pfull, double dt, double ptop, double rf_cutoff, double tau, int hydrostatic,
                  void* stream);

/* ---- HyperdiffusionDamping.__call__ (del2cubed.py:168-194) and apply_diffusive_heating
 * (temperature_adjust.py:8-43, first nlev levels of the compute domain). */
int64_t pace_del2cubed_workspace_bytes(const pace_geom_t* geom);
int pace_del2cubed(const pace_geom_t* geom, const pace_metrics_t* met, void* workspace, pace_real_t* qdel, double cd,
                   int nmax, void* stream);
int pace_apply_diffusive_heating(const pace_geom_t* geom, const pace_real_t* delp, const pace_real_t* delz,
                                 const pace_real_t* cappa, const pace_real_t* heat_source, pace_real_t* pt,
                                 double delt_time_factor, int nlev, void* stream);

/* ---- TracerAdvection (Fortran tracer_2d_1l): the stencils around FiniteVolumeTransport(hord = 8) in
 * fv3core/pace/fv3core/stencils/tracer_2d_1l.py -- flux_compute (:19-77), divide_fluxes_by_n_substeps (:80-106),
 * apply_mass_flux (:115-135), apply_tracer_flux (:138-158), swap_dp (:166-170).  The transport itself is pace_fvtp2d
 * with hord = 8 (monotone PPM, xppm.py:76-145,185-287). */
int pace_tracer_flux_compute(const pace_geom_t* geom, const pace_metrics_t* met, const pace_real_t* cx,
                             const pace_real_t* cy, pace_real_t* xfx, pace_real_t* yfx, void* stream);
int pace_tracer_divide_fluxes(const pace_geom_t* geom, pace_real_t* cxd, pace_real_t* xfx, pace_real_t* mfxd,
                              pace_real_t* cyd, pace_real_t* yfx, pace_real_t* mfyd, int n_split, void* stream);
int pace_apply_mass_flux(const pace_geom_t* geom, const pace_metrics_t* met, const pace_real_t* dp1,
                         const pace_real_t* x_mass_flux, const pace_real_t* y_mass_flux, pace_real_t* dp2,
                         void* stream);
int pace_apply_tracer_flux(const pace_geom_t* geom, const pace_metrics_t* met, pace_real_t* q, const pace_real_t* dp1,
                           const pace_real_t* fx, const pace_real_t* fy, const pace_real_t* dp2, void* stream);
int pace_swap_dp(const pace_geom_t* geom, pace_real_t* dp1, pace_real_t* dp2, void* stream);

/* ---- MapSingle (Fortran map_single / map1_ppm / map_scalar): fv3core/pace/fv3core/stencils/map_single.py:96-200 with
 * RemapProfile (cs_profile), remap_profile.py:566-681.  q1 is remapped in place from the nk layers bounded by the
 * interface values pe1 to those bounded by pe2 (both nk + 1 levels; pressure or log-pressure -- remapping.py:587-627).
 * kord: 9 or 10 (sign ignored, as the reference takes abs); iv: the reference's `mode` (-2 vertical velocity -- needs the
 * bottom value qs, a 2-D field --, -1 winds, 0 positive-definite tracers, 1 others, 2 as remap_profile.py:387-395).
 * xstag / ystag: the field is staggered in x / y and owns one more column / row (`dims` of the reference's constructor).
 * workspace: pace_map_single_workspace_bytes() of device memory (5 fields; the reference keeps 14). */
int64_t pace_map_single_workspace_bytes(const pace_geom_t* geom);
int pace_map_single(const pace_geom_t* geom, void* workspace, pace_real_t* q1, const pace_real_t* pe1,
                    const pace_real_t* pe2, const pace_real_t* qs, double qmin, int kord, int iv, int xstag,
                    int ystag, void* stream);

/* MapNTracer (Fortran mapn_tracer): fv3core/pace/fv3core/stencils/mapn_tracer.py:13-82 -- nq (<= 16) tracers that share
 * pe1 / pe2 are remapped (iv = 0, one kord) by ONE three-launch sequence instead of nq MapSingle calls.  tracers is a
 * HOST array of nq device pointers.  workspace: pace_mapn_tracer_workspace_bytes(geom, nq). */
int64_t pace_mapn_tracer_workspace_bytes(const pace_geom_t* geom, int nq);
int pace_mapn_tracer(const pace_geom_t* geom, void* workspace, pace_real_t* const* tracers, int nq,
                     const pace_real_t* pe1, const pace_real_t* pe2, int kord, void* stream);

/* FillNegativeTracerValues (Fortran fillz): fv3core/pace/fv3core/stencils/fillz.py:120-163 -- negative tracer masses
 * borrow from the layers above / below, then the column is rescaled; all nq tracers in one launch.  tracers: HOST array
 * of nq device pointers; dp2: layer thicknesses. */
int pace_fillz(const pace_geom_t* geom, pace_real_t* const* tracers, int nq, const pace_real_t* dp2, void* stream);

/* ---- LagrangianToEulerian: the stencils around the remaps (fv3core/pace/fv3core/stencils/remapping.py:286-695 calls them
 * in this order; pace_amd/fv3core/stencils/remapping.py is the host sequence).  Non-hydrostatic, kord_tm < 0, no saturation
 * adjustment.  water: HOST array of the six device pointers qvapor, qliquid, qrain, qsnow, qice, qgraupel; ak / bk: device
 * arrays of nk + 1 hybrid coefficients; ps: 2-D field.
 *   pace_l2e_prepare   = init_pe (:42-56) + moist_cv_pt_pressure (:85-171) + pn2_pk_delp (:174-193)
 *   pace_l2e_post      = undo_delz_adjust_and_copy_peln (:59-80) + moist_cv.moist_pkz (moist_cv.py:130-172)
 *   pace_l2e_pressures = pressures_mapu (dir 0, :196-227) / pressures_mapv (dir 1, :230-254)
 *   pace_l2e_finish    = update_ua + copy_from_below (:257-283), then moist_pt_last_step (moist_cv.py:84-122, last_step
 *                        != 0) or adjust_divide_stencil (pt / pkz) */
int pace_l2e_prepare(const pace_geom_t* geom, const pace_real_t* const* water, pace_real_t* q_con, pace_real_t* pt,
                     pace_real_t* cappa, pace_real_t* delp, pace_real_t* delz, const pace_real_t* pe,
                     pace_real_t* pe1, pace_real_t* pe2, const pace_real_t* ak, const pace_real_t* bk,
                     pace_real_t* dp2, pace_real_t* ps, pace_real_t* pn2, const pace_real_t* peln, pace_real_t* pk,
                     double ptop, double akap, double r_vir, void* stream);
int pace_l2e_post(const pace_geom_t* geom, const pace_real_t* const* water, pace_real_t* q_con, pace_real_t* pkz,
                  const pace_real_t* pt, pace_real_t* cappa, const pace_real_t* delp, pace_real_t* delz,
                  pace_real_t* peln, pace_real_t* pe0, const pace_real_t* pn2, double r_vir, void* stream);
int pace_l2e_pressures(const pace_geom_t* geom, int dir, const pace_real_t* pe, const pace_real_t* pe1,
                       const pace_real_t* ak, const pace_real_t* bk, pace_real_t* pe0, pace_real_t* pe3,
                       void* stream);
int pace_l2e_finish(const pace_geom_t* geom, const pace_real_t* const* water, pace_real_t* pe, const pace_real_t* pe2,
                    pace_real_t* pt, const pace_real_t* pkz, double r_vir, int last_step, void* stream);

/* ---- DynamicalCore (fv3core/pace/fv3core/stencils/fv_dynamics.py:92-624): the stencils it runs itself.  water: HOST
 * array of the six device pointers qvapor, qliquid, qrain, qsnow, qice, qgraupel.
 *   pace_fv_setup_pt  = moist_cv.fv_setup (moist_cv.py:175-234, moist_phys, nwat 6) + pt_to_potential_density_pt
 *                       (fv_dynamics.py:41-54): compute_preamble's two stencils in one pass
 *   pace_omega_from_w = fv_dynamics.py:57-67
 *   pace_neg_adj3     = AdjustNegativeTracerMixingRatio.__call__ (fv3core/pace/fv3core/stencils/neg_adj3.py:377-420),
 *                       non-hydrostatic: fix_neg_water, fillq(qgraupel), fillq(qrain), fix_water_vapor_down, fix_neg_cloud
 *   pace_c2l_ord      = CubedToLatLon's stencil (stencils/pace/stencils/c2l_ord.py:15-112), order 2 or 4 (order 4 expects
 *                       the halos of u, v updated); a11..a22: 2-D metric fields */
int pace_fv_setup_pt(const pace_geom_t* geom, pace_real_t* const* water, pace_real_t* q_con, pace_real_t* pkz,
                     pace_real_t* pt, pace_real_t* cappa, const pace_real_t* delp, const pace_real_t* delz,
                     pace_real_t* dp1, void* stream);
int pace_omega_from_w(const pace_geom_t* geom, const pace_real_t* delp, const pace_real_t* delz, const pace_real_t* w,
                      pace_real_t* omga, void* stream);
int pace_neg_adj3(const pace_geom_t* geom, pace_real_t* const* water, pace_real_t* qcld, pace_real_t* pt,
                  const pace_real_t* delp, void* stream);
int pace_c2l_ord(const pace_geom_t* geom, const pace_metrics_t* met, int order, const pace_real_t* u,
                 const pace_real_t* v, const pace_real_t* a11, const pace_real_t* a12, const pace_real_t* a21,
                 const pace_real_t* a22, pace_real_t* ua, pace_real_t* va, void* stream);

/* Per-stencil device implementations behind FrozenStencil (dsl/pace/dsl/stencil.py:395-434) for gtscript definitions that the
 * reference's Translate tests launch as stencils of their own: `id` selects the definition, `fields` are its field arguments in
 * the definition's order (2-D metric arguments come from `met`, K-fields are device arrays of nk + 1 entries), `scalars` its
 * float arguments, origin / domain the launch window (region statements write only inside it).
 *   PACE_ST_FLUX_CAPACITOR  d_sw.py:33-60     fields cx, cy, xflux, yflux, crx_adv, cry_adv, fx, fy
 *   PACE_ST_HEAT_DISS       d_sw.py:63-103    fields fx2, fy2, w, heat_source, diss_est, dw, damp_w (K), ke_bg (K); scalars dt
 *   PACE_ST_APPLY_FLUXES    d_sw.py:122-145   fields q, delp, gx, gy
 *   PACE_ST_UBKE / _VBKE    translate_d_sw.py:67-81 / 118-133 (d_sw.py interpolate_uc_vc_to_cell_corners)
 *                                             fields uc, vc, ut (vt), ub (vb); scalars dt5
 *   PACE_ST_COPY_CORNERS_X / _Y         corners.py:307-425   fields q_in, q_out
 *   PACE_ST_FILL_CORNERS_BGRID_X / _Y   corners.py:592-712   fields q_in, q_out
 *   PACE_ST_FILL_CORNERS_DGRID          corners.py:987-1151  fields x_in, x_out, y_in, y_out; scalars mysign
 *   PACE_ST_FILL_CORNERS_2CELLS_X / _Y  corners.py:170-177   fields q_out, q_in
 *   PACE_ST_XTP_U / _YTP_V   translate_xtp_u.py:13-23 / translate_ytp_v.py (xtp_u.py:9-91, ytp_v.py:9-91)
 *                            fields c (contravariant corner wind x dt), u (v), flux; scalars iord (5, 6 or 7)
 *   PACE_ST_MOIST_PT_LAST_STEP  moist_cv.py:84-118 (nwat = 6)  fields qvapor, qliquid, qrain, qsnow, qice, qgraupel, gz, pt, pkz;
 *                            scalars dtmp, r_vir
 *   PACE_ST_MOIST_PKZ        moist_cv.py:130-172 (nwat = 6; translate_moistcvpluspkz_2d.py:19)  fields qvapor, qliquid, qrain, qsnow,
 *                            qice, qgraupel, q_con, gz, cvm, pkz, pt, cappa, delp, delz; scalar r_vir
 *   PACE_ST_MOIST_PT         moist_cv.py:48-70 moist_pt_func as the stencil translate_moistcvpluspt_2d.py:9-50 builds  fields qvapor,
 *                            qliquid, qrain, qsnow, qice, qgraupel, q_con, pt, cappa, delp, delz; scalar r_vir */
enum {
  PACE_ST_FLUX_CAPACITOR = 1,
  PACE_ST_HEAT_DISS = 2,
  PACE_ST_APPLY_FLUXES = 3,
  PACE_ST_UBKE = 4,
  PACE_ST_VBKE = 5,
  PACE_ST_COPY_CORNERS_X = 6,
  PACE_ST_COPY_CORNERS_Y = 7,
  PACE_ST_FILL_CORNERS_BGRID_X = 8,
  PACE_ST_FILL_CORNERS_BGRID_Y = 9,
  PACE_ST_FILL_CORNERS_DGRID = 10,
  PACE_ST_FILL_CORNERS_2CELLS_X = 11,
  PACE_ST_FILL_CORNERS_2CELLS_Y = 12,
  PACE_ST_XTP_U = 13,
  PACE_ST_YTP_V = 14,
  PACE_ST_MOIST_PT_LAST_STEP = 15,
  PACE_ST_MOIST_PKZ = 16,
  PACE_ST_MOIST_PT = 17
};
int pace_stencil(const pace_geom_t* geom, const pace_metrics_t* met, int id, void* const* fields, int nfields, const double* scalars,
                 int nscalars, const int* origin, const int* domain, void* stream);


/* ---- Halo exchange pack / unpack: what HaloDataTransformer.async_pack / async_unpack do
 * (util/pace/util/halo_data_transformer.py:387-461 CPU, :560-921 GPU kernels), with the rotation
 * (rotate.py:4-50) and boundary slicing (_boundary_utils.py:58-95) folded into an affine index map.
 * Strip element (a, b, k), 0 <= a < na, 0 <= b < nb, 0 <= k < nk, lives at message offset (k*nb + b)*na + a
 * and at field index (i0 + a*di_a + b*di_b, j0 + a*dj_a + b*dj_b, k).  pack: buf = sign * field;
 * unpack: field = buf.  descs is a HOST array; field / buf are DEVICE pointers. */
typedef struct {
  pace_real_t* field;
  pace_real_t* buf;
  int32_t i0, j0, di_a, dj_a, di_b, dj_b;
  int32_t na, nb, nk, pad_;
  double sign;
} pace_halo_desc_t;
int pace_halo_pack(const pace_geom_t* geom, const pace_halo_desc_t* descs, int ndesc, void* stream);
int pace_halo_unpack(const pace_geom_t* geom, const pace_halo_desc_t* descs, int ndesc, void* stream);

const char* pace_version(void);
/* sizeof(pace_real_t) of this build: 8 (libpace_hip.so) or 4 (libpace_hip_f32.so). */
int pace_real_bytes(void);
/* Text of the last HIP error this library saw on the calling thread ("" if none). */
const char* pace_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
