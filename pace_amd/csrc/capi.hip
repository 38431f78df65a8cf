// extern "C" entry points of libpace_hip.so (declared in include/pace_hip.h).  Argument checking and
// struct unpacking only; kernels live in k_*.hip.
#include "common.h"
#include "kernels.h"

#include <cstdio>

thread_local char g_pace_err[256] = "";
int g_pace_sync_launches = getenv("PACE_SYNC_LAUNCHES") != nullptr;
void pace_set_err(const char* where, hipError_t e) {
  snprintf(g_pace_err, sizeof(g_pace_err), "%s: %s", where, hipGetErrorString(e));
}

// Entry prologue: argument check, then drop any stale "last error" another library (e.g. the caching
// allocator's event queries) left on this thread, so PACE_CHECK_LAUNCH only reports our own launches.
// The geometry is validated on every entry: the kernels address fields with 32-bit byte offsets from a uniform base
// (k_fvtp2d.hip), so a field must stay below 4 GB, and the strides must cover the (N+7) x (N+7) x (nk+1) storage.
static inline int geom_check(const pace_geom_t* g) {
  if (g == nullptr) return PACE_ERR_ARG;
  if (g->n < 1 || g->nk < 1) return PACE_ERR_ARG;
  if (g->sj < g->n + 7 || g->sk < (int64_t)g->sj * (g->n + 7)) return PACE_ERR_ARG;
  if ((int64_t)(g->nk + 1) * g->sk * (int64_t)sizeof(real) >= ((int64_t)1 << 32)) return PACE_ERR_UNSUPPORTED;
  return PACE_OK;
}

#define NEED(p)                              \
  if (!(p)) return PACE_ERR_ARG;             \
  {                                          \
    const int geom_rc_ = geom_check(geom);   \
    if (geom_rc_ != PACE_OK) return geom_rc_; \
  }                                          \
  (void)hipGetLastError()

static inline hipStream_t S(void* s) { return (hipStream_t)s; }

// PACE_LEGACY_COLUMN_SOLVERS=1 selects the round-1 thread-per-column kernels of k_riem3.hip (kept for more than 128 layers and
// for A/B measurements, and the form that walks a column's levels IN THE REFERENCE'S ORDER: the whole acoustic loop then agrees with
// the oracle to w 6e-9 / diss_estd 1e-7 where the scans' re-association gives 4e-7 / 2e-6 -- tests/helpers.py ACOUSTIC_TOL).
#include <cstdlib>
static bool legacy_column_solvers() {
  // (read at every call: tests switch between the two forms within one process)
  const char* e = getenv("PACE_LEGACY_COLUMN_SOLVERS");
  return e != nullptr && e[0] == '1';
}

extern "C" {

const char* pace_last_error(void) { return g_pace_err; }

int pace_real_bytes(void) { return (int)sizeof(real); }

const char* pace_version(void) {
#ifdef PACE_EMU
  return "pace_amd 0.1 (CPU emulation build -- test infrastructure only)";
#else
  return "pace_amd 0.1 (gfx950)";
#endif
}

int pace_fxadv(const pace_geom_t* geom, const pace_metrics_t* met, const real* uc, const real* vc, real* crx,
               real* cry, real* xfx, real* yfx, real* ut, real* vt, double dt, void* stream) {
  NEED(geom && met && uc && vc && crx && cry && xfx && yfx && ut && vt);
  return launch_fxadv(make_geo(geom), *met, uc, vc, crx, cry, xfx, yfx, ut, vt, dt, nullptr, nullptr, S(stream), 0, 1);
}

int pace_fvtp2d(const pace_geom_t* geom, const pace_metrics_t* met, const real* q, const real* crx,
                const real* cry, const real* xfx, const real* yfx, real* fx, real* fy, const real* xmf,
                const real* ymf, int hord, int nlev, void* stream) {
  NEED(geom && met && q && crx && cry && xfx && yfx && fx && fy);
  if ((xmf == nullptr) != (ymf == nullptr)) return PACE_ERR_ARG;
  if (nlev < 1 || nlev > geom->nk + 1) return PACE_ERR_ARG;
  return launch_fvtp2d(make_geo(geom), *met, q, crx, cry, xfx, yfx, fx, fy, xmf, ymf, hord, nlev, S(stream));
}

int pace_fvtp2d_update(const pace_geom_t* geom, const pace_metrics_t* met, const real* q, const real* crx,
                       const real* cry, const real* x_area_flux, const real* y_area_flux,
                       const real* x_mass_flux, const real* y_mass_flux, const real* delp, const real* damp_k,
                       const real* nord_k, int nmax, real* qout, int hord, int nlev, void* stream) {
  NEED(geom && met && q && crx && cry && x_area_flux && y_area_flux && x_mass_flux && y_mass_flux && delp && damp_k && nord_k && qout);
  if (nlev < 1 || nlev > geom->nk + 1 || qout == q) return PACE_ERR_ARG;
  FvDamp dp{};
  dp.damp_k = damp_k; dp.nord_k = nord_k; dp.nmax = nmax; dp.mass_given = 1; dp.mass = delp;
  dp.qout = qout; dp.amass = delp;
  return launch_transport(make_geo(geom), *met, q, crx, cry, x_area_flux, y_area_flux, nullptr, nullptr, x_mass_flux, y_mass_flux,
                          hord, nlev, 2, 1, dp, S(stream));
}

int pace_delnflux_nosg(const pace_geom_t* geom, const pace_metrics_t* met, const real* q, real* fx2, real* fy2,
                       const real* damp_k, const real* nord_k, int nmax, int mass_given, int nlev, void* stream) {
  NEED(geom && met && q && fx2 && fy2 && damp_k && nord_k);
  if (nlev < 1 || nlev > geom->nk + 1) return PACE_ERR_ARG;
  return launch_delnflux(make_geo(geom), *met, 0, q, fx2, fy2, nullptr, damp_k, nord_k, nmax, mass_given, nlev, S(stream));
}

int pace_delnflux(const pace_geom_t* geom, const pace_metrics_t* met, const real* q, real* fx, real* fy,
                  const real* mass, const real* damp_k, const real* nord_k, int nmax, int nlev, void* stream) {
  NEED(geom && met && q && fx && fy && damp_k && nord_k);
  if (nlev < 1 || nlev > geom->nk + 1) return PACE_ERR_ARG;
  return launch_delnflux(make_geo(geom), *met, mass ? 2 : 1, q, fx, fy, mass, damp_k, nord_k, nmax, mass ? 1 : 0, nlev,
                         S(stream));
}

int pace_a2b_ord4(const pace_geom_t* geom, const pace_metrics_t* met, real* qin, real* qout, int k0, int k1,
                  int replace, void* stream) {
  NEED(geom && met && qin && qout);
  if (k0 < 0 || k1 <= k0 || k1 > geom->nk + 1) return PACE_ERR_ARG;
  return launch_a2b_ord4(make_geo(geom), *met, qin, qout, k0, k1, replace, S(stream));
}

int pace_ppm(const pace_geom_t* geom, const pace_metrics_t* met, int axis, int iord, const real* q_in, const real* c,
             real* q_mean_advected, int i0, int j0, int k0, int ni, int nj, int nk, void* stream) {
  NEED(geom && met && q_in && c && q_mean_advected);
  if (axis < 0 || axis > 1 || ni < 1 || nj < 1 || nk < 1 || k0 < 0 || k0 + nk > geom->nk + 1) return PACE_ERR_ARG;
  // the window must leave the three cells an interface needs on either side inside the storage
  const int lo_i = axis == 0 ? 3 : 0, lo_j = axis == 1 ? 3 : 0;
  if (i0 < lo_i || j0 < lo_j || i0 + ni > geom->n + 7 - (axis == 0 ? 2 : 0) || j0 + nj > geom->n + 7 - (axis == 1 ? 2 : 0))
    return PACE_ERR_ARG;
  return launch_ppm1d(make_geo(geom), *met, axis, iord, q_in, c, q_mean_advected, i0, j0, k0, ni, nj, nk, S(stream));
}

int64_t pace_divergence_damping_workspace_bytes(const pace_geom_t* geom) {
  return geom ? (int64_t)2 * geom->sk * (geom->nk + 1) * (int64_t)sizeof(real) : 0;
}

int pace_divergence_damping(const pace_geom_t* geom, const pace_metrics_t* met, void* workspace, const real* u, const real* v,
                            const real* va, real* damped_rel_vort_bgrid, const real* ua, real* divg_d, real* vc,
                            real* uc, real* delpc, real* ke, const real* rel_vort_agrid, double dt,
                            const double* nord_col_host, const real* d2_bg_dev, double dddmp, double d4_bg, int nord,
                            void* stream) {
  NEED(geom && met && workspace && u && v && va && damped_rel_vort_bgrid && ua && divg_d && vc && uc && delpc && ke);
  NEED(rel_vort_agrid && nord_col_host && d2_bg_dev);
  const Geo g = make_geo(geom);
  // the column is split at the first level with nord > 0 (divergence_damping.py:307-331)
  int kstart = 0, nonzero_nord = nord;
  for (int k = 0; k < g.nk; ++k)
    if (nord_col_host[k] > 0) {
      kstart = k;
      nonzero_nord = (int)nord_col_host[k];
      break;
    }
  if (nonzero_nord < 0 || nonzero_nord > 3) return PACE_ERR_UNSUPPORTED;
  real* da = (real*)workspace;
  real* db = da + (long)g.sk * (g.nk + 1);
  return launch_divergence_damping(g, *met, u, v, va, damped_rel_vort_bgrid, ua, divg_d, vc, uc, delpc, ke, rel_vort_agrid, dt,
                                   d2_bg_dev, kstart, nonzero_nord, dddmp, d4_bg, da, db, S(stream));
}

int64_t pace_d_sw_workspace_bytes(const pace_geom_t* geom) { return geom ? dsw_workspace_bytes(make_geo(geom)) : 0; }

int pace_d_sw_pingpong_supported(const pace_geom_t* geom, const pace_dsw_config_t* cfg) {
  return (geom && cfg && dsw_pingpong_supported(make_geo(geom), cfg)) ? 1 : 0;
}
int pace_d_sw_wind_outputs_supported(const pace_geom_t* geom, const pace_dsw_config_t* cfg) {
  return (geom && cfg && dsw_winds_in_scalars(make_geo(geom), cfg)) ? 1 : 0;
}

int pace_d_sw_outputs_supported(const pace_geom_t* geom, const pace_column_t* col, const pace_dsw_config_t* cfg) {
  // the ONE predicate of what launch_d_sw accepts: the two queries above plus the condition on the column namelist they cannot see
  if (!geom || !col || !cfg || !col->nord_v || !col->nord_w || !col->nord_t) return 0;
  const Geo g = make_geo(geom);
  for (int k = 0; k < g.nk; ++k)
    if (col->nord_v[k] > 2 || col->nord_w[k] > 2 || col->nord_t[k] > 2) return 0;
  if (!dsw_pingpong_supported(g, cfg)) return 0;
  return dsw_winds_in_scalars(g, cfg) ? 3 : 1;
}

int pace_d_sw_prepare(const pace_geom_t* geom, const pace_column_t* col, void* workspace, void* stream) {
  NEED(geom && col && workspace);
  return dsw_prepare(make_geo(geom), col, workspace, S(stream));
}

static int d_sw_entry(int phases, const pace_geom_t* geom, const pace_metrics_t* met, const pace_column_t* col,
                      const pace_dsw_config_t* cfg, void* workspace, real* delpc, real* delp, real* pt, real* u,
                      real* v, real* w, real* uc, real* vc, const real* ua, const real* va, real* divgd,
                      real* mfx, real* mfy, real* cx, real* cy, real* crx, real* cry, real* xfx, real* yfx,
                      real* q_con, const real* zh, real* heat_source, real* diss_est, double dt, void* stream) {
  NEED(geom && met && col && cfg && workspace);
  NEED(delpc && delp && pt && u && v && w && uc && vc && ua && va && divgd && mfx && mfy && cx && cy);
  NEED(crx && cry && xfx && yfx && q_con && heat_source && diss_est);
  if (cfg->struct_bytes != (int32_t)sizeof(pace_dsw_config_t)) return PACE_ERR_ARG;  // built against another header
  {  // separate outputs: all four or none, none of them an input
    const int given = (cfg->delp_out != nullptr) + (cfg->pt_out != nullptr) + (cfg->w_out != nullptr) + (cfg->q_con_out != nullptr);
    if (given != 0 && given != 4) return PACE_ERR_ARG;
    if (given && (cfg->delp_out == delp || cfg->pt_out == pt || cfg->w_out == w || cfg->q_con_out == q_con)) return PACE_ERR_ARG;
    if (given && ((((uintptr_t)cfg->delp_out | (uintptr_t)cfg->pt_out | (uintptr_t)cfg->w_out | (uintptr_t)cfg->q_con_out) & 15) != 0))
      return PACE_ERR_ARG;  // (16-byte rows, as the header says)
    // ... of the winds: both or none, only with the four, only in a call that runs scalars and winds together
    const int winds = (cfg->u_out != nullptr) + (cfg->v_out != nullptr);
    if (winds == 1 || (winds && !given) || (winds && (cfg->u_out == u || cfg->v_out == v || cfg->u_out == cfg->v_out))) return PACE_ERR_ARG;
    // (they are written by the kernel of phases 2 + 4 + 8 / 256; a call without 2 and 8 does not touch the winds' outputs)
    if (winds && ((phases & 2) || (phases & 8)) && !((phases & 2) && (phases & 4) && (phases & 8))) return PACE_ERR_ARG;
  }
  return launch_d_sw(make_geo(geom), *met, col, cfg, workspace, delpc, delp, pt, u, v, w, uc, vc, ua, va, divgd, mfx, mfy,
                     cx, cy, crx, cry, xfx, yfx, q_con, zh, heat_source, diss_est, dt, phases, S(stream));
}

#define DSW_PARAMS                                                                                                        \
  const pace_geom_t *geom, const pace_metrics_t *met, const pace_column_t *col, const pace_dsw_config_t *cfg,            \
      void *workspace, real *delpc, real *delp, real *pt, real *u, real *v, real *w, real *uc, real *vc,  \
      const real *ua, const real *va, real *divgd, real *mfx, real *mfy, real *cx, real *cy, real *crx,   \
      real *cry, real *xfx, real *yfx, real *q_con, const real *zh, real *heat_source, real *diss_est,       \
      double dt, void *stream
#define DSW_ARGS_                                                                                                          \
  geom, met, col, cfg, workspace, delpc, delp, pt, u, v, w, uc, vc, ua, va, divgd, mfx, mfy, cx, cy, crx, cry, xfx, yfx,  \
      q_con, zh, heat_source, diss_est, dt, stream

int pace_d_sw(DSW_PARAMS) { return d_sw_entry(15, DSW_ARGS_); }
int pace_d_sw_transport(DSW_PARAMS) { return d_sw_entry(3, DSW_ARGS_); }
int pace_d_sw_winds(DSW_PARAMS) { return d_sw_entry(12, DSW_ARGS_); }
int pace_d_sw_phases(int phases, DSW_PARAMS) {
  if (phases < 1 || phases > 256 || ((phases & 1) && (phases & 48)) || ((phases & 4) && (phases & 192)) || ((phases & 256) && phases != 256))
    return PACE_ERR_ARG;  // (1 and 16 / 32, 4 and 64 / 128 are alternatives; 256 stands alone)
  return d_sw_entry(phases, DSW_ARGS_);
}

// d_sw with its wind half on a second stream, in ONE call (the host layer used to make four calls and three event operations
// per d_sw; the host time per substep, 105 - 120 us, is the runtime's launches and did not change).  `prep`: 1 = the whole flux preparation, 32 = its frame (the interior was
// started with phases 16).  Events: the caller's (e.g. torch.cuda.Event.cuda_event), recorded here; afterwards the caller makes
// its stream wait for ev_done before it touches u, v, uc, vc, heat_source, diss_est, delpc or divgd (pace_amd ... d_sw.py join()).
int pace_d_sw_overlapped(int prep, DSW_PARAMS, void* side_stream, void* ev_prep, void* ev_scalars, void* ev_done) {
#ifdef PACE_EMU
  (void)side_stream; (void)ev_prep; (void)ev_scalars; (void)ev_done;
  if (prep != 1 && prep != 32) return PACE_ERR_ARG;
  return d_sw_entry(prep | 14, DSW_ARGS_);
#else
  NEED(side_stream && ev_prep && ev_scalars && ev_done);
  if (prep != 1 && prep != 32) return PACE_ERR_ARG;
  hipStream_t main_s = S(stream), side_s = S(side_stream);
  hipEvent_t e_prep = (hipEvent_t)ev_prep, e_scal = (hipEvent_t)ev_scalars, e_done = (hipEvent_t)ev_done;
  int rc;
  if (geom && cfg && dsw_winds_in_scalars(make_geo(geom), cfg)) {
    // the winds are the last pass of the kernel that transports the scalars: one stream, nothing left for the side stream (the
    // kinetic energy there next to vorticity + divergence damping measured slower: profiles/r05_experiments x05)
    if ((rc = d_sw_entry(prep | 14, DSW_ARGS_))) return rc;
    if (hipEventRecord(e_done, main_s) != hipSuccess) return PACE_ERR_LAUNCH;
    return PACE_OK;
  }
  if ((rc = d_sw_entry(prep, DSW_ARGS_))) return rc;  // flux preparation on the caller's stream
  if (hipEventRecord(e_prep, main_s) != hipSuccess || hipStreamWaitEvent(side_s, e_prep, 0) != hipSuccess) return PACE_ERR_LAUNCH;
  {
    void* stream = side_stream;  // winds A next to the scalar transports
    if ((rc = d_sw_entry(4, DSW_ARGS_))) return rc;
  }
  if ((rc = d_sw_entry(2, DSW_ARGS_))) return rc;  // the scalars on the caller's stream
  if (hipEventRecord(e_scal, main_s) != hipSuccess || hipStreamWaitEvent(side_s, e_scal, 0) != hipSuccess) return PACE_ERR_LAUNCH;
  {
    void* stream = side_stream;  // winds B (they need the new delp) next to whatever the caller launches next
    if ((rc = d_sw_entry(8, DSW_ARGS_))) return rc;
  }
  if (hipEventRecord(e_done, side_s) != hipSuccess) return PACE_ERR_LAUNCH;
  return PACE_OK;
#endif
}

int64_t pace_riem_solver3_workspace_bytes(const pace_geom_t* geom) {
  return geom ? riem3_workspace_bytes(make_geo(geom)) : 0;
}

int pace_riem_solver3(const pace_geom_t* geom, void* workspace, int last_call, double dt, const real* cappa,
                      double ptop, const real* zs, const real* ws, real* delz, const real* q_con,
                      const real* delp, const real* pt, real* zh, real* pe, real* ppe, real* pk3, real* pk,
                      real* peln, real* w, double p_fac, void* stream) {
  NEED(geom && workspace && cappa && zs && ws && delz && q_con && delp && pt && zh && pe && ppe && pk3 && pk && peln && w);
  const Geo g = make_geo(geom);
  if (riem_column_supported(g) && !legacy_column_solvers())
    return launch_riem_solver3_column(g, last_call, dt, cappa, ptop, zs, ws, delz, q_con, delp, pt, zh, pe, ppe, pk3, pk, peln, w,
                                      p_fac, S(stream));
  return launch_riem_solver3(g, workspace, last_call, dt, cappa, ptop, zs, ws, delz, q_con, delp, pt, zh, pe, ppe, pk3, pk, peln,
                             w, p_fac, S(stream));
}

int64_t pace_c_sw_workspace_bytes(const pace_geom_t* geom) { return geom ? csw_workspace_bytes(make_geo(geom)) : 0; }

int pace_c_sw(const pace_geom_t* geom, const pace_metrics_t* met, void* workspace, real* delpc, real* ptc,
              real* delp, real* pt, const real* u, const real* v, real* w, real* uc,
              real* vc, real* ua, real* va, real* ut, real* vt, real* divgd, real* omga, double dt2,
              int nord, void* stream) {
  NEED(geom && met && workspace && delpc && ptc && delp && pt && u && v && w && uc && vc && ua && va && ut && vt && divgd && omga);
  return launch_c_sw(make_geo(geom), *met, workspace, delpc, ptc, delp, pt, u, v, w, uc, vc, ua, va, ut, vt, divgd, omga, dt2,
                     nord, S(stream));
}

int pace_c_sw_part(int part, const pace_geom_t* geom, const pace_metrics_t* met, void* workspace, real* delpc, real* ptc,
                   real* delp, real* pt, const real* u, const real* v, real* w, real* uc,
                   real* vc, real* ua, real* va, real* ut, real* vt, real* divgd, real* omga, double dt2,
                   int nord, void* stream) {
  NEED(geom && met && workspace && delpc && ptc && delp && pt && u && v && w && uc && vc && ua && va && ut && vt && divgd && omga);
  if (part < 0 || part > 2) return PACE_ERR_ARG;
  return launch_c_sw(make_geo(geom), *met, workspace, delpc, ptc, delp, pt, u, v, w, uc, vc, ua, va, ut, vt, divgd, omga, dt2,
                     nord, S(stream), part);
}

int pace_d2a2c_vect(const pace_geom_t* geom, const pace_metrics_t* met, void* workspace, real* uc, real* vc,
                    const real* u, const real* v, real* ua, real* va, real* utc, real* vtc, void* stream) {
  NEED(geom && met && workspace && uc && vc && u && v && ua && va && utc && vtc);
  return launch_d2a2c_vect(make_geo(geom), *met, workspace, uc, vc, u, v, ua, va, utc, vtc, S(stream));
}

int64_t pace_riem_solver_c_workspace_bytes(const pace_geom_t* geom) {
  return geom ? riemc_workspace_bytes(make_geo(geom)) : 0;
}

int pace_riem_solver_c(const pace_geom_t* geom, void* workspace, double dt2, const real* cappa, double ptop,
                       const real* hs, const real* ws, const real* ptc, const real* q_con,
                       const real* delpc, real* gz, real* pef, const real* w3, double p_fac, void* stream) {
  NEED(geom && workspace && cappa && hs && ws && ptc && q_con && delpc && gz && pef && w3);
  const Geo g = make_geo(geom);
  if (riem_column_supported(g) && !legacy_column_solvers())
    return launch_riem_solver_c_column(g, dt2, cappa, ptop, hs, ws, ptc, q_con, delpc, gz, pef, w3, p_fac, S(stream));
  return launch_riem_solver_c(g, workspace, dt2, cappa, ptop, hs, ws, ptc, q_con, delpc, gz, pef, w3, p_fac, S(stream));
}

int64_t pace_sim1_solver_workspace_bytes(const pace_geom_t* geom) { return geom ? sim1_workspace_bytes(make_geo(geom)) : 0; }

int pace_sim1_solver(const pace_geom_t* geom, void* workspace, int n_halo, double dt, double p_fac, const real* gamma,
                     const real* cp3, real* pe, const real* delta_mass, const real* pm, const real* pem, real* w,
                     real* dz, const real* potential_temperature, const real* ws, void* stream) {
  NEED(geom && workspace && gamma && cp3 && pe && delta_mass && pm && pem && w && dz && potential_temperature && ws);
  if (n_halo < 0 || n_halo > 3 || geom->nk < 2) return PACE_ERR_ARG;
  return launch_sim1_solver(make_geo(geom), workspace, n_halo, dt, p_fac, gamma, cp3, pe, delta_mass, pm, pem, w, dz,
                            potential_temperature, ws, S(stream));
}

int64_t pace_updatedzc_workspace_bytes(const pace_geom_t* geom) {
  return geom ? updatedzc_workspace_bytes(make_geo(geom)) : 0;
}

int pace_updatedzc(const pace_geom_t* geom, const pace_metrics_t* met, void* workspace, const real* dp_ref,
                   const real* zs, const real* ut, const real* vt, real* gz, real* ws, double dt,
                   void* stream) {
  NEED(geom && met && workspace && dp_ref && zs && ut && vt && gz && ws);
  return launch_updatedzc(make_geo(geom), *met, workspace, dp_ref, zs, ut, vt, gz, ws, dt, S(stream));
}

int64_t pace_updatedzd_workspace_bytes(const pace_geom_t* geom) {
  return geom ? updatedzd_workspace_bytes(make_geo(geom)) : 0;
}

int pace_updatedzd(const pace_geom_t* geom, const pace_metrics_t* met, void* workspace, const pace_updatedzd_k_t* kc,
                   const real* surface_height, real* height, const real* courant_number_x,
                   const real* courant_number_y, const real* x_area_flux, const real* y_area_flux, real* ws,
                   double dt, int hord_tm, void* stream) {
  NEED(geom && met && workspace && kc && surface_height && height && courant_number_x && courant_number_y && x_area_flux &&
       y_area_flux && ws);
  NEED(kc->gk && kc->beta && kc->gamma && kc->damp && kc->nord);
  return launch_updatedzd(make_geo(geom), *met, workspace, kc, surface_height, height, courant_number_x, courant_number_y,
                          x_area_flux, y_area_flux, ws, dt, hord_tm, S(stream));
}

int pace_zero_data(const pace_geom_t* geom, real* mfxd, real* mfyd, real* cxd, real* cyd, real* heat_source,
                   real* diss_estd, int first_timestep, void* stream) {
  NEED(geom && mfxd && mfyd && cxd && cyd && heat_source && diss_estd);
  return launch_zero_data(make_geo(geom), mfxd, mfyd, cxd, cyd, heat_source, diss_estd, first_timestep, S(stream));
}

int pace_interface_pressure_from_toa_pressure_and_thickness(const pace_geom_t* geom, const real* delp, real* pem,
                                                            double ptop, void* stream) {
  NEED(geom && delp && pem);
  return launch_interface_pressure(make_geo(geom), delp, pem, ptop, S(stream));
}

int pace_gz_from_surface_height_and_thicknesses(const pace_geom_t* geom, const real* zs, const real* delz,
                                                real* gz, void* stream) {
  NEED(geom && zs && delz && gz);
  return launch_gz_from_surface(make_geo(geom), zs, delz, gz, S(stream));
}

int pace_compute_geopotential(const pace_geom_t* geom, const real* zh, real* gz, void* stream) {
  NEED(geom && zh && gz);
  return launch_scale_copy(make_geo(geom), zh, gz, 9.80665, 1, 2, geom->nk + 1, S(stream));
}

int pace_copy(const pace_geom_t* geom, const real* src, real* dst, void* stream) {
  NEED(geom && src && dst);
  return launch_scale_copy(make_geo(geom), src, dst, 1.0, 0, 3, geom->nk + 1, S(stream));
}

int pace_p_grad_c(const pace_geom_t* geom, const pace_metrics_t* met, real* uc, real* vc, const real* delpc,
                  const real* pkc, const real* gz, double dt2, void* stream) {
  NEED(geom && met && uc && vc && delpc && pkc && gz);
  return launch_p_grad_c(make_geo(geom), *met, uc, vc, delpc, pkc, gz, dt2, S(stream));
}

int64_t pace_nh_p_grad_workspace_bytes(const pace_geom_t* geom) {
  return geom ? nh_p_grad_workspace_bytes(make_geo(geom)) : 0;
}

int pace_nh_p_grad(const pace_geom_t* geom, const pace_metrics_t* met, void* workspace, real* u, real* v,
                   real* pp, real* gz, real* pk3, real* delp, double dt, double ptop, double akap,
                   void* stream) {
  NEED(geom && met && workspace && u && v && pp && gz && pk3 && delp);
  return launch_nh_p_grad(make_geo(geom), *met, workspace, u, v, pp, gz, pk3, delp, dt, ptop, akap, S(stream));
}

int pace_edge_pe(const pace_geom_t* geom, real* pe, const real* delp, double ptop, void* stream) {
  NEED(geom && pe && delp);
  return launch_edge_pe(make_geo(geom), pe, delp, ptop, S(stream));
}

int pace_pk3_halo(const pace_geom_t* geom, real* pk3, const real* delp, double ptop, double akap, void* stream) {
  NEED(geom && pk3 && delp);
  return launch_pk3_halo(make_geo(geom), pk3, delp, ptop, akap, S(stream));
}

int pace_ray_fast(const pace_geom_t* geom, real* u, real* v, real* w, const double* dp, const double* pfull,
                  double dt, double ptop, double rf_cutoff, double tau, int hydrostatic, void* stream) {
  NEED(geom && u && v && w && dp && pfull);
  return launch_ray_fast(make_geo(geom), u, v, w, dp, pfull, dt, ptop, rf_cutoff, tau, hydrostatic, S(stream));
}

int64_t pace_del2cubed_workspace_bytes(const pace_geom_t* geom) {
  return geom ? del2cubed_workspace_bytes(make_geo(geom)) : 0;
}

int pace_del2cubed(const pace_geom_t* geom, const pace_metrics_t* met, void* workspace, real* qdel, double cd,
                   int nmax, void* stream) {
  NEED(geom && met && workspace && qdel);
  return launch_del2cubed(make_geo(geom), *met, workspace, qdel, cd, nmax, S(stream));
}

int pace_apply_diffusive_heating(const pace_geom_t* geom, const real* delp, const real* delz, const real* cappa,
                                 const real* heat_source, real* pt, double delt_time_factor, int nlev,
                                 void* stream) {
  NEED(geom && delp && delz && cappa && heat_source && pt);
  if (nlev < 0 || nlev > geom->nk) return PACE_ERR_ARG;
  return launch_diffusive_heating(make_geo(geom), delp, delz, cappa, heat_source, pt, delt_time_factor, nlev, S(stream));
}

int pace_tracer_flux_compute(const pace_geom_t* geom, const pace_metrics_t* met, const real* cx, const real* cy,
                             real* xfx, real* yfx, void* stream) {
  NEED(geom && met && cx && cy && xfx && yfx);
  return launch_tracer_flux_compute(make_geo(geom), *met, cx, cy, xfx, yfx, S(stream));
}

int pace_tracer_divide_fluxes(const pace_geom_t* geom, real* cxd, real* xfx, real* mfxd, real* cyd, real* yfx,
                              real* mfyd, int n_split, void* stream) {
  NEED(geom && cxd && xfx && mfxd && cyd && yfx && mfyd);
  if (n_split < 1) return PACE_ERR_ARG;
  return launch_tracer_divide(make_geo(geom), cxd, xfx, mfxd, cyd, yfx, mfyd, n_split, S(stream));
}

int pace_apply_mass_flux(const pace_geom_t* geom, const pace_metrics_t* met, const real* dp1, const real* x_mass_flux,
                         const real* y_mass_flux, real* dp2, void* stream) {
  NEED(geom && met && dp1 && x_mass_flux && y_mass_flux && dp2);
  return launch_apply_mass_flux(make_geo(geom), *met, dp1, x_mass_flux, y_mass_flux, dp2, S(stream));
}

int pace_apply_tracer_flux(const pace_geom_t* geom, const pace_metrics_t* met, real* q, const real* dp1,
                           const real* fx, const real* fy, const real* dp2, void* stream) {
  NEED(geom && met && q && dp1 && fx && fy && dp2);
  return launch_apply_tracer_flux(make_geo(geom), *met, q, dp1, fx, fy, dp2, S(stream));
}

int pace_swap_dp(const pace_geom_t* geom, real* dp1, real* dp2, void* stream) {
  NEED(geom && dp1 && dp2);
  return launch_swap_dp(make_geo(geom), dp1, dp2, S(stream));
}

static int halo_check(const pace_geom_t* geom, const pace_halo_desc_t* d, int n) {
  const int ni = geom->n + 7;
  for (int t = 0; t < n; ++t) {
    if (!d[t].field || !d[t].buf || d[t].na < 1 || d[t].nb < 1 || d[t].nk < 1 || d[t].nk > geom->nk + 1) return PACE_ERR_ARG;
    for (int ca = 0; ca < 2; ++ca)
      for (int cb = 0; cb < 2; ++cb) {
        const int a = ca ? d[t].na - 1 : 0, b = cb ? d[t].nb - 1 : 0;
        const int i = d[t].i0 + a * d[t].di_a + b * d[t].di_b, j = d[t].j0 + a * d[t].dj_a + b * d[t].dj_b;
        if (i < 0 || i >= ni || j < 0 || j >= ni) return PACE_ERR_ARG;
      }
  }
  return PACE_OK;
}

int64_t pace_map_single_workspace_bytes(const pace_geom_t* geom) {
  return geom ? map_single_workspace_bytes(make_geo(geom), 1) : 0;
}

int pace_map_single(const pace_geom_t* geom, void* workspace, real* q1, const real* pe1, const real* pe2,
                    const real* qs, double qmin, int kord, int iv, int xstag, int ystag, void* stream) {
  NEED(geom && workspace && q1 && pe1 && pe2);
  return launch_map_fields(make_geo(geom), workspace, &q1, 1, pe1, pe2, qs, qmin, kord, iv, xstag, ystag, S(stream));
}

int64_t pace_mapn_tracer_workspace_bytes(const pace_geom_t* geom, int nq) {
  return (geom && nq > 0) ? map_single_workspace_bytes(make_geo(geom), nq) : 0;
}

int pace_mapn_tracer(const pace_geom_t* geom, void* workspace, real* const* tracers, int nq, const real* pe1,
                     const real* pe2, int kord, void* stream) {
  NEED(geom && workspace && tracers && pe1 && pe2);
  return launch_map_fields(make_geo(geom), workspace, tracers, nq, pe1, pe2, nullptr, 0.0, kord, 0, 0, 0, S(stream));
}

int pace_fillz(const pace_geom_t* geom, real* const* tracers, int nq, const real* dp2, void* stream) {
  NEED(geom && tracers && dp2);
  return launch_fillz(make_geo(geom), tracers, nq, dp2, S(stream));
}

static bool six(const real* const* w) {
  if (!w) return false;
  for (int n = 0; n < 6; ++n)
    if (!w[n]) return false;
  return true;
}

int pace_l2e_prepare(const pace_geom_t* geom, const real* const* water, real* q_con, real* pt, real* cappa, real* delp,
                     real* delz, const real* pe, real* pe1, real* pe2, const real* ak, const real* bk, real* dp2,
                     real* ps, real* pn2, const real* peln, real* pk, double ptop, double akap, double r_vir,
                     void* stream) {
  NEED(geom && six(water) && q_con && pt && cappa && delp && delz && pe && pe1 && pe2 && ak && bk && dp2 && ps && pn2 && peln && pk);
  return launch_l2e_prepare(make_geo(geom), water, q_con, pt, cappa, delp, delz, pe, pe1, pe2, ak, bk, dp2, ps, pn2, peln, pk,
                            ptop, akap, r_vir, S(stream));
}

int pace_l2e_post(const pace_geom_t* geom, const real* const* water, real* q_con, real* pkz, const real* pt,
                  real* cappa, const real* delp, real* delz, real* peln, real* pe0, const real* pn2, double r_vir,
                  void* stream) {
  NEED(geom && six(water) && q_con && pkz && pt && cappa && delp && delz && peln && pe0 && pn2);
  return launch_l2e_post(make_geo(geom), water, q_con, pkz, pt, cappa, delp, delz, peln, pe0, pn2, r_vir, S(stream));
}

int pace_l2e_pressures(const pace_geom_t* geom, int dir, const real* pe, const real* pe1, const real* ak,
                       const real* bk, real* pe0, real* pe3, void* stream) {
  NEED(geom && pe && pe1 && ak && bk && pe0 && pe3);
  if (dir != 0 && dir != 1) return PACE_ERR_ARG;
  return launch_l2e_pressures(make_geo(geom), dir, pe, pe1, ak, bk, pe0, pe3, S(stream));
}

int pace_l2e_finish(const pace_geom_t* geom, const real* const* water, real* pe, const real* pe2, real* pt,
                    const real* pkz, double r_vir, int last_step, void* stream) {
  NEED(geom && six(water) && pe && pe2 && pt && pkz);
  return launch_l2e_finish(make_geo(geom), water, pe, pe2, pt, pkz, r_vir, last_step, S(stream));
}

static bool six_rw(real* const* w) {
  if (!w) return false;
  for (int n = 0; n < 6; ++n)
    if (!w[n]) return false;
  return true;
}

int pace_fv_setup_pt(const pace_geom_t* geom, real* const* water, real* q_con, real* pkz, real* pt, real* cappa,
                     const real* delp, const real* delz, real* dp1, void* stream) {
  NEED(geom && six_rw(water) && q_con && pkz && pt && cappa && delp && delz && dp1);
  return launch_fv_setup_pt(make_geo(geom), water, q_con, pkz, pt, cappa, delp, delz, dp1, S(stream));
}

int pace_omega_from_w(const pace_geom_t* geom, const real* delp, const real* delz, const real* w, real* omga,
                      void* stream) {
  NEED(geom && delp && delz && w && omga);
  return launch_omega_from_w(make_geo(geom), delp, delz, w, omga, S(stream));
}

int pace_neg_adj3(const pace_geom_t* geom, real* const* water, real* qcld, real* pt, const real* delp, void* stream) {
  NEED(geom && six_rw(water) && qcld && pt && delp);
  return launch_neg_adj3(make_geo(geom), water, qcld, pt, delp, S(stream));
}

int pace_c2l_ord(const pace_geom_t* geom, const pace_metrics_t* met, int order, const real* u, const real* v,
                 const real* a11, const real* a12, const real* a21, const real* a22, real* ua, real* va,
                 void* stream) {
  NEED(geom && met && u && v && a11 && a12 && a21 && a22 && ua && va);
  if (order != 2 && order != 4) return PACE_ERR_ARG;
  return launch_c2l(make_geo(geom), *met, order, u, v, a11, a12, a21, a22, ua, va, S(stream));
}


int pace_halo_pack(const pace_geom_t* geom, const pace_halo_desc_t* descs, int ndesc, void* stream) {
  NEED(geom && descs && ndesc > 0);
  if (halo_check(geom, descs, ndesc)) return PACE_ERR_ARG;
  return launch_halo_copy(make_geo(geom), descs, ndesc, 0, S(stream));
}

int pace_halo_unpack(const pace_geom_t* geom, const pace_halo_desc_t* descs, int ndesc, void* stream) {
  NEED(geom && descs && ndesc > 0);
  if (halo_check(geom, descs, ndesc)) return PACE_ERR_ARG;
  return launch_halo_copy(make_geo(geom), descs, ndesc, 1, S(stream));
}

int pace_stencil(const pace_geom_t* geom, const pace_metrics_t* met, int id, void* const* fields, int nfields, const double* scalars,
                 int nscalars, const int* origin, const int* domain, void* stream) {
  NEED(geom && met && fields && origin && domain && nfields >= 1 && nfields <= 16 && nscalars >= 0 && (nscalars == 0 || scalars));
  for (int n = 0; n < nfields; ++n)
    if (!fields[n]) return PACE_ERR_ARG;
  return launch_stencil(make_geo(geom), *met, id, fields, nfields, scalars, nscalars, origin, domain, S(stream));
}

}  // extern "C"
