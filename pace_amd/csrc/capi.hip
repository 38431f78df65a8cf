// extern "C" entry points of libpace_hip.so (declared in include/pace_hip.h).  Argument checking and
// struct unpacking only; kernels live in k_*.hip.
#include "common.h"
#include "kernels.h"

#include <cstdio>

thread_local char g_pace_err[256] = "";
void pace_set_err(const char* where, hipError_t e) {
  snprintf(g_pace_err, sizeof(g_pace_err), "%s: %s", where, hipGetErrorString(e));
}

// Entry prologue: argument check, then drop any stale "last error" another library (e.g. the caching
// allocator's event queries) left on this thread, so PACE_CHECK_LAUNCH only reports our own launches.
#define NEED(p)                      \
  if (!(p)) return PACE_ERR_ARG;     \
  (void)hipGetLastError()

static inline hipStream_t S(void* s) { return (hipStream_t)s; }

extern "C" {

const char* pace_last_error(void) { return g_pace_err; }

const char* pace_version(void) {
#ifdef PACE_EMU
  return "pace_amd 0.1 (CPU emulation build -- test infrastructure only)";
#else
  return "pace_amd 0.1 (gfx950)";
#endif
}

int pace_fxadv(const pace_geom_t* geom, const pace_metrics_t* met, const double* uc, const double* vc, double* crx,
               double* cry, double* xfx, double* yfx, double* ut, double* vt, double dt, void* stream) {
  NEED(geom && met && uc && vc && crx && cry && xfx && yfx && ut && vt);
  return launch_fxadv(make_geo(geom), *met, uc, vc, crx, cry, xfx, yfx, ut, vt, dt, S(stream));
}

int pace_fvtp2d(const pace_geom_t* geom, const pace_metrics_t* met, const double* q, const double* crx,
                const double* cry, const double* xfx, const double* yfx, double* fx, double* fy, const double* xmf,
                const double* ymf, int hord, int nlev, void* stream) {
  NEED(geom && met && q && crx && cry && xfx && yfx && fx && fy);
  if ((xmf == nullptr) != (ymf == nullptr)) return PACE_ERR_ARG;
  if (nlev < 1 || nlev > geom->nk + 1) return PACE_ERR_ARG;
  return launch_fvtp2d(make_geo(geom), *met, q, crx, cry, xfx, yfx, fx, fy, xmf, ymf, hord, nlev, S(stream));
}

int pace_delnflux_nosg(const pace_geom_t* geom, const pace_metrics_t* met, const double* q, double* fx2, double* fy2,
                       const double* damp_k, const double* nord_k, int nmax, int mass_given, int nlev, void* stream) {
  NEED(geom && met && q && fx2 && fy2 && damp_k && nord_k);
  if (nlev < 1 || nlev > geom->nk + 1) return PACE_ERR_ARG;
  return launch_delnflux(make_geo(geom), *met, 0, q, fx2, fy2, nullptr, damp_k, nord_k, nmax, mass_given, nlev, S(stream));
}

int pace_delnflux(const pace_geom_t* geom, const pace_metrics_t* met, const double* q, double* fx, double* fy,
                  const double* mass, const double* damp_k, const double* nord_k, int nmax, int nlev, void* stream) {
  NEED(geom && met && q && fx && fy && damp_k && nord_k);
  if (nlev < 1 || nlev > geom->nk + 1) return PACE_ERR_ARG;
  return launch_delnflux(make_geo(geom), *met, mass ? 2 : 1, q, fx, fy, mass, damp_k, nord_k, nmax, mass ? 1 : 0, nlev,
                         S(stream));
}

int pace_a2b_ord4(const pace_geom_t* geom, const pace_metrics_t* met, double* qin, double* qout, int k0, int k1,
                  int replace, void* stream) {
  NEED(geom && met && qin && qout);
  if (k0 < 0 || k1 <= k0 || k1 > geom->nk + 1) return PACE_ERR_ARG;
  return launch_a2b_ord4(make_geo(geom), *met, qin, qout, k0, k1, replace, S(stream));
}

int64_t pace_d_sw_workspace_bytes(const pace_geom_t* geom) { return geom ? dsw_workspace_bytes(make_geo(geom)) : 0; }

int pace_d_sw_prepare(const pace_geom_t* geom, const pace_column_t* col, void* workspace, void* stream) {
  NEED(geom && col && workspace);
  return dsw_prepare(make_geo(geom), col, workspace, S(stream));
}

int pace_d_sw(const pace_geom_t* geom, const pace_metrics_t* met, const pace_column_t* col,
              const pace_dsw_config_t* cfg, void* workspace, double* delpc, double* delp, double* pt, double* u,
              double* v, double* w, double* uc, double* vc, const double* ua, const double* va, double* divgd,
              double* mfx, double* mfy, double* cx, double* cy, double* crx, double* cry, double* xfx, double* yfx,
              double* q_con, const double* zh, double* heat_source, double* diss_est, double dt, void* stream) {
  NEED(geom && met && col && cfg && workspace);
  NEED(delpc && delp && pt && u && v && w && uc && vc && ua && va && divgd && mfx && mfy && cx && cy);
  NEED(crx && cry && xfx && yfx && q_con && heat_source && diss_est);
  return launch_d_sw(make_geo(geom), *met, col, cfg, workspace, delpc, delp, pt, u, v, w, uc, vc, ua, va, divgd, mfx, mfy,
                     cx, cy, crx, cry, xfx, yfx, q_con, zh, heat_source, diss_est, dt, S(stream));
}

int64_t pace_riem_solver3_workspace_bytes(const pace_geom_t* geom) {
  return geom ? riem3_workspace_bytes(make_geo(geom)) : 0;
}

int pace_riem_solver3(const pace_geom_t* geom, void* workspace, int last_call, double dt, const double* cappa,
                      double ptop, const double* zs, const double* ws, double* delz, const double* q_con,
                      const double* delp, const double* pt, double* zh, double* pe, double* ppe, double* pk3, double* pk,
                      double* peln, double* w, double p_fac, void* stream) {
  NEED(geom && workspace && cappa && zs && ws && delz && q_con && delp && pt && zh && pe && ppe && pk3 && pk && peln && w);
  return launch_riem_solver3(make_geo(geom), workspace, last_call, dt, cappa, ptop, zs, ws, delz, q_con, delp, pt, zh, pe,
                             ppe, pk3, pk, peln, w, p_fac, S(stream));
}

}  // extern "C"
