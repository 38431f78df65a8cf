// Host-side launch functions (one per reference class on the hot path).
#pragma once
#include "common.h"

int launch_fxadv(const Geo& g, const Met& m, const double* uc, const double* vc, double* crx, double* cry,
                 double* xfx, double* yfx, double* ut, double* vt, double dt, hipStream_t st);
int launch_fvtp2d(const Geo& g, const Met& m, const double* q, const double* crx, const double* cry,
                  const double* xfx, const double* yfx, double* fx, double* fy, const double* xmf,
                  const double* ymf, int hord, int nlev, hipStream_t st);
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
                double* heat_source, double* diss_est, double dt, hipStream_t st);
int64_t riem3_workspace_bytes(const Geo& g);
int launch_riem_solver3(const Geo& g, void* ws, int last_call, double dt, const double* cappa, double ptop,
                        const double* zs, const double* wsd, double* delz, const double* q_con, const double* delp,
                        const double* pt, double* zh, double* pe, double* ppe, double* pk3, double* pk, double* peln,
                        double* w, double p_fac, hipStream_t st);
