// What DynamicalCore (fv3core/pace/fv3core/stencils/fv_dynamics.py:424-624) runs around the acoustic loop, the tracer
// advection and the remapping:
//   k_fv_setup_pt     moist_cv.fv_setup (moist_cv.py:175-234) + pt_to_potential_density_pt (fv_dynamics.py:41-54), one pass
//   k_omega_from_w    fv_dynamics.py:57-67
//   k_fix_neg_water   neg_adj3.fix_neg_water (neg_adj3.py:11-140), per cell
//   k_neg_columns     fillq(qgraupel), fillq(qrain), fix_water_vapor_down(qvapor), fix_neg_cloud(qcld) (neg_adj3.py:143-281):
//                     four independent column operators in ONE launch (blockIdx.z picks the operator), each restated as
//                     streaming sweeps with the few values in flight in registers (the reference keeps whole-column
//                     upper_fix / lower_fix / dp_bottom fields)
//   k_c2l             CubedToLatLon, both orders (stencils/pace/stencils/c2l_ord.py:15-112)
#include "common.h"
#include "kernels.h"

#define PACE_GRAV 9.80665
#define PACE_RDGAS 287.05
#define PACE_RVGAS 461.50
#define PACE_CP_AIR 1004.6
#define PACE_CV_AIR (PACE_CP_AIR - PACE_RDGAS)
#define PACE_RDG (-PACE_RDGAS / PACE_GRAV)
#define PACE_ZVIR (PACE_RVGAS / PACE_RDGAS - 1)
#define PACE_CV_VAP (3.0 * PACE_RVGAS)
#define PACE_C_ICE 1972.0
#define PACE_C_LIQ 4.1855e3
#define PACE_HLV 2.5e6
#define PACE_HLF 3.3358e5
#define PACE_TICE 273.16
#define PACE_DC_ICE (PACE_C_LIQ - PACE_C_ICE)
#define PACE_LI0 (PACE_HLF - PACE_DC_ICE * PACE_TICE)

struct Water6 {
  real *qvapor, *qliquid, *qrain, *qsnow, *qice, *qgraupel;
};

#define CELL_IJK()                                     \
  const int i = g.is + blockIdx.x * 64 + threadIdx.x;  \
  const int j = g.js + blockIdx.y * 4 + threadIdx.y;   \
  const int k = blockIdx.z;                            \
  if (i > g.ie || j > g.je) return;                    \
  const long c = IDX3(g, i, j, k)

__global__ void __launch_bounds__(256)
k_fv_setup_pt(Geo g, Water6 q, real* __restrict__ q_con, real* __restrict__ pkz, real* __restrict__ pt,
              real* __restrict__ cappa, const real* __restrict__ delp, const real* __restrict__ delz,
              real* __restrict__ dp1) {
  CELL_IJK();
  const double qv = q.qvapor[c];
  const double ql = q.qliquid[c] + q.qrain[c];
  const double qs = q.qice[c] + q.qsnow[c] + q.qgraupel[c];
  const double gz = ql + qs;
  const double cvm = (1.0 - (qv + gz)) * PACE_CV_AIR + qv * PACE_CV_VAP + ql * PACE_C_LIQ + qs * PACE_C_ICE;
  const double d1 = PACE_ZVIR * qv;
  const double cp = PACE_RDGAS / (PACE_RDGAS + cvm / (1.0 + d1));
  const double p = pt[c];
  const double pz = exp(cp * log(PACE_RDG * delp[c] * p * (1.0 + d1) * (1.0 - gz) / delz[c]));
  q_con[c] = gz;
  dp1[c] = d1;
  cappa[c] = cp;
  pkz[c] = pz;
  pt[c] = p * (1.0 + d1) * (1.0 - gz) / pz;
}

__global__ void __launch_bounds__(256)
k_omega_from_w(Geo g, const real* __restrict__ delp, const real* __restrict__ delz, const real* __restrict__ w,
               real* __restrict__ omga) {
  CELL_IJK();
  omga[c] = delp[c] / delz[c] * w[c];
}

__global__ void __launch_bounds__(256)
k_fix_neg_water(Geo g, Water6 q, real* __restrict__ ptf, double lv00, double d0_vap) {
  CELL_IJK();
  double qv = q.qvapor[c], ql = q.qliquid[c], qr = q.qrain[c], qs = q.qsnow[c], qi = q.qice[c], qg = q.qgraupel[c];
  double pt = ptf[c];
  const double q_liq = 0.0 > ql + qr ? 0.0 : ql + qr;
  const double q_sol = 0.0 > qi + qs ? 0.0 : qi + qs;
  const double cpm = (1.0 - (qv + q_liq + q_sol)) * PACE_CV_AIR + qv * PACE_CV_VAP + q_liq * PACE_C_LIQ + q_sol * PACE_C_ICE;
  const double lcpk = (lv00 + d0_vap * pt) / cpm;
  const double icpk = (PACE_LI0 + PACE_DC_ICE * pt) / cpm;
  double dq;
  // fix_negative_ice (neg_adj3.py:13-54)
  {
    const double qsum = qi + qs;
    if (qsum > 0.0) {
      if (qi < 0.0) {
        qi = 0.0;
        qs = qsum;
      } else if (qs < 0.0) {
        qs = 0.0;
        qi = qsum;
      }
    } else {
      qi = 0.0;
      qs = 0.0;
      qg = qg + qsum;
    }
    if (qg < 0.0) {
      dq = qs < -qg ? qs : -qg;
      qs = qs - dq;
      qg = qg + dq;
      if (qg < 0.0) {
        dq = qi < -qg ? qi : -qg;
        qi = qi - dq;
        qg = qg + dq;
      }
    }
    if (qg < 0.0 && qr > 0.0) {
      dq = qr < -qg ? qr : -qg;
      qg = qg + dq;
      ql = ql - dq;
      pt = pt + dq * icpk;
    }
    if (qg < 0.0 && ql > 0.0) {
      dq = ql < -qg ? ql : -qg;
      qg = qg + dq;
      ql = ql - dq;
      pt = pt + dq * icpk;
    }
    if (qg < 0.0 && qv > 0.0) {
      dq = 0.999 * qv < -qg ? 0.999 * qv : -qg;
      qg = qg + dq;
      qv = qv - dq;
      pt = pt + dq * (icpk + lcpk);
    }
  }
  // fix_negative_liq (neg_adj3.py:57-98)
  {
    const double qsum = ql + qr;
    const double pos_qg = 0.0 > qg ? 0.0 : qg;
    if (qsum > 0.0) {
      if (qr < 0.0) {
        qr = 0.0;
        ql = qsum;
      } else if (ql < 0.0) {
        ql = 0.0;
        qr = qsum;
      }
    } else {
      ql = 0.0;
      double qr_tmp = qsum;
      dq = pos_qg < -qr_tmp ? pos_qg : -qr_tmp;
      qr_tmp = qr_tmp + dq;
      qg = qg - dq;
      pt = pt - dq * icpk;
      if (qr < 0.0) {
        dq = (qi + qs) < -qr_tmp ? qi + qs : -qr_tmp;
        qr_tmp = qr_tmp + dq;
        const double dq1 = dq < qs ? dq : qs;
        qs = qs - dq1;
        qi = qi + dq1 - dq;
        pt = pt - dq * icpk;
      }
      qr = qr_tmp;
      if (qr < 0.0 && qv > 0.0) {
        dq = 0.999 * qv < -qr ? 0.999 * qv : -qr;
        qv = qv - dq;
        qr = qr + dq;
        pt = pt + dq * lcpk;
      }
    }
  }
  q.qvapor[c] = qv; q.qliquid[c] = ql; q.qrain[c] = qr; q.qsnow[c] = qs; q.qice[c] = qi; q.qgraupel[c] = qg;
  ptf[c] = pt;
}

// ---- column operators of neg_adj3 -------------------------------------------------------------------------------
#define Q(k) q[c0 + (long)(k) * sk]
#define DP(k) dp[c0 + (long)(k) * sk]

__device__ __forceinline__ void col_fillq(real* __restrict__ q, const real* __restrict__ dp, long c0, long sk, int km) {
  double s1 = 0.0, s2 = 0.0;
  for (int k = 0; k < km; ++k) {
    const double v = Q(k);
    if (v > 0) s1 = s1 + v * DP(k);
  }
  for (int k = km - 1; k >= 0; --k) {
    const double v = Q(k);
    if (v < 0.0 && s1 >= 0) {
      const double d = DP(k);
      const double dq = s1 < -v * d ? s1 : -v * d;
      s1 = s1 - dq;
      s2 = s2 + dq;
      Q(k) = v + dq / d;
    }
  }
  for (int k = km - 1; k >= 0; --k) {
    const double v = Q(k);
    if (v > 0.0 && s1 >= 1e-12 && s2 > 0) {
      const double d = DP(k);
      const double dq = s2 < v * d ? s2 : v * d;
      s2 = s2 - dq;
      Q(k) = v - dq / d;
    }
  }
}

__device__ __forceinline__ void col_vapor_down(real* __restrict__ q, const real* __restrict__ dp, long c0, long sk, int km) {
  double dpm = DP(0), dpk = DP(1);
  double qm = Q(0), qk = Q(1);
  if (qm < 0) qk = qk + qm * dpm / dpk;
  if (qm < 0.0) qm = 0.0;
  double lower_prev = 0.0;
  for (int k = 1; k < km - 1; ++k) {
    double dq = qm * dpm;
    double upper = 0.0, lower = 0.0;
    if (lower_prev != 0) qk += lower_prev / dpk;
    if (qk < 0 && qm > 0) {
      dq = dq < -qk * dpk ? dq : -qk * dpk;
      upper = dq;
      qk += dq / dpk;
    }
    if (qk < 0) {
      lower = qk * dpk;
      qk = 0;
    }
    if (upper != 0) qm = qm - upper / dpm;  // (level k-1 <= km-3: neg_adj3.py:206-209)
    Q(k - 1) = qm;
    qm = qk; dpm = dpk;
    lower_prev = lower;
    qk = Q(k + 1);
    dpk = DP(k + 1);
  }
  Q(km - 2) = qm;
  // bottom level (neg_adj3.py:210-219): qk / dpk are its value and thickness
  if (lower_prev > 0) qk = qk + 0.0 / dpk;
  double un = qk;
  const double dpb = dpk;
  for (int k = km - 2; k >= 0; --k) {
    const double v = Q(k), d = DP(k);
    double dq = v * d;
    if (un < 0 && v > 0) {
      if (dq >= -un * dpb) dq = -un * dpb;
      Q(k) = v - dq / d;
      un = un + dq / dpb;
    }
  }
  Q(km - 1) = un;
}

__device__ __forceinline__ void col_neg_cloud(real* __restrict__ q, const real* __restrict__ dp, long c0, long sk, int km) {
  double qm = Q(0), dpm = DP(0);  // level k-1 as the forward sweep sees it (before the clamp)
  for (int k = 1; k < km - 1; ++k) {
    double v = Q(k);
    const double d = DP(k);
    if (qm < 0.0) v = v + qm * dpm / d;
    if (k - 1 >= 1) Q(k - 1) = qm < 0.0 ? 0.0 : qm;
    qm = v;
    dpm = d;
  }
  // qm = level km-2 before the clamp
  double q2 = qm < 0.0 ? 0.0 : qm;
  if (km - 2 < 1) q2 = qm;
  const double d2 = dpm;
  double qb = Q(km - 1);
  const double db = DP(km - 1);
  if (qb < 0.0 && q2 > 0) {
    const double dq = -q2 * d2 < qb * db ? -q2 * d2 : qb * db;
    q2 = q2 - dq / d2;
  }
  Q(km - 2) = q2;
  if (qb < 0 && q2 > 0.0) {
    const double dq = -qb * db < q2 * d2 ? -qb * db : q2 * d2;
    qb = qb + dq / db;
    qb = 0.0 > qb ? 0.0 : qb;
  }
  Q(km - 1) = qb;
}
#undef Q
#undef DP

__global__ void __launch_bounds__(64)
k_neg_columns(Geo g, real* __restrict__ qgraupel, real* __restrict__ qrain, real* __restrict__ qvapor,
              real* __restrict__ qcld, const real* __restrict__ dp) {
  const int i = g.is + blockIdx.x * 64 + threadIdx.x;
  const int j = g.js + blockIdx.y;
  if (i > g.ie) return;
  const long c0 = IDX2(g, i, j);
  switch (blockIdx.z) {
    case 0: col_fillq(qgraupel, dp, c0, g.sk, g.nk); break;
    case 1: col_fillq(qrain, dp, c0, g.sk, g.nk); break;
    case 2: col_vapor_down(qvapor, dp, c0, g.sk, g.nk); break;
    default: col_neg_cloud(qcld, dp, c0, g.sk, g.nk); break;
  }
}

// ---- CubedToLatLon ----------------------------------------------------------------------------------------------
template <int ORD>
__global__ void __launch_bounds__(256)
k_c2l(Geo g, Met m, const real* __restrict__ u, const real* __restrict__ v, const real* __restrict__ a11,
      const real* __restrict__ a12, const real* __restrict__ a21, const real* __restrict__ a22, real* __restrict__ ua,
      real* __restrict__ va) {
  const int h = ORD == 2 ? 1 : 0;  // compute_halos (c2l_ord.py:147-152)
  const int i = g.is - h + blockIdx.x * 64 + threadIdx.x;
  const int j = g.js - h + blockIdx.y * 4 + threadIdx.y;
  const int k = blockIdx.z;
  if (i > g.ie + h || j > g.je + h) return;
  const long c2 = IDX2(g, i, j);
  const long c = c2 + (long)k * g.sk;
  const long sj = g.sj;
  const double dx0 = m.dx[c2], dx1 = m.dx[c2 + sj], dy0 = m.dy[c2], dy1 = m.dy[c2 + 1];
  double ut, vt;
  const bool edge = ORD == 2 || i == g.is || i == g.ie || j == g.js || j == g.je;
  if (edge) {
    if (ORD == 2) {
      const double wu0 = u[c] * dx0, wu1 = u[c + sj] * dx1, wv0 = v[c] * dy0, wv1 = v[c + 1] * dy1;
      ut = 2.0 * (wu0 + wu1) / (dx0 + dx1);
      vt = 2.0 * (wv0 + wv1) / (dy0 + dy1);
    } else {
      vt = 2.0 * ((v[c] * dy0) + (v[c + 1] * dy1)) / (dy0 + dy1);
      ut = 2.0 * (u[c] * dx0 + u[c + sj] * dx1) / (dx0 + dx1);
    }
  } else {
    ut = -0.125 * (u[c - sj] + u[c + 2 * sj]) + 1.125 * (u[c] + u[c + sj]);
    vt = -0.125 * (v[c - 1] + v[c + 2]) + 1.125 * (v[c] + v[c + 1]);
  }
  ua[c] = a11[c2] * ut + a12[c2] * vt;
  va[c] = a21[c2] * ut + a22[c2] * vt;
}

static dim3 cell_grid(const Geo& g, int pad) { return dim3((g.n + 2 * pad + 63) / 64, (g.n + 2 * pad + 3) / 4, g.nk); }

static Water6 water6(real* const* w) { return Water6{w[0], w[1], w[2], w[3], w[4], w[5]}; }

int launch_fv_setup_pt(const Geo& g, real* const* water, real* q_con, real* pkz, real* pt, real* cappa,
                       const real* delp, const real* delz, real* dp1, hipStream_t st) {
  hipLaunchKernelGGL(k_fv_setup_pt, cell_grid(g, 0), dim3(64, 4), 0, st, g, water6(water), q_con, pkz, pt, cappa, delp, delz, dp1);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}

int launch_omega_from_w(const Geo& g, const real* delp, const real* delz, const real* w, real* omga, hipStream_t st) {
  hipLaunchKernelGGL(k_omega_from_w, cell_grid(g, 0), dim3(64, 4), 0, st, g, delp, delz, w, omga);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}

int launch_neg_adj3(const Geo& g, real* const* water, real* qcld, real* pt, const real* delp, hipStream_t st) {
  if (g.nk < 4) return PACE_ERR_UNSUPPORTED;
  const double d0_vap = PACE_CV_VAP - PACE_C_LIQ;        // non-hydrostatic (neg_adj3.py:327-332)
  const double lv00 = PACE_HLV - d0_vap * PACE_TICE;
  const Water6 q = water6(water);
  hipLaunchKernelGGL(k_fix_neg_water, cell_grid(g, 0), dim3(64, 4), 0, st, g, q, pt, lv00, d0_vap);
  hipLaunchKernelGGL(k_neg_columns, dim3((g.n + 63) / 64, g.n, 4), dim3(64), 0, st, g, q.qgraupel, q.qrain, q.qvapor, qcld, delp);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}

int launch_c2l(const Geo& g, const Met& m, int order, const real* u, const real* v, const real* a11, const real* a12,
               const real* a21, const real* a22, real* ua, real* va, hipStream_t st) {
  if (order == 2) hipLaunchKernelGGL(k_c2l<2>, cell_grid(g, 1), dim3(64, 4), 0, st, g, m, u, v, a11, a12, a21, a22, ua, va);
  else hipLaunchKernelGGL(k_c2l<4>, cell_grid(g, 0), dim3(64, 4), 0, st, g, m, u, v, a11, a12, a21, a22, ua, va);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}
