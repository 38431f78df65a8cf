// The stencils of LagrangianToEulerian around the vertical remaps (fv3core/pace/fv3core/stencils/remapping.py:42-283,
// moist_cv.py:16-172), for the modes the reference implements (non-hydrostatic, kord_tm < 0) without the saturation
// adjustment.  All of them are local in (i, j) and -- once ps = pe[km] is read -- in k, so each is ONE thread-per-cell
// launch: [k][j][i] storage, lanes along i, 64 x 4 patches.
//   k_l2e_prepare    init_pe + moist_cv_pt_pressure + pn2_pk_delp (3 stencils, 15 computations)
//   k_l2e_post       undo_delz_adjust_and_copy_peln + moist_pkz
//   k_l2e_pressures  pressures_mapu / pressures_mapv (the bottom-pressure broadcast fields never exist)
//   k_l2e_finish     update_ua + copy_from_below, then moist_pt_last_step or the division by pkz
#include "common.h"
#include "kernels.h"

// util/pace/util/constants.py:36-75 (GFS_PHYS branch)
#define PACE_GRAV 9.80665
#define PACE_RDGAS 287.05
#define PACE_RVGAS 461.50
#define PACE_CP_AIR 1004.6
#define PACE_CV_AIR (PACE_CP_AIR - PACE_RDGAS)
#define PACE_RDG (-PACE_RDGAS / PACE_GRAV)
#define PACE_CV_VAP (3.0 * PACE_RVGAS)
#define PACE_C_ICE 1972.0
#define PACE_C_LIQ 4.1855e3

struct L2eWater {
  const real *qvapor, *qliquid, *qrain, *qsnow, *qice, *qgraupel;
};

// moist_cv_nwat6_fn + moist_cvm + set_cappa (moist_cv.py:16-46)
__device__ __forceinline__ void moist_cv(const L2eWater& q, long c, double r_vir, double& gz, double& cappa) {
  const double qv = q.qvapor[c];
  const double ql = q.qliquid[c] + q.qrain[c];
  const double qs = q.qice[c] + q.qsnow[c] + q.qgraupel[c];
  gz = ql + qs;
  const double cvm = (1.0 - (qv + gz)) * PACE_CV_AIR + qv * PACE_CV_VAP + ql * PACE_C_LIQ + qs * PACE_C_ICE;
  cappa = PACE_RDGAS / (PACE_RDGAS + cvm / (1.0 + r_vir * qv));
}

#define L2E_CELL(NJX, NKX)                                       \
  const int i = g.is + blockIdx.x * 64 + threadIdx.x;            \
  const int j = g.js + blockIdx.y * 4 + threadIdx.y;             \
  const int k = blockIdx.z;                                      \
  (void)k;                                                       \
  if (i > g.ie || j > g.je + (NJX)) return;                      \
  const long c2 = IDX2(g, i, j);                                 \
  const long c = c2 + (long)k * g.sk;                            \
  const int km = g.nk;                                           \
  (void)km

__global__ void __launch_bounds__(256)
k_l2e_prepare(Geo g, L2eWater q, real* __restrict__ q_con, real* __restrict__ pt, real* __restrict__ cappa,
              real* __restrict__ delp, real* __restrict__ delz, const real* __restrict__ pe, real* __restrict__ pe1,
              real* __restrict__ pe2, const real* __restrict__ ak, const real* __restrict__ bk, real* __restrict__ dp2,
              real* __restrict__ ps, real* __restrict__ pn2, const real* __restrict__ peln, real* __restrict__ pk,
              double ptop, double akap, double r_vir) {
  L2E_CELL(1, 1);
  // init_pe over the compute domain + the extra row je+1 (remapping.py:42-56)
  pe1[c] = pe[c];
  if (k == 0) pe2[c] = ptop;
  if (k == km) pe2[c] = pe[c];
  if (j > g.je) return;
  const double psv = pe[c2 + (long)km * g.sk];
  if (k == km) {
    ps[c2] = psv;
    pn2[c] = peln[c];
    return;
  }
  // moist_pt_func + delz_adjust (remapping.py:133-154)
  double gz, cp;
  moist_cv(q, c, r_vir, gz, cp);
  q_con[c] = gz;
  cappa[c] = cp;
  const double p = pt[c], dpo = delp[c], dzo = delz[c];
  pt[c] = p * exp(cp / (1.0 - cp) * log(PACE_RDG * dpo / dzo * p));
  delz[c] = -dzo / dpo;
  // pressure_updates + pn2_pk_delp (remapping.py:155-193)
  const double e0 = (k == 0) ? ptop : ak[k] + bk[k] * psv;
  const double e1 = (k + 1 == km) ? psv : ak[k + 1] + bk[k + 1] * psv;
  if (k > 0) pe2[c] = e0;
  const double d = e1 - e0;
  dp2[c] = d;
  delp[c] = d;
  const double lp = log(e0);
  pn2[c] = lp;
  pk[c] = exp(akap * lp);
}

__global__ void __launch_bounds__(256)
k_l2e_post(Geo g, L2eWater q, real* __restrict__ q_con, real* __restrict__ pkz, const real* __restrict__ pt,
           real* __restrict__ cappa, const real* __restrict__ delp, real* __restrict__ delz, real* __restrict__ peln,
           real* __restrict__ pe0, const real* __restrict__ pn2, double r_vir) {
  L2E_CELL(0, 1);
  pe0[c] = peln[c];
  peln[c] = pn2[c];
  if (k == km) return;
  const double dz = -delz[c] * delp[c];
  delz[c] = dz;
  double gz, cp;
  moist_cv(q, c, r_vir, gz, cp);
  q_con[c] = gz;
  cappa[c] = cp;
  pkz[c] = exp(cp * log(PACE_RDG * delp[c] / dz * pt[c]));
}

// dir 0: pressures_mapu (neighbour to the south, window + 1 row); dir 1: pressures_mapv (neighbour to the west, + 1 column)
template <int DIR>
__global__ void __launch_bounds__(256)
k_l2e_pressures(Geo g, const real* __restrict__ pe, const real* __restrict__ pe1, const real* __restrict__ ak,
                const real* __restrict__ bk, real* __restrict__ pe0, real* __restrict__ pe3) {
  const int i = g.is + blockIdx.x * 64 + threadIdx.x;
  const int j = g.js + blockIdx.y * 4 + threadIdx.y;
  const int k = blockIdx.z;
  if (i > g.ie + (DIR == 1 ? 1 : 0) || j > g.je + (DIR == 0 ? 1 : 0)) return;
  const int km = g.nk;
  const long c2 = IDX2(g, i, j);
  const long c = c2 + (long)k * g.sk;
  const long nb = DIR == 0 ? -(long)g.sj : -1L;
  const long cb = c2 + (long)km * g.sk;
  if (DIR == 0) {
    pe0[c] = (k == 0) ? pe[c] : 0.5 * (pe[c + nb] + pe1[c]);
    const double bkh = 0.5 * bk[k];
    pe3[c] = ak[k] + bkh * (pe[cb + nb] + pe[cb]);
  } else {
    if (k == 0) {
      pe3[c] = ak[0];
      pe0[c] = pe[c];
    } else {
      const double bkh = 0.5 * bk[k];
      pe0[c] = 0.5 * (pe[c + nb] + pe[c]);
      pe3[c] = ak[k] + bkh * (pe[cb + nb] + pe[cb]);
    }
  }
}

__global__ void __launch_bounds__(256)
k_l2e_finish(Geo g, L2eWater q, real* __restrict__ pe, const real* __restrict__ pe2, real* __restrict__ pt,
             const real* __restrict__ pkz, double r_vir, int last_step) {
  L2E_CELL(0, 1);
  if (k >= 1 && k < km) pe[c] = pe2[c];  // update_ua + copy_from_below: pe becomes the Eulerian interfaces
  if (last_step) {
    // moist_pt_last_step over km + 1 levels with dtmp = 0 (moist_cv.py:73-122, remapping.py:680-692)
    const double gz = q.qliquid[c] + q.qrain[c] + q.qice[c] + q.qsnow[c] + q.qgraupel[c];
    pt[c] = (pt[c] + 0.0 * pkz[c]) / ((1.0 + r_vir * q.qvapor[c]) * (1.0 - gz));
  } else if (k < km) {
    pt[c] = pt[c] / pkz[c];
  }
}

static dim3 l2e_grid(const Geo& g, int xi, int xj, int nlev) { return dim3((g.n + xi + 63) / 64, (g.n + xj + 3) / 4, nlev); }

int launch_l2e_prepare(const Geo& g, const real* const* water, real* q_con, real* pt, real* cappa, real* delp,
                       real* delz, const real* pe, real* pe1, real* pe2, const real* ak, const real* bk, real* dp2,
                       real* ps, real* pn2, const real* peln, real* pk, double ptop, double akap, double r_vir,
                       hipStream_t st) {
  L2eWater q{water[0], water[1], water[2], water[3], water[4], water[5]};
  hipLaunchKernelGGL(k_l2e_prepare, l2e_grid(g, 0, 1, g.nk + 1), dim3(64, 4), 0, st, g, q, q_con, pt, cappa, delp, delz, pe, pe1,
                     pe2, ak, bk, dp2, ps, pn2, peln, pk, ptop, akap, r_vir);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}

int launch_l2e_post(const Geo& g, const real* const* water, real* q_con, real* pkz, const real* pt, real* cappa,
                    const real* delp, real* delz, real* peln, real* pe0, const real* pn2, double r_vir, hipStream_t st) {
  L2eWater q{water[0], water[1], water[2], water[3], water[4], water[5]};
  hipLaunchKernelGGL(k_l2e_post, l2e_grid(g, 0, 0, g.nk + 1), dim3(64, 4), 0, st, g, q, q_con, pkz, pt, cappa, delp, delz, peln,
                     pe0, pn2, r_vir);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}

int launch_l2e_pressures(const Geo& g, int dir, const real* pe, const real* pe1, const real* ak, const real* bk,
                         real* pe0, real* pe3, hipStream_t st) {
  if (dir == 0) hipLaunchKernelGGL(k_l2e_pressures<0>, l2e_grid(g, 0, 1, g.nk + 1), dim3(64, 4), 0, st, g, pe, pe1, ak, bk, pe0, pe3);
  else hipLaunchKernelGGL(k_l2e_pressures<1>, l2e_grid(g, 1, 0, g.nk + 1), dim3(64, 4), 0, st, g, pe, pe1, ak, bk, pe0, pe3);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}

int launch_l2e_finish(const Geo& g, const real* const* water, real* pe, const real* pe2, real* pt, const real* pkz,
                      double r_vir, int last_step, hipStream_t st) {
  L2eWater q{water[0], water[1], water[2], water[3], water[4], water[5]};
  hipLaunchKernelGGL(k_l2e_finish, l2e_grid(g, 0, 0, g.nk + 1), dim3(64, 4), 0, st, g, q, pe, pe2, pt, pkz, r_vir, last_step);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}
