// Sim1Solver on its own (sim1_solver.py:20-219): the semi-implicit solver as the reference's class exposes it -- gamma, cp3,
// delta_mass, pm, pem, potential temperature in; pe out; w, dz inout -- on the compute domain widened by n_halo.
// NOT the hot path: inside riem_solver3 / riem_solver_c the same arithmetic is part of the k-cooperative column kernel
// (k_riem3f.hip).  This one is a plain thread-per-column Thomas solve with its work columns in the caller's workspace
// (five fields), written for the stand-alone class and its test.
#include "common.h"
#include "kernels.h"

#define RDGAS 287.05  // constants.py:RDGAS (as in k_riem3.hip)

struct Sim1Work {
  real *pp, *gam, *aa, *w1, *p1;
};

int64_t sim1_workspace_bytes(const Geo& g) { return 5 * (int64_t)g.sk * (g.nk + 1) * (int64_t)sizeof(real); }

__global__ void __launch_bounds__(64)
k_sim1_solver(Geo g, Sim1Work W, int halo, double dt, double p_fac, const real* __restrict__ gm, const real* __restrict__ cp3,
              real* __restrict__ pe, const real* __restrict__ dm, const real* __restrict__ pm,
              const real* __restrict__ pem, real* __restrict__ w, real* __restrict__ dz, const real* __restrict__ pt,
              const real* __restrict__ ws) {
  const int span = g.n + 2 * halo;
  const int t = (int)(blockIdx.x * 64 + threadIdx.x);
  if (t >= span * span) return;
  const int j = g.js - halo + t / span, i = g.is - halo + t % span;
  const long c0 = IDX2(g, i, j);
  const int km = g.nk;
  const double t1g = 2.0 * dt * dt, rdt = 1.0 / dt;
#define AT(k) (c0 + (long)(k)*g.sk)
  // :57-62 (PARALLEL): the pressure of the layers, w1 = w
  for (int k = 0; k < km; ++k) {
    pe[AT(k)] = exp(gm[AT(k)] * log(-dm[AT(k)] / dz[AT(k)] * RDGAS * pt[AT(k)])) - pm[AT(k)];
    W.w1[AT(k)] = w[AT(k)];
  }
  // :63-87: first system (pp on interfaces), forward
  double g_prev = dm[AT(0)] / dm[AT(1)];
  double bet = 2.0 * (1.0 + g_prev);
  W.pp[AT(0)] = 0.0;
  double pp_prev = 3.0 * (pe[AT(0)] + g_prev * pe[AT(1)]) / bet;
  W.pp[AT(1)] = pp_prev;
  for (int k = 1; k < km; ++k) {
    const double gam = g_prev / bet;
    W.gam[AT(k)] = gam;
    double g_rat, bb, dd;
    if (k < km - 1) {
      g_rat = dm[AT(k)] / dm[AT(k + 1)];
      bb = 2.0 * (1.0 + g_rat);
      dd = 3.0 * (pe[AT(k)] + g_rat * pe[AT(k + 1)]);
    } else {
      g_rat = 0.0;
      bb = 2.0;
      dd = 3.0 * pe[AT(k)];
    }
    bet = bb - gam;
    pp_prev = (dd - pp_prev) / bet;
    W.pp[AT(k + 1)] = pp_prev;
    g_prev = g_rat;
  }
  // :88-99: backward, pp and aa
  for (int k = km - 1; k >= 1; --k) {
    const double ppk = W.pp[AT(k)] - W.gam[AT(k)] * W.pp[AT(k + 1)];
    W.pp[AT(k)] = ppk;
    W.aa[AT(k)] = t1g * 0.5 * (gm[AT(k - 1)] + gm[AT(k)]) / (dz[AT(k - 1)] + dz[AT(k)]) * (pem[AT(k)] + ppk);
  }
  // :100-124: second system (w), forward
  bet = dm[AT(0)] - W.aa[AT(1)];
  double w_prev = (dm[AT(0)] * W.w1[AT(0)] + dt * W.pp[AT(1)]) / bet;
  w[AT(0)] = w_prev;
  for (int k = 1; k < km; ++k) {
    const double aa = W.aa[AT(k)];
    const double gam = aa / bet;
    W.gam[AT(k)] = gam;
    if (k < km - 1) {
      bet = dm[AT(k)] - (aa + W.aa[AT(k + 1)] + aa * gam);
      w_prev = (dm[AT(k)] * W.w1[AT(k)] + dt * (W.pp[AT(k + 1)] - W.pp[AT(k)]) - aa * w_prev) / bet;
    } else {
      const double p1 = t1g * gm[AT(k)] / dz[AT(k)] * (pem[AT(k + 1)] + W.pp[AT(k + 1)]);
      bet = dm[AT(k)] - (aa + p1 + aa * gam);
      w_prev = (dm[AT(k)] * W.w1[AT(k)] + dt * (W.pp[AT(k + 1)] - W.pp[AT(k)]) - p1 * ws[c0] - aa * w_prev) / bet;
    }
    w[AT(k)] = w_prev;
  }
  // :125-126: backward
  for (int k = km - 2; k >= 0; --k) w[AT(k)] = w[AT(k)] - W.gam[AT(k + 1)] * w[AT(k + 1)];
  // :127-131: perturbation pressure on the interfaces
  pe[AT(0)] = 0.0;
  for (int k = 1; k <= km; ++k) pe[AT(k)] = pe[AT(k - 1)] + dm[AT(k - 1)] * (w[AT(k - 1)] - W.w1[AT(k - 1)]) * rdt;
  // :132-141: p1 (backward), then dz
  {
    const int k = km - 1;
    W.p1[AT(k)] = (pe[AT(k)] + 2.0 * pe[AT(k + 1)]) * 1.0 / 3.0;
  }
  for (int k = km - 2; k >= 0; --k) {
    const double g_rat = dm[AT(k)] / dm[AT(k + 1)];
    const double bb = 2.0 * (1.0 + g_rat);
    W.p1[AT(k)] = (pe[AT(k)] + bb * pe[AT(k + 1)] + g_rat * pe[AT(k + 2)]) * 1.0 / 3.0 - g_rat * W.p1[AT(k + 1)];
  }
  for (int k = 0; k < km; ++k) {
    const double p1 = W.p1[AT(k)];
    // NB the reference compares p_fac * delta_mass (not p_fac * pm) -- sim1_solver.py:134
    const double maxp = (p_fac * dm[AT(k)] > p1 + pm[AT(k)]) ? p_fac * pm[AT(k)] : p1 + pm[AT(k)];
    dz[AT(k)] = -dm[AT(k)] * RDGAS * pt[AT(k)] * exp((cp3[AT(k)] - 1.0) * log(maxp));
  }
#undef AT
}

int launch_sim1_solver(const Geo& g, void* ws_, int n_halo, double dt, double p_fac, const real* gamma, const real* cp3,
                       real* pe, const real* delta_mass, const real* pm, const real* pem, real* w, real* dz,
                       const real* pt, const real* ws, hipStream_t st) {
  real* p = (real*)ws_;
  const long field = (long)g.sk * (g.nk + 1);
  Sim1Work W{p, p + field, p + 2 * field, p + 3 * field, p + 4 * field};
  const int span = g.n + 2 * n_halo;
  hipLaunchKernelGGL(k_sim1_solver, dim3((unsigned)((span * span + 63) / 64)), dim3(64), 0, st, g, W, n_halo, dt, p_fac, gamma,
                     cp3, pe, delta_mass, pm, pem, w, dz, pt, ws);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}
