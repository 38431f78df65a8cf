// NonhydrostaticVerticalSolver (Fortran Riem_Solver3) with the semi-implicit solver sim1_solver.
// Reference: fv3core/pace/fv3core/stencils/riem_solver3.py:26-321 (precompute / finalize) and
// sim1_solver.py:20-141 -- 3 launches, ~12 sequential k sweeps through 9 stencil temporaries.
//
// Structure (HBM-bound, but only N^2 = 36 864 columns exist at C192, so latency is the enemy):
//   A  column-sequential, arithmetic-free of transcendentals: interface-pressure prefix sums
//   B  fully parallel over (i,j,k): every exp/log of `precompute` and the hydrostatic pressure
//      perturbation pe0 -- 2.9 M threads instead of 37 k
//   C  column-sequential: the two tridiagonal solves (+, -, *, / only), sweeps software-pipelined
//      in register chunks of CH levels so CH independent loads per array are in flight
//   D  fully parallel: the dz update (exp/log), last_call copies
//   E  column-sequential: rebuild zh from the surface
// Lanes run along i in all five kernels, so every level access is a coalesced row.  Recomputable
// temporaries (delta_mass, gamma, g_rat, bb, dd, pe_init, log p) are never stored.
#include "common.h"
#include "kernels.h"

#define RDGAS 287.05
#define GRAV 9.80665
#define RGRAV (1.0 / GRAV)
#define CP_AIR 1004.6
#define KAPPA (RDGAS / CP_AIR)
#ifndef RIEM_CH
#define RIEM_CH 8
#endif
#define CH RIEM_CH

// The solvers' intermediates (workspace fields) are double in BOTH builds: the float32-storage build rounds what the reference's
// fields hold, not the tridiagonal systems' pivots and right-hand sides.
typedef double wreal;
struct Riem3Work {
  wreal *pem, *pm, *w1, *gam, *pp, *aa;
};
#define RIEM3_NFIELDS 6

int64_t riem3_workspace_bytes(const Geo& g) {
  return (int64_t)g.sk * (g.nk + 1) * RIEM3_NFIELDS * (int64_t)sizeof(wreal);
}

#define COLUMN_IJ(g) COLUMN_IJH(g, 0)
#define COLUMN_IJH(g, h)                                        \
  const int i = (g).is - (h) + blockIdx.x * 64 + threadIdx.x;   \
  const int j = (g).js - (h) + blockIdx.y;                      \
  if (i > (g).ie + (h) || j > (g).je + (h)) return;             \
  const long sk = (g).sk;                                   \
  const long c0 = IDX2(g, i, j);                            \
  const int km = (g).nk;                                    \
  (void)km
#define AT(k) (c0 + (long)(k) * sk)

// A: p_interface (-> pem) and the gas-only interface pressure (-> aa, reused later) by prefix sums
//    (riem_solver3.py:63-81 without the logs)
template <int CG>
__global__ void __launch_bounds__(64)
k_riem3_prefix(Geo g, Riem3Work W, double ptop, const real* __restrict__ delp, const real* __restrict__ q_con) {
  COLUMN_IJH(g, CG);
  double p_int = ptop, pg = ptop;
  W.pem[AT(0)] = ptop;
  W.aa[AT(0)] = ptop;
  for (int k0 = 0; k0 < km; k0 += CH) {
    double d_[CH], q_[CH];
#pragma unroll
    for (int t = 0; t < CH; ++t) {
      const int k = (k0 + t < km) ? k0 + t : km - 1;
      d_[t] = delp[AT(k)];
      q_[t] = q_con[AT(k)];
    }
#pragma unroll
    for (int t = 0; t < CH; ++t) {
      const int k = k0 + t;
      if (k < km) {
        p_int = p_int + d_[t];
        pg = pg + d_[t] * (1.0 - q_[t]);
        W.pem[AT(k + 1)] = p_int;
        W.aa[AT(k + 1)] = pg;
      }
    }
  }
}

// B: logs / exps of precompute + first statement of sim1_solver (sim1_solver.py:70-75), all levels in parallel
__global__ void __launch_bounds__(256)
k_riem3_parallel_pre(Geo g, Riem3Work W, int last_call, double peln1, double ptk, const real* __restrict__ cappa,
                     const real* __restrict__ delp, const real* __restrict__ pt, const real* __restrict__ zh,
                     real* __restrict__ delz, real* __restrict__ ppe, real* __restrict__ pk3,
                     real* __restrict__ peln) {
  PLANE_IJK(g);
  if (i < g.is || i > g.ie || j < g.js || j > g.je) return;
  const long c = IDX3(g, i, j, k);
  const int km = g.nk;
  const wreal* pg = W.aa;
  if (k == 0) {
    pk3[c] = ptk;
    if (last_call) peln[c] = peln1;
  } else {
    const double logp = log(W.pem[c]);
    pk3[c] = exp(KAPPA * logp);
    if (last_call) peln[c] = logp;
  }
  if (k < km) {
    const double pg0 = pg[c], pg1 = pg[c + g.sk];
    const double l0 = (k == 0) ? peln1 : log(pg0);
    const double l1 = log(pg1);
    const double pmk = (pg1 - pg0) / (l1 - l0);
    W.pm[c] = pmk;
    const double dz = zh[c + g.sk] - zh[c];
    delz[c] = dz;
    const double dm = delp[c] * RGRAV;
    const double gm = 1.0 / (1.0 - cappa[c]);
    ppe[c] = exp(gm * log(-dm / dz * RDGAS * pt[c])) - pmk;
  }
}

// C: the two tridiagonal systems of sim1_solver (sim1_solver.py:76-132); pe0 arrives in ppe
// CG = 0: D-grid solver on the compute domain, delta_mass = delp * RGRAV (riem_solver3.py:86);
// CG = 1: C-grid solver on compute +- 1, delta_mass = delpc / GRAV (riem_solver_c.py:84)
// (T: real for riem_solver3, whose delz / ppe / w are the caller's fields; wreal for riem_solver_c, which hands workspace fields)
template <int CG, class T>
__global__ void __launch_bounds__(64)
k_riem3_tridiag(Geo g, Riem3Work W, double dt, const real* __restrict__ cappa, const real* __restrict__ ws,
                const T* __restrict__ delz, const real* __restrict__ delp, T* __restrict__ ppe,
                T* __restrict__ w) {
  COLUMN_IJH(g, CG);
  const double t1g = 2.0 * dt * dt, rdt = 1.0 / dt;
#define DM(x) (CG ? (x) / GRAV : (x)*RGRAV)
  // ---- sweep 1 (forward): gam, pp of the first system
  {
    double dm_k = DM(delp[AT(0)]), dm_n = DM(delp[AT(1)]);
    double pe_k = ppe[AT(0)], pe_n = ppe[AT(1)];
    double g_rat_prev = dm_k / dm_n;
    double bet = 2.0 * (1.0 + g_rat_prev);  // bb[0]
    W.pp[AT(0)] = 0.0;
    double pp_prev = 3.0 * (pe_k + g_rat_prev * pe_n) / bet;  // pp[1] = dd[0] / bet
    W.pp[AT(1)] = pp_prev;
    dm_k = dm_n;
    pe_k = pe_n;
    for (int k0 = 1; k0 < km; k0 += CH) {
      double d_[CH], p_[CH];
#pragma unroll
      for (int t = 0; t < CH; ++t) {  // level k+1 inputs
        const int kk = (k0 + t + 1 < km) ? k0 + t + 1 : km - 1;
        d_[t] = delp[AT(kk)];
        p_[t] = ppe[AT(kk)];
      }
#pragma unroll
      for (int t = 0; t < CH; ++t) {
        const int k = k0 + t;
        if (k < km) {
          const double gam = g_rat_prev / bet;
          W.gam[AT(k)] = gam;
          double bb, dd, g_rat;
          if (k < km - 1) {
            dm_n = DM(d_[t]);
            pe_n = p_[t];
            g_rat = dm_k / dm_n;
            bb = 2.0 * (1.0 + g_rat);
            dd = 3.0 * (pe_k + g_rat * pe_n);
            dm_k = dm_n;
            pe_k = pe_n;
          } else {
            g_rat = 0.0;
            bb = 2.0;
            dd = 3.0 * pe_k;
          }
          bet = bb - gam;
          pp_prev = (dd - pp_prev) / bet;
          W.pp[AT(k + 1)] = pp_prev;
          g_rat_prev = g_rat;
        }
      }
    }
  }
  // ---- sweep 2 (backward): pp back-substitution and aa
  {
    double pp_next = W.pp[AT(km)];
    double gm_k = 0.0, dz_k = 0.0;  // values at level k (carried from the previous iteration's k-1 loads)
    bool have = false;
    for (int k0 = km - 1; k0 >= 1; k0 -= CH) {
      double pp_[CH], ga_[CH], cm_[CH], dzm_[CH], pem_[CH], c0_[1], dz0_[1];
#pragma unroll
      for (int t = 0; t < CH; ++t) {
        const int k = (k0 - t >= 1) ? k0 - t : 1;
        pp_[t] = W.pp[AT(k)];
        ga_[t] = W.gam[AT(k)];
        cm_[t] = cappa[AT(k - 1)];
        dzm_[t] = delz[AT(k - 1)];
        pem_[t] = W.pem[AT(k)];
      }
      if (!have) {
        c0_[0] = cappa[AT(k0)];
        dz0_[0] = delz[AT(k0)];
        gm_k = 1.0 / (1.0 - c0_[0]);
        dz_k = dz0_[0];
        have = true;
      }
#pragma unroll
      for (int t = 0; t < CH; ++t) {
        const int k = k0 - t;
        if (k >= 1) {
          const double ppk = pp_[t] - ga_[t] * pp_next;
          W.pp[AT(k)] = ppk;
          const double gm_m = 1.0 / (1.0 - cm_[t]);
          W.aa[AT(k)] = t1g * 0.5 * (gm_m + gm_k) / (dzm_[t] + dz_k) * (pem_[t] + ppk);
          pp_next = ppk;
          gm_k = gm_m;
          dz_k = dzm_[t];
        }
      }
    }
  }
  // ---- sweep 3 (forward): second system, w
  double bet;
  {
    bet = DM(delp[AT(0)]) - W.aa[AT(1)];
    const double w0 = w[AT(0)];
    W.w1[AT(0)] = w0;
    double w_prev = (DM(delp[AT(0)]) * w0 + dt * W.pp[AT(1)]) / bet;
    w[AT(0)] = w_prev;
    double aa_k = W.aa[AT(1)], pp_k = W.pp[AT(1)];
    for (int k0 = 1; k0 < km; k0 += CH) {
      double aan_[CH], ppn_[CH], d_[CH], w_[CH];
#pragma unroll
      for (int t = 0; t < CH; ++t) {
        const int k = (k0 + t < km) ? k0 + t : km - 1;
        aan_[t] = (k + 1 < km) ? W.aa[AT(k + 1)] : 0.0;
        ppn_[t] = W.pp[AT(k + 1)];
        d_[t] = delp[AT(k)];
        w_[t] = w[AT(k)];
      }
#pragma unroll
      for (int t = 0; t < CH; ++t) {
        const int k = k0 + t;
        if (k < km) {
          const double dm = DM(d_[t]);
          const double w1k = w_[t];
          W.w1[AT(k)] = w1k;
          const double gam = aa_k / bet;
          W.gam[AT(k)] = gam;
          if (k < km - 1) {
            bet = dm - (aa_k + aan_[t] + aa_k * gam);
            w_prev = (dm * w1k + dt * (ppn_[t] - pp_k) - aa_k * w_prev) / bet;
          } else {
            const double gmk = 1.0 / (1.0 - cappa[AT(k)]);
            const double p1 = t1g * gmk / delz[AT(k)] * (W.pem[AT(k + 1)] + ppn_[t]);
            bet = dm - (aa_k + p1 + aa_k * gam);
            w_prev = (dm * w1k + dt * (ppn_[t] - pp_k) - p1 * ws[c0] - aa_k * w_prev) / bet;
          }
          w[AT(k)] = w_prev;
          aa_k = aan_[t];
          pp_k = ppn_[t];
        }
      }
    }
  }
  // ---- sweep 4 (backward): w back-substitution
  {
    double w_next = w[AT(km - 1)];
    for (int k0 = km - 2; k0 >= 0; k0 -= CH) {
      double w_[CH], g_[CH];
#pragma unroll
      for (int t = 0; t < CH; ++t) {
        const int k = (k0 - t >= 0) ? k0 - t : 0;
        w_[t] = w[AT(k)];
        g_[t] = W.gam[AT(k + 1)];
      }
#pragma unroll
      for (int t = 0; t < CH; ++t) {
        const int k = k0 - t;
        if (k >= 0) {
          w_next = w_[t] - g_[t] * w_next;
          w[AT(k)] = w_next;
        }
      }
    }
  }
  // ---- sweep 5 (forward): perturbation pressure on interfaces -> ppe
  {
    double pek = 0.0;
    ppe[AT(0)] = 0.0;
    for (int k0 = 1; k0 <= km; k0 += CH) {
      double d_[CH], w_[CH], o_[CH];
#pragma unroll
      for (int t = 0; t < CH; ++t) {
        const int k = (k0 + t <= km) ? k0 + t : km;
        d_[t] = delp[AT(k - 1)];
        w_[t] = w[AT(k - 1)];
        o_[t] = W.w1[AT(k - 1)];
      }
#pragma unroll
      for (int t = 0; t < CH; ++t) {
        const int k = k0 + t;
        if (k <= km) {
          pek = pek + DM(d_[t]) * (w_[t] - o_[t]) * rdt;
          ppe[AT(k)] = pek;
        }
      }
    }
  }
  // ---- sweep 6 (backward): p1 -> pp (reused)
  {
    double p1_next = 0.0;
    double pe1 = ppe[AT(km)], pe2 = 0.0;  // pe[k+1], pe[k+2]
    double dm_n = 0.0;                    // dm[k+1]
    for (int k0 = km - 1; k0 >= 0; k0 -= CH) {
      double p_[CH], d_[CH];
#pragma unroll
      for (int t = 0; t < CH; ++t) {
        const int k = (k0 - t >= 0) ? k0 - t : 0;
        p_[t] = ppe[AT(k)];
        d_[t] = delp[AT(k)];
      }
#pragma unroll
      for (int t = 0; t < CH; ++t) {
        const int k = k0 - t;
        if (k >= 0) {
          const double dm = DM(d_[t]);
          double p1;
          if (k == km - 1) {
            p1 = (p_[t] + 2.0 * pe1) * 1.0 / 3.0;
          } else {
            const double g_rat = dm / dm_n;
            const double bb = 2.0 * (1.0 + g_rat);
            p1 = (p_[t] + bb * pe1 + g_rat * pe2) * 1.0 / 3.0 - g_rat * p1_next;
          }
          W.pp[AT(k)] = p1;
          p1_next = p1;
          pe2 = pe1;
          pe1 = p_[t];
          dm_n = dm;
        }
      }
    }
  }
#undef DM
}

// D: dz update (sim1_solver.py:133-141) + the last_call copies of finalize (riem_solver3.py:136-141)
__global__ void __launch_bounds__(256)
k_riem3_parallel_post(Geo g, Riem3Work W, int last_call, const real* __restrict__ cappa,
                      const real* __restrict__ delp, const real* __restrict__ pt, real* __restrict__ delz,
                      const real* __restrict__ pk3, real* __restrict__ pk, real* __restrict__ pe, double p_fac) {
  PLANE_IJK(g);
  if (i < g.is || i > g.ie || j < g.js || j > g.je) return;
  const long c = IDX3(g, i, j, k);
  if (k < g.nk) {
    const double dm = delp[c] * RGRAV;
    const double p1 = W.pp[c], pmk = W.pm[c];
    // NB: the reference tests p_fac * delta_mass (sim1_solver.py:134), kept as is
    const double maxp = (p_fac * dm > p1 + pmk) ? p_fac * pmk : p1 + pmk;
    delz[c] = -dm * RDGAS * pt[c] * exp((cappa[c] - 1.0) * log(maxp));
  }
  if (last_call) {
    pk[c] = pk3[c];
    pe[c] = W.pem[c];
  }
}

// E: zh from the surface up (riem_solver3.py:142-145)
__global__ void __launch_bounds__(64)
k_riem3_zh(Geo g, const real* __restrict__ zs, const real* __restrict__ delz, real* __restrict__ zh) {
  COLUMN_IJ(g);
  double z = zs[c0];
  zh[AT(km)] = z;
  for (int k0 = km - 1; k0 >= 0; k0 -= CH) {
    double d_[CH];
#pragma unroll
    for (int t = 0; t < CH; ++t) {
      const int k = (k0 - t >= 0) ? k0 - t : 0;
      d_[t] = delz[AT(k)];
    }
#pragma unroll
    for (int t = 0; t < CH; ++t) {
      const int k = k0 - t;
      if (k >= 0) {
        z = z - d_[t];
        zh[AT(k)] = z;
      }
    }
  }
}

int launch_riem_solver3(const Geo& g, void* ws, int last_call, double dt, const real* cappa, double ptop,
                        const real* zs, const real* wsd, real* delz, const real* q_con, const real* delp,
                        const real* pt, real* zh, real* pe, real* ppe, real* pk3, real* pk, real* peln,
                        real* w, double p_fac, hipStream_t st) {
  Riem3Work W;
  wreal* p = (wreal*)ws;
  const long field = g.sk * (g.nk + 1);
  W.pem = p;
  W.pm = p + field;
  W.w1 = p + 2 * field;
  W.gam = p + 3 * field;
  W.pp = p + 4 * field;
  W.aa = p + 5 * field;
  const double peln1 = log(ptop);
  const double ptk = exp(KAPPA * peln1);
  const dim3 cgrid((g.n + 63) / 64, g.n, 1), cblock(64);
  const dim3 pgrid = plane_grid(g, g.nk + 1), pblock(256);
  hipLaunchKernelGGL(k_riem3_prefix<0>, cgrid, cblock, 0, st, g, W, ptop, delp, q_con);
  hipLaunchKernelGGL(k_riem3_parallel_pre, pgrid, pblock, 0, st, g, W, last_call, peln1, ptk, cappa, delp, pt, zh, delz, ppe,
                     pk3, peln);
  hipLaunchKernelGGL((k_riem3_tridiag<0, real>), cgrid, cblock, 0, st, g, W, dt, cappa, wsd, delz, delp, ppe, w);
  hipLaunchKernelGGL(k_riem3_parallel_post, pgrid, pblock, 0, st, g, W, last_call, cappa, delp, pt, delz, pk3, pk, pe, p_fac);
  hipLaunchKernelGGL(k_riem3_zh, cgrid, cblock, 0, st, g, zs, delz, zh);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}

// =================================================================================================
// NonhydrostaticVerticalSolverCGrid (Fortran Riem_Solver_c), riem_solver_c.py:21-250, on compute +- 1.
// Same five-stage structure; w3 is not updated (the solver works on a copy), outputs are gz and pef.
// =================================================================================================
struct RiemCWork {
  Riem3Work r;
  wreal *dz, *pe, *w;
};
#define RIEMC_NFIELDS 9

int64_t riemc_workspace_bytes(const Geo& g) {
  return (int64_t)g.sk * (g.nk + 1) * RIEMC_NFIELDS * (int64_t)sizeof(wreal);
}

// precompute (riem_solver_c.py:21-88) without the prefix sums + first statement of sim1_solver
__global__ void __launch_bounds__(256)
k_riemc_parallel_pre(Geo g, RiemCWork W, const real* __restrict__ cappa, const real* __restrict__ delpc,
                     const real* __restrict__ ptc, const real* __restrict__ gz, const real* __restrict__ w3) {
  PLANE_IJK(g);
  if (i < g.is - 1 || i > g.ie + 1 || j < g.js - 1 || j > g.je + 1 || k >= g.nk) return;
  const long c = IDX3(g, i, j, k);
  const wreal* peg = W.r.aa;
  const double pmk = (peg[c + g.sk] - peg[c]) / log(peg[c + g.sk] / peg[c]);
  W.r.pm[c] = pmk;
  const double dz = gz[c + g.sk] - gz[c];
  W.dz[c] = dz;
  const double dm = delpc[c] / GRAV;
  const double gm = 1.0 / (1.0 - cappa[c]);
  W.pe[c] = exp(gm * log(-dm / dz * RDGAS * ptc[c])) - pmk;
  W.w[c] = w3[c];
}

// sim1_solver.py:133-141 (dz) + finalize (riem_solver_c.py:91-123): pef
__global__ void __launch_bounds__(256)
k_riemc_parallel_post(Geo g, RiemCWork W, double ptop, const real* __restrict__ cappa, const real* __restrict__ delpc,
                      const real* __restrict__ ptc, real* __restrict__ pef, double p_fac) {
  PLANE_IJK(g);
  if (i < g.is - 1 || i > g.ie + 1 || j < g.js - 1 || j > g.je + 1) return;
  const long c = IDX3(g, i, j, k);
  if (k < g.nk) {
    const double dm = delpc[c] / GRAV;
    const double p1 = W.r.pp[c], pmk = W.r.pm[c];
    const double maxp = (p_fac * dm > p1 + pmk) ? p_fac * pmk : p1 + pmk;
    W.dz[c] = -dm * RDGAS * ptc[c] * exp((cappa[c] - 1.0) * log(maxp));
  }
  pef[c] = (k == 0) ? ptop : W.pe[c] + W.r.pem[c];
}

__global__ void __launch_bounds__(64)
k_riemc_gz(Geo g, const real* __restrict__ hs, const wreal* __restrict__ dz, real* __restrict__ gz) {
  COLUMN_IJH(g, 1);
  double z = hs[c0];
  gz[AT(km)] = z;
  for (int k0 = km - 1; k0 >= 0; k0 -= CH) {
    double d_[CH];
#pragma unroll
    for (int t = 0; t < CH; ++t) {
      const int k = (k0 - t >= 0) ? k0 - t : 0;
      d_[t] = dz[AT(k)];
    }
#pragma unroll
    for (int t = 0; t < CH; ++t) {
      const int k = k0 - t;
      if (k >= 0) {
        z = z - d_[t] * GRAV;
        gz[AT(k)] = z;
      }
    }
  }
}

int launch_riem_solver_c(const Geo& g, void* ws, double dt2, const real* cappa, double ptop, const real* hs,
                         const real* ws3, const real* ptc, const real* q_con, const real* delpc, real* gz,
                         real* pef, const real* w3, double p_fac, hipStream_t st) {
  RiemCWork W;
  wreal* p = (wreal*)ws;
  const long field = g.sk * (g.nk + 1);
  W.r.pem = p;
  W.r.pm = p + field;
  W.r.w1 = p + 2 * field;
  W.r.gam = p + 3 * field;
  W.r.pp = p + 4 * field;
  W.r.aa = p + 5 * field;
  W.dz = p + 6 * field;
  W.pe = p + 7 * field;
  W.w = p + 8 * field;
  const dim3 cgrid((g.n + 2 + 63) / 64, g.n + 2, 1), cblock(64);
  const dim3 pgrid = plane_grid(g, g.nk + 1), pblock(256);
  hipLaunchKernelGGL(k_riem3_prefix<1>, cgrid, cblock, 0, st, g, W.r, ptop, delpc, q_con);
  hipLaunchKernelGGL(k_riemc_parallel_pre, pgrid, pblock, 0, st, g, W, cappa, delpc, ptc, gz, w3);
  hipLaunchKernelGGL((k_riem3_tridiag<1, wreal>), cgrid, cblock, 0, st, g, W.r, dt2, cappa, ws3, W.dz, delpc, W.pe, W.w);
  hipLaunchKernelGGL(k_riemc_parallel_post, pgrid, pblock, 0, st, g, W, ptop, cappa, delpc, ptc, pef, p_fac);
  hipLaunchKernelGGL(k_riemc_gz, cgrid, cblock, 0, st, g, hs, W.dz, gz);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}
