// MapSingle (Fortran map_single / map1_ppm / map_scalar): conservative remapping of one field from the Lagrangian
// layers bounded by pe1 to the layers bounded by pe2, with the cubic-spline / PPM sub-grid profile of RemapProfile
// (cs_profile).  Reference: fv3core/pace/fv3core/stencils/map_single.py:14-200 (3 stencils + RemapProfile),
// remap_profile.py:154-681 (3 stencils, 6 three-dimensional work fields + 3 boolean ones).
//
// Column-local work, [k][j][i] storage: lanes run along i, so every level access of a wave is one coalesced row.
//   k_remap_interfaces   one thread per column: the tridiagonal solve for the interface values q (forward elimination
//                        + back substitution, remap_profile.py:176-253), and the copy of the field (a4_1)
//   k_remap_coefficients one thread per CELL: everything after the solve is local in k once gam is recognised as the
//                        difference of a4_1 (remap_profile.py:295) -- the limited interface values of the four
//                        neighbouring interfaces, the extremum flags of the three neighbouring layers and the layer's
//                        own coefficients are evaluated in registers; a4_2, a4_3, a4_4 are the only fields written
//                        (the reference's q, gam, tmp, tmp2, extm, ext5, ext6 never exist)
//   k_remap_layers       the walk over the source layers (map_single.py:44-93), one thread per column and block of 8
//                        target levels: the reference's running search index is re-derived by bisection (see there)
// kord 9 and 10 (the reference asserts |kord| <= 10; kord < 9 is not implemented), every iv.
#include "common.h"
#include "kernels.h"

struct RemapWin {
  int i0, i1, j0, j1;  // inclusive window of columns
  int km;
};

// Fields remapped by one launch sequence (MapNTracer: all tracers share pe1 / pe2; MapSingle: one).  Field f uses the
// f-th group of five work fields.
#define REMAP_MAXQ 16
#define REMAP_NFIELDS 5
#define REMAP_KB 8  // target levels per thread in k_remap_layers
struct RemapBatch {
  real* q[REMAP_MAXQ];
  int n;
};

// ---- the three constraint functions (remap_profile.py:52-151), on scalars ----
__device__ __forceinline__ void posdef_constraint_iv0(double a1, double& a2, double& a3, double& a4) {
  if (a1 <= 0.0) {
    a2 = a1; a3 = a1; a4 = 0.0;
  } else if (fabs(a3 - a2) < -a4 && (a1 + 0.25 * ((a3 - a2) * (a3 - a2)) / a4 + a4 * (1.0 / 12.0)) < 0.0) {
    if (a1 < a3 && a1 < a2) {
      a3 = a1; a2 = a1; a4 = 0.0;
    } else if (a3 > a2) {
      a4 = 3.0 * (a2 - a1);
      a3 = a2 - a4;
    } else {
      a4 = 3.0 * (a3 - a1);
      a2 = a3 - a4;
    }
  }
}

__device__ __forceinline__ void posdef_constraint_iv1(double a1, double& a2, double& a3, double& a4) {
  const double da1 = a3 - a2;
  const double da2 = da1 * da1;
  const double a6da = a4 * da1;
  if (((a1 - a2) * (a1 - a3)) >= 0.0) {
    a2 = a1; a3 = a1; a4 = 0.0;
  } else if (a6da < -1.0 * da2) {
    a4 = 3.0 * (a2 - a1);
    a3 = a2 - a4;
  } else if (a6da > da2) {
    a4 = 3.0 * (a3 - a1);
    a2 = a3 - a4;
  }
}

__device__ __forceinline__ void remap_constraint(double a1, double& a2, double& a3, double& a4, bool extm) {
  const double da1 = a3 - a2;
  const double da2 = da1 * da1;
  const double a6da = a4 * da1;
  if (extm) {
    a2 = a1; a3 = a1; a4 = 0.0;
  } else if (a6da < -da2) {
    a4 = 3.0 * (a2 - a1);
    a3 = a2 - a4;
  } else if (a6da > da2) {
    a4 = 3.0 * (a3 - a1);
    a2 = a3 - a4;
  }
}

__device__ __forceinline__ double min3(double a, double p, double l) { return (a < p && a < l) ? a : (p < l ? p : l); }
__device__ __forceinline__ double max3(double a, double p, double l) { return (a > p && a > l) ? a : (p > l ? p : l); }

// ---------------------------------------------------------------------------------------------------------------
// interface values: set_initial_vals.  qi gets km + 1 levels, gw (work) km; a1 = copy of q1 (km levels).
// ---------------------------------------------------------------------------------------------------------------
template <bool IVM2>
__global__ void __launch_bounds__(64)
k_remap_interfaces(Geo g, RemapWin w, RemapBatch B, const real* __restrict__ pe1, const real* __restrict__ qs,
                   real* __restrict__ ws) {
  const int i = w.i0 + blockIdx.x * 64 + threadIdx.x;
  const int j = w.j0 + blockIdx.y;
  if (i > w.i1 || j > w.j1) return;
  const long c0 = IDX2(g, i, j);
  const long sk = g.sk;
  const int km = w.km;
  const long field = sk * (g.nk + 1);
  const real* __restrict__ q1 = B.q[blockIdx.z];
  real* __restrict__ a1 = ws + (long)blockIdx.z * REMAP_NFIELDS * field;
  real* __restrict__ qi = a1 + field;
  real* __restrict__ gw = qi + field;  // (a2's storage: the back substitution's multipliers live there until the
                                         // coefficient kernel overwrites it)
#define Q1(k) q1[c0 + (long)(k) * sk]
#define DP(k) (pe1[c0 + (long)((k) + 1) * sk] - pe1[c0 + (long)(k) * sk])  // set_dp, map_single.py:14-18
#define QI(k) qi[c0 + (long)(k) * sk]
#define GW(k) gw[c0 + (long)(k) * sk]
  // Loads are issued a chunk of RC levels ahead of the recurrences that consume them (the loop-carried values are two
  // doubles; what a level costs otherwise is the latency of its own loads, with less than one wave per SIMD to hide it).
  constexpr int RC = 8;
  double pch[RC + 3], qch[RC + 1];
  if (IVM2) {
    // remap_profile.py:183-239 for iv == -2: gw[k] = gam[k]
    const double qsv = qs[c0];
    double qp = 0.0, gam = 0.5;
    for (int kc = 0; kc < km; kc += RC) {
#pragma unroll
      for (int t = 0; t < RC + 3; ++t) {
        const int l = kc - 2 + t;  // pch[t] = pe1[kc - 2 + t]
        pch[t] = (l >= 0 && l <= km) ? pe1[c0 + (long)l * sk] : 0.0;
      }
#pragma unroll
      for (int t = 0; t < RC + 1; ++t) {
        const int l = kc - 1 + t;  // qch[t] = q1[kc - 1 + t]
        qch[t] = (l >= 0 && l < km) ? Q1(l) : 0.0;
      }
#pragma unroll
      for (int t = 0; t < RC; ++t) {
        const int k = kc + t;
        if (k >= km) break;
        const double q0 = qch[t + 1], qm = qch[t];
        a1[c0 + (long)k * sk] = q0;
        const double pm2 = pch[t], pm1 = pch[t + 1], p0 = pch[t + 2], pp1 = pch[t + 3];  // pe1[k-2 .. k+1]
        if (k == 0) {
          qp = 1.5 * q0;
        } else if (k == 1) {
          gam = 0.5;
          GW(1) = gam;
          const double gr = (p0 - pm1) / (pp1 - p0);
          const double bet = 2.0 + gr + gr - gam;
          qp = (3.0 * (qm + q0) - qp) / bet;
        } else {
          const double old_gr = (pm1 - pm2) / (p0 - pm1);
          const double old_bet = 2.0 + old_gr + old_gr - gam;
          gam = old_gr / old_bet;
          GW(k) = gam;
          const double gr = (p0 - pm1) / (pp1 - p0);
          if (k < km - 1) {
            const double bet = 2.0 + gr + gr - gam;
            qp = (3.0 * (qm + q0) - qp) / bet;
          } else {
            qp = (3.0 * (qm + q0) - gr * qsv - qp) / (2.0 + gr + gr - gam);
          }
        }
        QI(k) = qp;
      }
    }
    QI(km) = qsv;
    double qn = qp;  // = q[km-1]
    for (int kc = km - 2; kc >= 0; kc -= RC) {
      double qc[RC], gc[RC];
#pragma unroll
      for (int t = 0; t < RC; ++t) {
        const int k = kc - t;
        qc[t] = k >= 0 ? QI(k) : 0.0;
        gc[t] = k >= 0 ? GW(k + 1) : 0.0;
      }
#pragma unroll
      for (int t = 0; t < RC; ++t) {
        const int k = kc - t;
        if (k < 0) break;
        qn = qc[t] - gc[t] * qn;
        QI(k) = qn;
      }
    }
  } else {
    // remap_profile.py:188-253 for iv != -2
    double gam = 0.0, qp = 0.0;
    for (int kc = 0; kc < km; kc += RC) {
#pragma unroll
      for (int t = 0; t < RC + 2; ++t) {
        const int l = kc - 1 + t;  // pch[t] = pe1[kc - 1 + t]
        pch[t] = (l >= 0 && l <= km) ? pe1[c0 + (long)l * sk] : 0.0;
      }
#pragma unroll
      for (int t = 0; t < RC + 1; ++t) {
        const int l = kc - 1 + t;  // qch[t] = q1[kc - 1 + t]
        qch[t] = (l >= 0 && l < km) ? Q1(l) : 0.0;
      }
#pragma unroll
      for (int t = 0; t < RC; ++t) {
        const int k = kc + t;
        if (k >= km) break;
        const double q0 = qch[t + 1], qm = qch[t];
        a1[c0 + (long)k * sk] = q0;
        const double pm1 = pch[t], p0 = pch[t + 1], pp1 = pch[t + 2];  // pe1[k-1], pe1[k], pe1[k+1]
        if (k == 0) {
          const double pp2 = pch[t + 3];  // pe1[2]
          const double gr = (pp2 - pp1) / (pp1 - p0);
          const double bet = gr * (gr + 0.5);
          qp = ((gr + gr) * (gr + 1.0) * q0 + qch[t + 2]) / bet;
          gam = (1.0 + gr * (gr + 1.5)) / bet;
        } else {
          const double d4 = (p0 - pm1) / (pp1 - p0);
          const double bet = 2.0 + d4 + d4 - gam;
          qp = (3.0 * (qm + d4 * q0) - qp) / bet;
          gam = d4 / bet;
        }
        QI(k) = qp;
        GW(k) = gam;
      }
    }
    {
      const double d4 = DP(km - 2) / DP(km - 1);
      const double a_bot = 1.0 + d4 * (d4 + 1.5);
      qp = (2.0 * d4 * (d4 + 1.0) * Q1(km - 1) + Q1(km - 2) - a_bot * qp) / (d4 * (d4 + 0.5) - a_bot * gam);
      QI(km) = qp;
    }
    double qn = qp;
    for (int kc = km - 1; kc >= 0; kc -= RC) {
      double qc[RC], gc[RC];
#pragma unroll
      for (int t = 0; t < RC; ++t) {
        const int k = kc - t;
        qc[t] = k >= 0 ? QI(k) : 0.0;
        gc[t] = k >= 0 ? GW(k) : 0.0;
      }
#pragma unroll
      for (int t = 0; t < RC; ++t) {
        const int k = kc - t;
        if (k < 0) break;
        qn = qc[t] - gc[t] * qn;
        QI(k) = qn;
      }
    }
  }
#undef Q1
#undef DP
#undef QI
#undef GW
}

// ---------------------------------------------------------------------------------------------------------------
// coefficients: apply_constraints + set_interpolation_coefficients, one thread per cell
// ---------------------------------------------------------------------------------------------------------------
struct RemapCol {
  const real* a1;  // column base (level 0)
  const real* qi;
  long sk;
  int km;
  __device__ __forceinline__ double A1(int k) const { return a1[(long)k * sk]; }
  // gam of apply_constraints (remap_profile.py:295), levels 1 .. km-1
  __device__ __forceinline__ double gam(int k) const { return A1(k) - A1(k - 1); }
  // the limited interface value q[k] that apply_constraints leaves (remap_profile.py:290-323), k = 0 .. km
  template <int IV>
  __device__ __forceinline__ double qcon(int k) const {
    double q = qi[(long)k * sk];
    if (k == 0 || k == km) return q;
    const double am = A1(k - 1), a0 = A1(k);
    const double tmp = am > a0 ? am : a0;
    const double tmp2 = am < a0 ? am : a0;
    if (k == 1 || k == km - 1) {
      if (q >= tmp) q = tmp;
      if (q <= tmp2) q = tmp2;
      return q;
    }
    const double gm = gam(k - 1), gp = gam(k + 1);
    if (gm * gp > 0) {
      if (q >= tmp) q = tmp;
      if (q <= tmp2) q = tmp2;
    } else if (gm > 0) {
      if (q <= tmp2) q = tmp2;
    } else {
      if (q >= tmp) q = tmp;
      if (IV == 0) {
        if (q < 0.0) q = 0.0;
      }
    }
    return q;
  }
  __device__ __forceinline__ bool extm_inner(int k) const { return gam(k) * gam(k + 1) < 0.0; }  // k = 1 .. km-2
};

template <int KORD, int IV>
__global__ void __launch_bounds__(256)
k_remap_coefficients(Geo g, RemapWin w, real* __restrict__ ws, double qmin) {
  const int i = w.i0 + blockIdx.x * 64 + threadIdx.x;
  const int j = w.j0 + blockIdx.y * 4 + threadIdx.y;
  const int km = w.km;
  const int f = blockIdx.z / km, k = blockIdx.z - f * km;
  if (i > w.i1 || j > w.j1) return;
  const long c0 = IDX2(g, i, j);
  const long field = g.sk * (g.nk + 1);
  const real* __restrict__ a1 = ws + (long)f * REMAP_NFIELDS * field;
  const real* __restrict__ qi = a1 + field;
  real* __restrict__ a2o = ws + ((long)f * REMAP_NFIELDS + 2) * field;
  real* __restrict__ a3o = a2o + field;
  real* __restrict__ a4o = a3o + field;
  RemapCol C{a1 + c0, qi + c0, g.sk, km};
  const double A1 = C.A1(k);
  double a2 = C.template qcon<IV>(k), a3 = C.template qcon<IV>(k + 1), a4 = 0.0;
  if (k <= 1) {
    // the top two layers (remap_profile.py:370-405)
    if (k == 0) {
      if (IV == 0) {
        if (a2 < 0.0) a2 = 0.0;
      }
      if (IV == -1) {
        if (a2 * A1 <= 0.0) a2 = 0.0;
      }
      if (IV == 2) {
        a2 = A1; a3 = A1; a4 = 0.0;
      } else {
        a4 = 3.0 * (2.0 * A1 - (a2 + a3));
        posdef_constraint_iv1(A1, a2, a3, a4);
      }
    } else {
      a4 = 3.0 * (2.0 * A1 - (a2 + a3));
      remap_constraint(A1, a2, a3, a4, C.extm_inner(1));
    }
  } else if (k >= km - 2) {
    // the bottom two layers (remap_profile.py:546-563)
    if (k == km - 1) {
      if (IV == 0) {
        if (a3 < 0.0) a3 = 0.0;
      }
      if (IV == -1) {
        if (a3 * A1 <= 0.0) a3 = 0.0;
      }
    }
    a4 = 3.0 * (2.0 * A1 - (a2 + a3));
    if (k == km - 2) remap_constraint(A1, a2, a3, a4, C.extm_inner(km - 2));
    else posdef_constraint_iv1(A1, a2, a3, a4);
  } else {
    // inner layers, 2 .. km-3
    const double g0 = C.gam(k), gm = C.gam(k - 1), gp = C.gam(k + 1), gpp = C.gam(k + 2);
    const double pmp_1 = A1 - 2.0 * gp;
    const double lac_1 = pmp_1 + 1.5 * gpp;
    const double pmp_2 = A1 + 2.0 * g0;
    const double lac_2 = pmp_2 - 1.5 * gm;
    if (KORD == 9) {
      // set_inner_as_kord9 (remap_profile.py:449-498)
      const bool e0 = C.extm_inner(k), em = C.extm_inner(k - 1), ep = C.extm_inner(k + 1);
      if ((e0 && em) || (e0 && ep) || (e0 && (qmin > 0.0 && A1 < qmin))) {
        a2 = A1; a3 = A1; a4 = 0.0;
      } else {
        a4 = 6.0 * A1 - 3.0 * (a2 + a3);
        if (fabs(a4) > fabs(a2 - a3)) {
          double tmin = min3(A1, pmp_1, lac_1);
          double tmax0 = a2 > tmin ? a2 : tmin;
          double tmax = max3(A1, pmp_1, lac_1);
          a2 = tmax0 < tmax ? tmax0 : tmax;
          tmin = min3(A1, pmp_2, lac_2);
          tmax0 = a3 > tmin ? a3 : tmin;
          tmax = max3(A1, pmp_2, lac_2);
          a3 = tmax0 < tmax ? tmax0 : tmax;
          a4 = 6.0 * A1 - 3.0 * (a2 + a3);
        }
      }
    } else {
      // set_exts (remap_profile.py:331-338) of the three layers + set_inner_as_kord10 (:500-544)
      bool e5[3], e6[3];
#pragma unroll
      for (int d = -1; d <= 1; ++d) {
        const double b1 = C.A1(k + d);
        const double b2 = d == 0 ? a2 : C.template qcon<IV>(k + d);
        const double b3 = d == 0 ? a3 : (d == -1 ? a2 : C.template qcon<IV>(k + 2));
        const double x0 = 2.0 * b1 - (b2 + b3);
        const double x1 = fabs(b2 - b3);
        e5[d + 1] = fabs(x0) > x1;
        e6[d + 1] = fabs(3.0 * x0) > x1;
      }
      const double tmin2 = min3(A1, pmp_1, lac_1);
      const double tmax2 = max3(A1, pmp_1, lac_1);
      const double t2 = a2 > tmin2 ? a2 : tmin2;
      double tmin3 = A1 < pmp_2 ? A1 : pmp_2;
      tmin3 = lac_2 < tmin3 ? lac_2 : tmin3;
      double tmax3 = A1 > pmp_2 ? A1 : pmp_2;
      tmax3 = lac_2 > tmax3 ? lac_2 : tmax3;
      const double t3 = a3 > tmin3 ? a3 : tmin3;
      const bool n5 = e5[0] || e5[2], n6 = e6[0] || e6[2];
      if (e5[1]) {
        if (n5) {
          a2 = A1; a3 = A1;
        } else if (n6) {
          a2 = t2 < tmax2 ? t2 : tmax2;
          a3 = t3 < tmax3 ? t3 : tmax3;
        }
      } else if (e6[1]) {
        if (n5) {
          a2 = t2 < tmax2 ? t2 : tmax2;
          a3 = t3 < tmax3 ? t3 : tmax3;
        }
      }
      a4 = 3.0 * (2.0 * A1 - (a2 + a3));
    }
    if (IV == 0) posdef_constraint_iv0(A1, a2, a3, a4);
  }
  const long c = c0 + (long)k * g.sk;
  a2o[c] = a2;
  a3o[c] = a3;
  a4o[c] = a4;
}

// ---------------------------------------------------------------------------------------------------------------
// lagrangian_contributions (map_single.py:21-93)
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_remap_layers(Geo g, RemapWin w, RemapBatch B, const real* __restrict__ pe1, const real* __restrict__ pe2,
               const real* __restrict__ ws) {
  const int i = w.i0 + blockIdx.x * 64 + threadIdx.x;
  const int j = w.j0 + blockIdx.y;
  if (i > w.i1 || j > w.j1) return;
  const long c0 = IDX2(g, i, j);
  const long sk = g.sk;
  const int km = w.km;
  const long field = sk * (g.nk + 1);
  const int nblk = (km + REMAP_KB - 1) / REMAP_KB;
  const int fld = blockIdx.z / nblk, kblk = blockIdx.z - fld * nblk;
  real* __restrict__ q = B.q[fld];
  const real* __restrict__ a1 = ws + (long)fld * REMAP_NFIELDS * field;
  const real* __restrict__ a2 = a1 + 2 * field;
  const real* __restrict__ a3 = a2 + field;
  const real* __restrict__ a4 = a3 + field;
#define AT(f, l) f[c0 + (long)(l) * sk]
  // The reference walks the column with one running source-layer index.  Its value when level k begins is
  // min { L : pe1[L+1] >= pe2[k] } for every k >= 1 (it only ever advances past layers that end strictly above the target
  // interface, map_single.py:47,65), and 0 for k = 0 -- so a block of levels can start on its own: bisection instead of the
  // walk from the top, and the column's levels are spread over km / REMAP_KB threads.
  const int k0 = kblk * REMAP_KB, k1 = (k0 + REMAP_KB < km) ? k0 + REMAP_KB : km;
  double p2a = AT(pe2, k0);
  int L = 0;
  if (k0 > 0) {
    int lo = 0, hi = km - 1;  // the answer lies in [lo, hi]
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (AT(pe1, mid + 1) >= p2a) hi = mid;
      else lo = mid + 1;
    }
    L = lo;
  }
  double p1a = AT(pe1, L), p1b = AT(pe1, L + 1);  // the source layer's bounds
  for (int k = k0; k < k1; ++k) {
    // Memory safety on inputs the operator is not defined for (non-monotone or non-finite coordinates, a target column that
    // reaches below the source's: the reference indexes out of bounds there): the source index never leaves the column.  On
    // valid input L <= km - 1 always holds and this does nothing.
    if (L > km - 1) {
      L = km - 1;
      p1a = AT(pe1, L);
      p1b = AT(pe1, L + 1);
    }
    const double p2b = AT(pe2, k + 1);
    double dpl = p1b - p1a;
    const double pl = (p2a - p1a) / dpl;
    double out;
    if (p2b <= p1b) {
      const double pr = (p2b - p1a) / dpl;
      const double b2 = AT(a2, L), b3 = AT(a3, L), b4 = AT(a4, L);
      out = b2 + 0.5 * (b4 + b3 - b2) * (pr + pl) - b4 * 1.0 / 3.0 * (pr * (pr + pl) + pl * pl);
    } else {
      double qsum;
      {
        const double b2 = AT(a2, L), b3 = AT(a3, L), b4 = AT(a4, L);
        qsum = (p1b - p2a) * (b2 + 0.5 * (b4 + b3 - b2) * (1.0 + pl) - b4 * 1.0 / 3.0 * (1.0 + pl * (1.0 + pl)));
      }
      L = L + 1;
      p1a = p1b;
      p1b = (L + 1 <= km) ? AT(pe1, L + 1) : NAN;
      while (p1b < p2b) {
        qsum += (p1b - p1a) * AT(a1, L);
        L = L + 1;
        p1a = p1b;
        p1b = (L + 1 <= km) ? AT(pe1, L + 1) : NAN;
      }
      const double dp = p2b - p1a;
      dpl = p1b - p1a;
      const double esl = dp / dpl;
      const int Lc = L < km ? L : km - 1;  // (L == km only on invalid input, see above)
      const double b2 = AT(a2, Lc), b3 = AT(a3, Lc), b4 = AT(a4, Lc);
      qsum += dp * (b2 + 0.5 * esl * (b3 - b2 + b4 * (1.0 - (2.0 / 3.0) * esl)));
      out = qsum / (p2b - p2a);
    }
    AT(q, k) = out;
    p2a = p2b;
  }
#undef AT
}

// ---------------------------------------------------------------------------------------------------------------
// FillNegativeTracerValues / fix_tracer (fillz.py:15-117, Fortran fillz).  The reference runs five sequential and four
// parallel computations over the column with five work fields; here one forward sweep carries the two levels in flight
// in registers (a level is final once the level below has decided how much it borrows from it), and a second sweep
// rescales the column only where something was fixed.  One thread per (column, tracer).
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_fillz(Geo g, RemapWin w, RemapBatch B, const real* __restrict__ dpf) {
  const int i = w.i0 + blockIdx.x * 64 + threadIdx.x;
  const int j = w.j0 + blockIdx.y;
  if (i > w.i1 || j > w.j1) return;
  const long c0 = IDX2(g, i, j);
  const long sk = g.sk;
  const int km = w.km;
  real* __restrict__ q = B.q[blockIdx.z];
#define Q(k) q[c0 + (long)(k) * sk]
#define DPK(k) dpf[c0 + (long)(k) * sk]
  int zfix = 0;
  double sum0 = 0.0, sum1 = 0.0;
  // fix_top (fillz.py:41-52)
  double dpm = DPK(0), dpk = DPK(1);
  double qm = Q(0), qk = Q(1);
  if (qm < 0.0) qk = qk + qm * dpm / dpk;
  if (qm < 0) qm = 0;
  double lower_prev = 0.0;
  double qn = Q(2), dpn = DPK(2);
  for (int k = 1; k < km; ++k) {
    // here: qm = q[k-1] as the reference's fix_interior leaves it, qk = q[k] not yet treated, qn = q[k+1] (original)
    double upper = 0.0, lower = 0.0;
    if (lower_prev != 0.0) qk = qk - (lower_prev / dpk);
    if (k < km - 1) {
      // fix_interior (fillz.py:54-79)
      if (qk < 0.0) {
        zfix += 1;
        if (qm > 0.0) {
          const double dq = (qm * dpm < -(qk * dpk)) ? qm * dpm : -(qk * dpk);
          qk = qk + dq / dpk;
          upper = dq;
        }
        if (qk < 0.0 && qn > 0.0) {
          const double dq = (qn * dpn < -(qk * dpk)) ? qn * dpn : -(qk * dpk);
          qk = qk + dq / dpk;
          lower = dq;
        }
      }
    } else {
      // fix_bottom (fillz.py:87-100)
      const double qup = qm * dpm;
      const double qly = -qk * dpk;
      const double dup = qup < qly ? qup : qly;
      if (qk < 0.0 && qm > 0.0) {
        zfix += 1;
        qk = qk + (dup / dpk);
        upper = dup;
      }
    }
    // level k-1 is final now (fillz.py:80-85, 101-106)
    if (upper != 0.0) qm = qm - upper / dpm;
    Q(k - 1) = qm;
    if (k - 1 >= 1) {
      const double dm = qm * dpm;
      sum0 += dm;
      sum1 += dm > 0.0 ? dm : 0.0;
    }
    qm = qk; dpm = dpk;
    qk = qn; dpk = dpn;
    lower_prev = lower;
    if (k + 2 < km) {
      qn = Q(k + 2);
      dpn = DPK(k + 2);
    }
  }
  // qm / dpm hold the bottom level
  Q(km - 1) = qm;
  {
    const double dm = qm * dpm;
    sum0 += dm;
    sum1 += dm > 0.0 ? dm : 0.0;
  }
  // final_check (fillz.py:111-117)
  const double fac = sum0 > 0.0 ? sum0 / sum1 : 0.0;
  if (zfix > 0 && fac > 0.0) {
    for (int k = 1; k < km; ++k) {
      const double d = DPK(k);
      const double dm = Q(k) * d;
      const double v = fac * dm / d;
      Q(k) = v > 0.0 ? v : 0.0;
    }
  }
#undef Q
#undef DPK
}

int launch_fillz(const Geo& g, real* const* q, int nq, const real* dp, hipStream_t st) {
  if (nq < 1 || nq > REMAP_MAXQ) return PACE_ERR_ARG;
  if (g.nk < 4) return PACE_ERR_UNSUPPORTED;
  RemapBatch B{};
  B.n = nq;
  for (int f = 0; f < nq; ++f) {
    if (!q[f]) return PACE_ERR_ARG;
    B.q[f] = q[f];
  }
  RemapWin w{g.is, g.ie, g.js, g.je, g.nk};
  hipLaunchKernelGGL(k_fillz, dim3((g.n + 63) / 64, g.n, nq), dim3(64), 0, st, g, w, B, dp);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}

int64_t map_single_workspace_bytes(const Geo& g, int nq) {
  return (int64_t)g.sk * (g.nk + 1) * (int64_t)sizeof(real) * REMAP_NFIELDS * nq;
}

template <int KORD>
static int launch_coeffs(int iv, dim3 grid, hipStream_t st, const Geo& g, const RemapWin& w, real* ws, double qmin) {
  const dim3 block(64, 4);
#define RC(IV) hipLaunchKernelGGL((k_remap_coefficients<KORD, IV>), grid, block, 0, st, g, w, ws, qmin)
  switch (iv) {
    case 0: RC(0); break;
    case -1: RC(-1); break;
    case 2: RC(2); break;
    default: RC(1); break;  // iv < -1, iv == 1, iv > 2 share one variant (remap_profile.py:398-400)
  }
#undef RC
  return PACE_OK;
}

// nq fields that share pe1 / pe2, kord, iv, qs and qmin, in one three-launch sequence
int launch_map_fields(const Geo& g, void* ws_, real* const* q, int nq, const real* pe1, const real* pe2, const real* qs,
                      double qmin, int kord, int iv, int xstag, int ystag, hipStream_t st) {
  kord = kord < 0 ? -kord : kord;
  if (kord != 9 && kord != 10) return PACE_ERR_UNSUPPORTED;
  if (g.nk < 6) return PACE_ERR_UNSUPPORTED;
  if (nq < 1 || nq > REMAP_MAXQ) return PACE_ERR_ARG;
  if (iv == -2 && !qs) return PACE_ERR_ARG;
  real* ws = (real*)ws_;
  RemapBatch B{};
  B.n = nq;
  for (int f = 0; f < nq; ++f) {
    if (!q[f]) return PACE_ERR_ARG;
    B.q[f] = q[f];
  }
  RemapWin w{g.is, g.ie + (xstag ? 1 : 0), g.js, g.je + (ystag ? 1 : 0), g.nk};
  const int nx = w.i1 - w.i0 + 1, ny = w.j1 - w.j0 + 1;
  const dim3 cgrid((nx + 63) / 64, ny, nq);
  if (iv == -2) hipLaunchKernelGGL(k_remap_interfaces<true>, cgrid, dim3(64), 0, st, g, w, B, pe1, qs, ws);
  else hipLaunchKernelGGL(k_remap_interfaces<false>, cgrid, dim3(64), 0, st, g, w, B, pe1, qs, ws);
  const dim3 pgrid((nx + 63) / 64, (ny + 3) / 4, g.nk * nq);
  if (kord == 9) launch_coeffs<9>(iv, pgrid, st, g, w, ws, qmin);
  else launch_coeffs<10>(iv, pgrid, st, g, w, ws, qmin);
  const dim3 lgrid((nx + 63) / 64, ny, nq * ((g.nk + REMAP_KB - 1) / REMAP_KB));
  hipLaunchKernelGGL(k_remap_layers, lgrid, dim3(64), 0, st, g, w, B, pe1, pe2, ws);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}
