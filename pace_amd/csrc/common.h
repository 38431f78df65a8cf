// Shared device-side definitions for the FV3 acoustic-step kernels (gfx950 / CDNA4).
//
// Data layout (DESIGN.md "HBM layout"): every 3-D field is [k][j][i] with i fastest, row stride
// sj >= ni = N+7 doubles (padded to a multiple of 16 = 128 B so each 64-lane wave issues aligned,
// fully coalesced 512 B rows), level stride sk = sj*nj.  Index space is the reference's
// (dsl/pace/dsl/stencil.py:629-667): halo 3, compute domain is..ie = 3..N+2 in both directions,
// one tile per device so the tile edges coincide with is/ie/js/je.
#pragma once
#ifdef PACE_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>
#include <stdio.h>

#include "../../include/pace_hip.h"

struct Geo {
  int n, nk, ni, nj, sj;
  long sk;
  int is, ie, js, je;
};

static inline Geo make_geo(const pace_geom_t* g) {
  Geo o;
  o.n = g->n;
  o.nk = g->nk;
  o.ni = g->n + 7;
  o.nj = g->n + 7;
  o.sj = g->sj;
  o.sk = g->sk;
  o.is = 3;
  o.ie = g->n + 2;
  o.js = 3;
  o.je = g->n + 2;
  return o;
}

typedef pace_metrics_t Met;
// storage type of fields, metrics and device K-arrays (include/pace_hip.h); arithmetic is double in both builds
typedef pace_real_t real;
#define REAL_SHIFT (sizeof(real) == 8 ? 3 : 2)  // log2 of the element size, for the byte-offset addressing of some kernels

#define IDX2(g, i, j) ((long)(i) + (long)(j) * (g).sj)
#define IDX3(g, i, j, k) ((long)(i) + (long)(j) * (g).sj + (long)(k) * (g).sk)

// ppm.py:6-19
#define PPM_C1 (-2.0 / 14.0)
#define PPM_C2 (11.0 / 14.0)
#define PPM_C3 (5.0 / 14.0)
#define PPM_P1 (7.0 / 12.0)
#define PPM_P2 (-1.0 / 12.0)

// ---------------------------------------------------------------------------------------------
// Corner index maps (stencils/pace/stencils/corners.py:307-425; closed forms derived in
// oracle/corner_ops.py).  A read of an A-grid field "with corners copied in x (or y)" is a read
// of the raw field at the mapped index, so no kernel ever rewrites the corner halos.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void remap_agrid_x(const Geo& g, int& i, int& j) {
  const bool w = i < g.is, e = i > g.ie, s = j < g.js, n = j > g.je;
  if (!((w || e) && (s || n))) return;
  const int a = w ? g.is - 1 - i : i - g.ie - 1;  // cells away from the tile edge in i
  const int b = s ? g.js - 1 - j : j - g.je - 1;  // ... in j
  if (a > 2 || b > 2) return;                     // the storage has one spare row/column past the halo
  i = w ? g.is - 1 - b : g.ie + 1 + b;
  j = s ? g.js + a : g.je - a;
}

__device__ __forceinline__ void remap_agrid_y(const Geo& g, int& i, int& j) {
  const bool w = i < g.is, e = i > g.ie, s = j < g.js, n = j > g.je;
  if (!((w || e) && (s || n))) return;
  const int a = w ? g.is - 1 - i : i - g.ie - 1;
  const int b = s ? g.js - 1 - j : j - g.je - 1;
  if (a > 2 || b > 2) return;
  i = w ? g.is + b : g.ie - b;
  j = s ? g.js - 1 - a : g.je + 1 + a;
}

// ---------------------------------------------------------------------------------------------
// PPM interface value / flux for ord < 8 (xppm.py:19-72,148-181; yppm.py is the transpose).
// q[] holds the six cell values at positions pos-3 .. pos+2 along the sweep axis, d[] the matching
// A-grid spacings (only read next to a tile edge).  s/e are the tile start/end along the axis.
// ---------------------------------------------------------------------------------------------
// The one-sided forms next to a tile edge read four A-grid spacings along the sweep axis: positions s-2 .. s+1 for the forms
// at the start of the tile, e-1 .. e+2 at its end.  A "spacing provider" P(lo, idx, base) returns the idx-th of them (lo: the
// start set; base: the position of idx 0).  SpacingFn adapts a function of the position (a load per call -- under the
// position tests that is a branch and a wait per interface); EdgeSpacing holds both sets, loaded once per run before the
// arithmetic (k_fvtp2d.hip: measured 7 serialized L2 round trips per run in the tiles along an edge before).
template <class DX>
struct SpacingFn {
  DX f;
  __device__ __forceinline__ double operator()(bool, int idx, int base) const { return f(base + idx); }
};
struct EdgeSpacing {
  double S[4], E[4];
  __device__ __forceinline__ double operator()(bool lo, int idx, int) const { return lo ? S[idx] : E[idx]; }
};

template <class P>
__device__ __forceinline__ double ppm_al_p(const double* q, int off, int pos, int s, int e, const P& sp) {
  // q[off + m] is the cell value at position pos + m
  const double qm2 = q[off - 2], qm1 = q[off - 1], q0 = q[off], qp1 = q[off + 1];
  if (pos == s - 1 || pos == e) return PPM_C1 * qm2 + PPM_C2 * qm1 + PPM_C3 * q0;
  if (pos == s || pos == e + 1) {
    const bool lo = pos == s;
    const double dm2 = sp(lo, 0, pos - 2), dm1 = sp(lo, 1, pos - 2), d0 = sp(lo, 2, pos - 2), dp1 = sp(lo, 3, pos - 2);
    return 0.5 * (((2.0 * dm1 + dm2) * qm1 - dm1 * qm2) / (dm2 + dm1) + ((2.0 * d0 + dp1) * q0 - d0 * qp1) / (d0 + dp1));
  }
  if (pos == s + 1 || pos == e + 2) return PPM_C3 * qm1 + PPM_C2 * q0 + PPM_C1 * qp1;
  return PPM_P1 * (qm1 + q0) + PPM_P2 * (qm2 + qp1);
}
template <class DX>
__device__ __forceinline__ double ppm_al(const double* q, int off, int pos, int s, int e, DX dxa) {
  return ppm_al_p(q, off, pos, s, e, SpacingFn<DX>{dxa});
}

// Interior form: valid when no interface of the stencil lies within 2 of a tile edge (block-uniform test in the callers).
__device__ __forceinline__ double ppm_al_interior(const double* q, int off) {
  return PPM_P1 * (q[off - 1] + q[off]) + PPM_P2 * (q[off - 2] + q[off + 1]);
}

// Mean value advected through the interface at position pos (between cells pos-1 and pos).
// q6[m] = cell value at pos-3+m, m = 0..5.
template <int MORD, bool EDGE = true, class DX>
__device__ __forceinline__ double ppm_flux6(const double* q6, double c, int pos, int s, int e, DX dxa) {
  const double al_m = EDGE ? ppm_al(q6, 2, pos - 1, s, e, dxa) : ppm_al_interior(q6, 2);
  const double al_0 = EDGE ? ppm_al(q6, 3, pos, s, e, dxa) : ppm_al_interior(q6, 3);
  const double al_p = EDGE ? ppm_al(q6, 4, pos + 1, s, e, dxa) : ppm_al_interior(q6, 4);
  const double qm = q6[2], q0 = q6[3];
  const double bl_m = al_m - qm, br_m = al_0 - qm, b0_m = bl_m + br_m;
  const double bl_0 = al_0 - q0, br_0 = al_p - q0, b0_0 = bl_0 + br_0;
  bool s_m, s_0;
  if (MORD == 5) {
    s_m = bl_m * br_m < 0;
    s_0 = bl_0 * br_0 < 0;
  } else {
    s_m = (3.0 * fabs(b0_m)) < fabs(bl_m - br_m);
    s_0 = (3.0 * fabs(b0_0)) < fabs(bl_0 - br_0);
  }
  const double mask = (s_m || s_0) ? 1.0 : 0.0;
  if (c > 0.0) {
    const double fx1 = (1.0 - c) * (br_m - c * b0_m);
    return qm + fx1 * mask;
  } else {
    const double fx1 = (1.0 + c) * (bl_0 + c * b0_0);
    return q0 + fx1 * mask;
  }
}

// ---------------------------------------------------------------------------------------------
// Monotone PPM, ord 8 (xppm.py:76-145,185-287 with xt_minmax = True; ppm.py:22-35) for a run of F interfaces.
// The cell at position x needs q[x-2 .. x+2]; the six-cell window of an interface covers both of its cells.
// ---------------------------------------------------------------------------------------------
#define PPM_S11 (11.0 / 14.0)
#define PPM_S14 (4.0 / 7.0)
#define PPM_S15 (3.0 / 14.0)
__device__ __forceinline__ double sign_like(double a, double b) { return (b > 0.0) ? fabs(a) : -fabs(a); }  // basic_operations.sign

__device__ __forceinline__ double dm_ord8(double qm1, double q0, double qp1) {  // dm_iord8plus
  const double xt = 0.25 * (qp1 - qm1);
  const double dqr = fmax(fmax(q0, qm1), qp1) - q0;
  const double dql = q0 - fmin(fmin(q0, qm1), qp1);
  return sign_like(fmin(fmin(fabs(xt), dqr), dql), xt);
}

__device__ __forceinline__ void pert_ppm_standard_constraint(double& al, double& ar) {  // ppm.py:22-35
  if (al * ar < 0.0) {
    const double da1 = al - ar;
    const double da2 = da1 * da1;
    const double a6da = 3.0 * (al + ar) * da1;
    if (a6da < -da2) ar = -2.0 * al;
    else if (a6da > da2) al = -2.0 * ar;
  } else {
    al = 0.0;
    ar = 0.0;
  }
}

template <bool EDGE, int F, class P>
__device__ __forceinline__ void ppm_run8_p(const double* Q, const double* c, int pos0, int s, int e, const P& sp, double* out) {
  // Q[u] = cell pos0-3+u.  dm at u = 1 .. F+3, al (interface between u-1 and u) at u = 2 .. F+3, cells u = 2 .. F+2
  double dm[F + 5], al[F + 5];
#pragma unroll
  for (int u = 1; u <= F + 3; ++u) dm[u] = dm_ord8(Q[u - 1], Q[u], Q[u + 1]);
#pragma unroll
  for (int u = 2; u <= F + 3; ++u) al[u] = 0.5 * (Q[u - 1] + Q[u]) + 1.0 / 3.0 * (dm[u - 1] - dm[u]);
  double bl[F + 3], br[F + 3], b0[F + 3];
#pragma unroll
  for (int u = 2; u <= F + 2; ++u) {
    const double q0 = Q[u];
    const double xt2 = 2.0 * dm[u];
    double l = -1.0 * sign_like(fmin(fabs(xt2), fabs(al[u] - q0)), xt2);
    double r = sign_like(fmin(fabs(xt2), fabs(al[u + 1] - q0)), xt2);
    if (EDGE) {
      const int x = pos0 - 3 + u;
      const bool near = (x >= s - 1 && x <= s + 1) || (x >= e - 1 && x <= e + 1);
      if (near) {
        const double qm2 = Q[u - 2], qm1 = Q[u - 1], qp1 = Q[u + 1], qp2 = Q[u + 2];
        double xt_bl, xt_br;
        if (x == s - 1 || x == e) {
          const bool lo = x == s - 1;
          const double dm1 = sp(lo, 0, x - 1), d0 = sp(lo, 1, x - 1), dp1 = sp(lo, 2, x - 1), dp2 = sp(lo, 3, x - 1);
          double e0 = 0.5 * (((2.0 * d0 + dm1) * q0 - d0 * qm1) / (dm1 + d0) + ((2.0 * dp1 + dp2) * qp1 - dp1 * qp2) / (dp1 + dp2));
          e0 = fmin(fmax(e0, fmin(fmin(fmin(qm1, q0), qp1), qp2)), fmax(fmax(fmax(qm1, q0), qp1), qp2));
          xt_bl = (x == s - 1) ? PPM_S14 * dm[u - 1] + PPM_S11 * (qm1 - q0) + q0 : PPM_S15 * q0 + PPM_S11 * qm1 + PPM_S14 * dm[u - 1];
          xt_br = e0;
        } else if (x == s || x == e + 1) {
          const bool lo = x == s;
          const double dm2 = sp(lo, 0, x - 2), dm1 = sp(lo, 1, x - 2), d0 = sp(lo, 2, x - 2), dp1 = sp(lo, 3, x - 2);
          double e1 = 0.5 * (((2.0 * dm1 + dm2) * qm1 - dm1 * qm2) / (dm2 + dm1) + ((2.0 * d0 + dp1) * q0 - d0 * qp1) / (d0 + dp1));
          e1 = fmin(fmax(e1, fmin(fmin(fmin(qm2, qm1), q0), qp1)), fmax(fmax(fmax(qm2, qm1), q0), qp1));
          xt_bl = e1;
          xt_br = (x == s) ? PPM_S15 * q0 + PPM_S11 * qp1 - PPM_S14 * dm[u + 1] : PPM_S11 * (qp1 - q0) - PPM_S14 * dm[u + 1] + q0;
        } else if (x == s + 1) {
          xt_bl = PPM_S15 * qm1 + PPM_S11 * q0 - PPM_S14 * dm[u];
          xt_br = al[u + 1];
        } else {  // x == e - 1
          xt_bl = al[u];
          xt_br = PPM_S15 * qp1 + PPM_S11 * q0 + PPM_S14 * dm[u];
        }
        l = xt_bl - q0;
        r = xt_br - q0;
        pert_ppm_standard_constraint(l, r);
      }
    }
    bl[u] = l;
    br[u] = r;
    b0[u] = l + r;
  }
#pragma unroll
  for (int f = 0; f < F; ++f) {
    const double cc = c[f];
    if (cc > 0.0) {
      const double fx1 = (1.0 - cc) * (br[f + 2] - cc * b0[f + 2]);
      out[f] = Q[f + 2] + fx1 * 1.0;
    } else {
      const double fx1 = (1.0 + cc) * (bl[f + 3] + cc * b0[f + 3]);
      out[f] = Q[f + 3] + fx1 * 1.0;
    }
  }
}

template <bool EDGE, int F, class DX>
__device__ __forceinline__ void ppm_run8(const double* Q, const double* c, int pos0, int s, int e, DX dxa, double* out) {
  ppm_run8_p<EDGE, F>(Q, c, pos0, s, e, SpacingFn<DX>{dxa}, out);
}

// A run of F consecutive interfaces pos0 .. pos0+F-1 evaluated by one thread: the F+2 interface values and the F+1
// cell reconstructions (bl, br, b0, steepness flag) are computed once and shared, instead of three interface values
// and two reconstructions per flux.  Q[u] = cell pos0-3+u, u = 0 .. F+4.  Same expressions as ppm_flux6 -> same bits.
// the fluxes of a run from its F + 2 interface values (shared by the forms of ppm_run below)
template <int MORD, int F>
__device__ __forceinline__ void ppm_fluxes_from_al(const double* Q, const double* al, const double* c, double* out) {
  // Cell x of the run (Q[x + 2], between al[x] and al[x + 1]): bl = al[x] - q, br = al[x + 1] - q, b0 = bl + br, and whether the
  // parabola is steep (xppm.py:40-53).  The fluxes are formed in the order of the faces with only the two cells of a face live
  // (round 5: all F + 1 reconstructions at once were 6 * (3 doubles + a flag) of register pressure at the kernel's peak).
  auto cell = [&](int x, double& b0, bool& steep) {
    const double qx = Q[x + 2];
    const double bl = al[x] - qx;
    const double br = al[x + 1] - qx;
    b0 = bl + br;
    steep = (MORD == 5) ? (bl * br < 0) : ((3.0 * fabs(b0)) < fabs(bl - br));
  };
  double b0_lo, b0_hi;
  bool steep_lo, steep_hi;
  cell(0, b0_lo, steep_lo);
#pragma unroll
  for (int f = 0; f < F; ++f) {
    cell(f + 1, b0_hi, steep_hi);
    const double mask = (steep_lo || steep_hi) ? 1.0 : 0.0;
    const double cc = c[f];
    // xppm.py:56-72: c > 0: q[i-1] + (1 - c) * (br[i-1] - c * b0[i-1]), else q[i] + (1 + c) * (bl[i] + c * b0[i]).  With a = |c|
    // both are Q + (1 - a) * (X - a * B) on the upwind cell's values -- the same bits (1 + c == 1 - |c| and bl + c * b0 ==
    // bl - |c| * b0 for c <= 0: a sign moved, no rounding) -- selected as values instead of as branches: a divergent branch per
    // face costs seven scalar instructions of exec-mask bookkeeping and both arms.
    // (X = br of the cell behind the face or bl of the cell in front of it: both are al[f + 1] - that cell's value, the very
    // expressions of the cell function above -- one select less)
    const bool up = cc > 0.0;
    const double a = fabs(cc);
    const double Qs = up ? Q[f + 2] : Q[f + 3];
    const double X = al[f + 1] - Qs;
    const double B = up ? b0_lo : b0_hi;
    const double fx1 = (1.0 - a) * (X - a * B);
    out[f] = Qs + fx1 * mask;
    b0_lo = b0_hi;
    steep_lo = steep_hi;
  }
}

template <int MORD, bool EDGE, int F, class P>
__device__ __forceinline__ void ppm_run_p(const double* Q, const double* c, int pos0, int s, int e, const P& sp, double* out) {
  if (MORD == 8) {
    ppm_run8_p<EDGE, F>(Q, c, pos0, s, e, sp, out);
    return;
  }
  double al[F + 2];
#pragma unroll
  for (int a = 0; a < F + 2; ++a) al[a] = EDGE ? ppm_al_p(Q, a + 2, pos0 - 1 + a, s, e, sp) : ppm_al_interior(Q, a + 2);
  ppm_fluxes_from_al<MORD, F>(Q, al, c, out);
}

// The same for a run whose one-sided interface values sit at positions known at compile time: the run's interface values
// number A, A + 1, A + 2 lie at s-1, s, s+1 (lo: the start of the tile) or e, e+1, e+2 (its end).  ppm_al_p finds them by
// comparing every interface position with s and e -- three branches per interface value, and since each wave of an x-sweep
// holds a lane next to the edge, every wave of an edge tile walked through all of that (measured: 6.4 k instead of 3.4 k
// cycles per sweep).  Here: interior form everywhere, then the three values of the edge lane are overwritten.  Same
// expressions as ppm_al_p -> same bits.
template <int A, class P>
__device__ __forceinline__ void ppm_patch_edge(double* al, const double* Q, bool lo, const P& sp) {
  al[A] = PPM_C1 * Q[A] + PPM_C2 * Q[A + 1] + PPM_C3 * Q[A + 2];
  {
    const double qm2 = Q[A + 1], qm1 = Q[A + 2], q0 = Q[A + 3], qp1 = Q[A + 4];
    const double dm2 = sp(lo, 0, 0), dm1 = sp(lo, 1, 0), d0 = sp(lo, 2, 0), dp1 = sp(lo, 3, 0);
    al[A + 1] = 0.5 * (((2.0 * dm1 + dm2) * qm1 - dm1 * qm2) / (dm2 + dm1) + ((2.0 * d0 + dp1) * q0 - d0 * qp1) / (d0 + dp1));
  }
  al[A + 2] = PPM_C3 * Q[A + 3] + PPM_C2 * Q[A + 4] + PPM_C1 * Q[A + 5];
}
// lane_lo: this run starts at the first interface of the tile (A = 0); lane_hi: it holds the end of the tile at A = AHI
template <int MORD, int F, int AHI, class SP = EdgeSpacing>
__device__ __forceinline__ void ppm_run_canon(const double* Q, const double* c, bool lane_lo, bool lane_hi, const SP& sp,
                                              double* out) {
  static_assert(MORD != 8 && AHI >= 0 && AHI <= F - 1, "ppm_run_canon");
  double al[F + 2];
#pragma unroll
  for (int a = 0; a < F + 2; ++a) al[a] = ppm_al_interior(Q, a + 2);
  if (lane_lo) ppm_patch_edge<0>(al, Q, true, sp);
  if (lane_hi) ppm_patch_edge<AHI>(al, Q, false, sp);
  ppm_fluxes_from_al<MORD, F>(Q, al, c, out);
}
template <int MORD, bool EDGE, int F, class DX>
__device__ __forceinline__ void ppm_run(const double* Q, const double* c, int pos0, int s, int e, DX dxa, double* out) {
  ppm_run_p<MORD, EDGE, F>(Q, c, pos0, s, e, SpacingFn<DX>{dxa}, out);
}

// Region launches.  A stencil with edge / corner logic is split into an INTERIOR box (straight-line code, ~95 % of
// the points; 64 x 4 patches so rows stay coalesced) and up to four thin frame strips that run the general code (their
// points are flattened over the 256 threads of a block so lanes stay busy whatever the strip's orientation) -- all in
// ONE launch: blockIdx.x is split into per-region ranges, the branch on the region is block-uniform.
#define MAX_REGIONS 9
struct Regions {
  int n;
  int ib[MAX_REGIONS], ie[MAX_REGIONS], jb[MAX_REGIONS], je[MAX_REGIONS];
  int first[MAX_REGIONS + 1];  // first block of each region; first[n] = total
  int nbx0;                    // 64-wide patches per row of region 0
  int nplain;                  // regions [0, nplain) are "interior" (their points take a kernel's plain forms); set by add_region: 1
};
static inline void add_region(Regions& r, int ib, int ie, int jb, int je) {
  if (ie < ib || je < jb) return;
  const int q = r.n++;
  r.ib[q] = ib; r.ie[q] = ie; r.jb[q] = jb; r.je[q] = je;
  int nb;
  if (q == 0) {
    r.nplain = 1;
    r.nbx0 = (ie - ib + 64) / 64;
    nb = r.nbx0 * ((je - jb + 4) / 4);
  } else {
    nb = ((ie - ib + 1) * (je - jb + 1) + 255) / 256;
  }
  r.first[q + 1] = r.first[q] + nb;
}
// region 0 = interior box [is+di, ie+1-di] x [js+dj, je+1-dj] of the B-grid domain is..ie+1, the rest = frame strips
static inline Regions bgrid_regions(const Geo& g, int d) {
  Regions r{};
  add_region(r, g.is + d, g.ie + 1 - d, g.js + d, g.je + 1 - d);
  add_region(r, g.is, g.is + d - 1, g.js, g.je + 1);
  add_region(r, g.ie + 2 - d, g.ie + 1, g.js, g.je + 1);
  add_region(r, g.is + d, g.ie + 1 - d, g.js, g.js + d - 1);
  add_region(r, g.is + d, g.ie + 1 - d, g.je + 2 - d, g.je + 1);
  return r;
}
// sets i, j, k and `interior`; returns from the kernel for padding threads
#define REGION_POINT(R)                                                                        \
  int reg__ = 0;                                                                               \
  while (reg__ + 1 < (R).n && (int)blockIdx.x >= (R).first[reg__ + 1]) ++reg__;               \
  const int b__ = (int)blockIdx.x - (R).first[reg__];                                          \
  const int t__ = (int)threadIdx.y * 64 + (int)threadIdx.x;                                    \
  const bool interior = reg__ < (R).nplain;                                                         \
  int i, j;                                                                                    \
  const int k = (int)blockIdx.z;                                                               \
  if (reg__ == 0) {                                                                            \
    i = (R).ib[0] + (b__ % (R).nbx0) * 64 + (int)threadIdx.x;                                   \
    j = (R).jb[0] + (b__ / (R).nbx0) * 4 + (int)threadIdx.y;                                    \
    if (i > (R).ie[0] || j > (R).je[0]) return;                                                \
  } else {                                                                                     \
    const int w__ = (R).ie[reg__] - (R).ib[reg__] + 1;                                         \
    const int p__ = b__ * 256 + t__;                                                           \
    j = (R).jb[reg__] + p__ / w__;                                                             \
    i = (R).ib[reg__] + p__ % w__;                                                             \
    if (j > (R).je[reg__]) return;                                                             \
  }
static inline dim3 regions_grid(const Regions& r, int nlev) { return dim3((unsigned)r.first[r.n], 1, (unsigned)nlev); }
// The same with CH levels per thread in the interior patches and ONE in the frame strips (their edge forms are several times the
// interior's cost per point: CH levels of them in a row would make the frame workgroups the last to finish), in one launch so
// that the strips run beside the interior: a 1-D grid, first the interior patches of every chunk of levels, then the frame
// blocks of every level (the other order measured slower: 99 against 91 us for k_d2a2c_b).  Sets i, j, `interior`, k0 (first level) and nk_here (levels this thread takes).
static inline dim3 regions_grid_chunked(const Regions& r, int nlev, int ch) {
  const int nbi = r.first[r.nplain > 0 ? r.nplain : 1], nfr = r.first[r.n] - nbi;
  return dim3((unsigned)(nbi * ((nlev + ch - 1) / ch) + nfr * nlev), 1, 1);
}
#define REGION_POINT_CHUNKED(R, CH, NLEV)                                                      \
  const int nbi__ = (R).first[(R).nplain > 0 ? (R).nplain : 1];                                \
  const int ni__ = nbi__ * (((NLEV) + (CH)-1) / (CH));                                         \
  int bx__, k0, nk_here;                                                                       \
  if ((int)blockIdx.x < ni__) {                                                                \
    const int ch__ = (int)blockIdx.x / nbi__;                                                  \
    bx__ = (int)blockIdx.x - ch__ * nbi__;                                                     \
    k0 = ch__ * (CH);                                                                          \
    nk_here = (NLEV)-k0 < (CH) ? (NLEV)-k0 : (CH);                                             \
  } else {                                                                                     \
    const int f__ = (int)blockIdx.x - ni__, nfr__ = (R).first[(R).n] - nbi__;                  \
    k0 = f__ / nfr__;                                                                          \
    bx__ = nbi__ + (f__ - k0 * nfr__);                                                         \
    nk_here = 1;                                                                               \
  }                                                                                            \
  int reg__ = 0;                                                                               \
  while (reg__ + 1 < (R).n && bx__ >= (R).first[reg__ + 1]) ++reg__;                           \
  const int b__ = bx__ - (R).first[reg__];                                                     \
  const int t__ = (int)threadIdx.y * 64 + (int)threadIdx.x;                                    \
  const bool interior = reg__ < (R).nplain;                                                         \
  int i, j;                                                                                    \
  if (reg__ == 0) {                                                                            \
    i = (R).ib[0] + (b__ % (R).nbx0) * 64 + (int)threadIdx.x;                                   \
    j = (R).jb[0] + (b__ / (R).nbx0) * 4 + (int)threadIdx.y;                                    \
    if (i > (R).ie[0] || j > (R).je[0]) return;                                                \
  } else {                                                                                     \
    const int w__ = (R).ie[reg__] - (R).ib[reg__] + 1;                                         \
    const int p__ = b__ * 256 + t__;                                                           \
    j = (R).jb[reg__] + p__ / w__;                                                             \
    i = (R).ib[reg__] + p__ % w__;                                                             \
    if (j > (R).je[reg__]) return;                                                             \
  }
// The same with the workgroups of a level on ONE XCD (workgroups are dealt to the eight XCDs round-robin in launch order; each
// XCD has its own L2): for kernels whose points re-read rows of their j-neighbours -- with the plain order the patch above and
// the patch below run on other XCDs and every XCD fetches its own copy of the shared rows.  XCD x works through levels x, x + 8,
// ... (affinity only; the map is a bijection of the launch's workgroups).
#ifdef PACE_EMU
#define REGION_POINT_XCD(R) REGION_POINT(R)
#else
#define REGION_POINT_XCD(R)                                                                    \
  int bx__, bz__;                                                                              \
  {                                                                                            \
    const int nbx__ = (int)gridDim.x, nlev__ = (int)gridDim.z;                                 \
    const int lin__ = (int)blockIdx.x + nbx__ * (int)blockIdx.z;                               \
    const int full__ = (nlev__ / 8) * 8;                                                       \
    if (lin__ < full__ * nbx__) {                                                              \
      const int slot__ = lin__ >> 3;                                                           \
      bz__ = (slot__ / nbx__) * 8 + (lin__ & 7);                                               \
      bx__ = slot__ - (slot__ / nbx__) * nbx__;                                                \
    } else {                                                                                   \
      bz__ = lin__ / nbx__;                                                                    \
      bx__ = lin__ - bz__ * nbx__;                                                             \
    }                                                                                          \
  }                                                                                            \
  int reg__ = 0;                                                                               \
  while (reg__ + 1 < (R).n && bx__ >= (R).first[reg__ + 1]) ++reg__;                           \
  const int b__ = bx__ - (R).first[reg__];                                                     \
  const int t__ = (int)threadIdx.y * 64 + (int)threadIdx.x;                                    \
  const bool interior = reg__ < (R).nplain;                                                         \
  int i, j;                                                                                    \
  const int k = bz__;                                                                          \
  if (reg__ == 0) {                                                                            \
    i = (R).ib[0] + (b__ % (R).nbx0) * 64 + (int)threadIdx.x;                                   \
    j = (R).jb[0] + (b__ / (R).nbx0) * 4 + (int)threadIdx.y;                                    \
    if (i > (R).ie[0] || j > (R).je[0]) return;                                                \
  } else {                                                                                     \
    const int w__ = (R).ie[reg__] - (R).ib[reg__] + 1;                                         \
    const int p__ = b__ * 256 + t__;                                                           \
    j = (R).jb[reg__] + p__ / w__;                                                             \
    i = (R).ib[reg__] + p__ % w__;                                                             \
    if (j > (R).je[reg__]) return;                                                             \
  }
#endif

// Last HIP error text seen by this library on the calling thread (pace_last_error()).
extern thread_local char g_pace_err[256];
void pace_set_err(const char* where, hipError_t e);

// PACE_SYNC_LAUNCHES=1 in the environment: name every launch site on stderr and wait for the device there (debugging aid:
// an asynchronous device fault is then reported right after the launch that caused it).
extern int g_pace_sync_launches;
#define PACE_CHECK_LAUNCH()                                        \
  do {                                                             \
    hipError_t err__ = hipGetLastError();                          \
    if (err__ != hipSuccess) {                                     \
      pace_set_err(__func__, err__);                               \
      return PACE_ERR_LAUNCH;                                      \
    }                                                              \
    if (g_pace_sync_launches) {                                    \
      fprintf(stderr, "[pace] %s:%d\n", __func__, __LINE__);       \
      fflush(stderr);                                              \
      (void)hipDeviceSynchronize();                                \
    }                                                              \
  } while (0)

#ifndef PACE_EMU
// A stream of the calling thread's own (high priority) with the two events of a fork / join around it, for launchers that put a
// short chain of latency-bound kernels (edge forms: few points, dependent reads) BESIDE a streaming or tile kernel on the caller's
// stream instead of in front of it.  Created on first use; nullptr if the runtime refuses.
struct PaceSideStream {
  hipStream_t s;
  hipEvent_t in, out;
  // side waits for everything queued on `main` so far
  bool fork(hipStream_t main) const { return hipEventRecord(in, main) == hipSuccess && hipStreamWaitEvent(s, in, 0) == hipSuccess; }
  // `main` waits for everything queued on the side stream so far
  bool join(hipStream_t main) const { return hipEventRecord(out, s) == hipSuccess && hipStreamWaitEvent(main, out, 0) == hipSuccess; }
};
static inline const PaceSideStream* pace_side_stream() {
  static thread_local PaceSideStream side{nullptr, nullptr, nullptr};
  static thread_local bool failed = false;
  if (side.s == nullptr && !failed) {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&side.s, hipStreamNonBlocking, hi) != hipSuccess ||
        hipEventCreateWithFlags(&side.in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&side.out, hipEventDisableTiming) != hipSuccess) {
      failed = true;
      side.s = nullptr;
    }
  }
  return side.s != nullptr ? &side : nullptr;
}
#endif

// Interior / frame split of a plane launch around a halo exchange: mode 0 = every point, 1 = the points inside the box,
// 2 = the points outside it (launched after mode 1; together they cover the plane exactly once).
struct SplitBox {
  int i0, i1, j0, j1;
  int mode;
  __device__ __forceinline__ bool skip(int i, int j) const {
    if (mode == 0) return false;
    const bool in = i >= i0 && i <= i1 && j >= j0 && j <= j1;
    return mode == 1 ? !in : in;
  }
};

static inline dim3 plane_grid(const Geo& g, int nlev) {
  return dim3((unsigned)(((long)g.sj * g.nj + 255) / 256), (unsigned)nlev, 1);
}
// 2-D patches for point kernels that read a (j+1) or (j-1) neighbour: a 64 x 4 patch per workgroup serves three of four
// such reads from the same workgroup's cache lines; with the flattened map below every row is fetched twice (the
// neighbouring row belongs to a workgroup on another XCD).  Launch with dim3(64, 4) threads.
#ifndef PATCH_W
#define PATCH_W 64
#endif
#define PATCH_H (256 / PATCH_W)
#define PATCH_BLOCK dim3(PATCH_W, PATCH_H)
static inline dim3 patch_grid(const Geo& g, int nlev) {
  return dim3((unsigned)((g.ni + PATCH_W - 1) / PATCH_W), (unsigned)((g.nj + PATCH_H - 1) / PATCH_H), (unsigned)nlev);
}
#define PATCH_IJK(g)                                              \
  const int i = (int)blockIdx.x * PATCH_W + (int)threadIdx.x;     \
  const int j = (int)blockIdx.y * PATCH_H + (int)threadIdx.y;     \
  const int k = (int)blockIdx.z;                                  \
  if (j >= (g).nj || i >= (g).ni) return;
// flattened plane index -> (i, j); returns false for pad lanes / out of plane
#define PLANE_IJK(g)                                                  \
  const long p__ = (long)blockIdx.x * 256 + threadIdx.x;             \
  const int j = (int)(p__ / (g).sj);                                  \
  const int i = (int)(p__ - (long)j * (g).sj);                        \
  const int k = (int)blockIdx.y;                                      \
  if (j >= (g).nj || i >= (g).ni) return;

// ------------------------------------------------------------------------------------------------
// advect_u_along_x / advect_v_along_y (xtp_u.py:9-91, ytp_v.py:9-91), ord < 8: the value advected through the face at `pos`
// from the six cells q6 = q(pos-3 .. pos+2); zero_m / zero_0: the reconstruction of cell pos-1 / pos is zeroed (xtp_u.py:41-49)
template <int MORD, bool EDGE = true, class DX>
__device__ __forceinline__ double wind_flux6(const double* q6, double csign, double cfl, int pos, int s, int e,
                                             DX dxa, bool zero_m, bool zero_0) {
  const double al_m = EDGE ? ppm_al(q6, 2, pos - 1, s, e, dxa) : ppm_al_interior(q6, 2);
  const double al_0 = EDGE ? ppm_al(q6, 3, pos, s, e, dxa) : ppm_al_interior(q6, 3);
  const double al_p = EDGE ? ppm_al(q6, 4, pos + 1, s, e, dxa) : ppm_al_interior(q6, 4);
  const double qm = q6[2], q0 = q6[3];
  double bl_m = al_m - qm, br_m = al_0 - qm;
  double bl_0 = al_0 - q0, br_0 = al_p - q0;
  if (zero_m) { bl_m = 0.0; br_m = 0.0; }
  if (zero_0) { bl_0 = 0.0; br_0 = 0.0; }
  const double b0_m = bl_m + br_m, b0_0 = bl_0 + br_0;
  bool s_m, s_0;
  if (MORD == 5) {
    s_m = bl_m * br_m < 0;
    s_0 = bl_0 * br_0 < 0;
  } else {
    s_m = (3.0 * fabs(b0_m)) < fabs(bl_m - br_m);
    s_0 = (3.0 * fabs(b0_0)) < fabs(bl_0 - br_0);
  }
  const double mask = (s_m || s_0) ? 1.0 : 0.0;
  const double fx0 = (cfl > 0.0) ? (1.0 - cfl) * (br_m - cfl * b0_m) : (1.0 + cfl) * (bl_0 + cfl * b0_0);
  return (csign > 0.0) ? (qm + fx0 * mask) : (q0 + fx0 * mask);
}

