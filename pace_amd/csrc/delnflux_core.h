// Tile-local core of DelnFluxNoSG / DelnFlux (delnflux.py:1209-1261): shared by the stand-alone kernel
// (k_delnflux.hip) and the fused transport kernel (k_fvtp2d.hip).  See k_delnflux.hip for the design notes.
#pragma once
#include <type_traits>

#include "common.h"

#ifndef DN_TI
#define DN_TI 32
#define DN_TJ 24
#endif
#define DW (DN_TI + 6)
#define DH (DN_TJ + 6)
#define DWP (DW + 1)
#define DN_NE ((DW * DH + 255) / 256)

// On entry `src` (pitch DWP) holds q on the footprint [i0-3, i0+TI+3) x [j0-3, j0+TJ+3), zero outside the storage.
// On exit (after a barrier) sfx / sfy hold the damping fluxes; the x-flux of face (i0+ii, j0+jj) is
// sfx[(jj+3)*DWP + ii+3].  Each thread owns DN_NE fixed points of the footprint: their LDS slot, validity flags and the
// three metric values are worked out once, so an iteration is LDS reads and a handful of flops.
__device__ __forceinline__ void delnflux_core(const Geo& g, const Met& m, const double* src, double* sd, double* sfx,
                                              double* sfy, int i0, int j0, double d0, bool hi_order, int nmax) {
  const int tid = threadIdx.x;
  const int ilo = i0 - 3, jlo = j0 - 3;
  // the corner-copy index maps only matter to workgroups whose footprint reaches a corner of the halo (block-uniform)
  const bool rc = hi_order && (ilo < g.is || ilo + DW - 1 > g.ie) && (jlo < g.js || jlo + DH - 1 > g.je);
  const int iters = hi_order ? nmax : 0;
  int lidx[DN_NE], pgi[DN_NE], pgj[DN_NE];
  bool own[DN_NE], flx[DN_NE], cel[DN_NE];
  double dv[DN_NE], du[DN_NE], ra[DN_NE];
#pragma unroll
  for (int t = 0; t < DN_NE; ++t) {
    const int e = tid + 256 * t;
    const int jj = e / DW, ii = e - jj * DW;
    const int gi = ilo + ii, gj = jlo + jj;
    own[t] = e < DW * DH;
    const bool stored = own[t] && gi >= 0 && gi < g.ni && gj >= 0 && gj < g.nj;
    flx[t] = stored && gi >= 1 && gj >= 1 && ii >= 1 && jj >= 1;
    cel[t] = stored && ii < DW - 1 && jj < DH - 1;
    lidx[t] = jj * DWP + ii;
    pgi[t] = gi;
    pgj[t] = gj;
    // byte offset: uniform base + 32-bit lane offset.  The loads are unconditional (points outside the storage read the
    // first stored cell and discard it) so that all of them are in flight together instead of one branch + wait per point.
    // (Batching the LDS reads of the rounds below the same way was measured: -8 % in this stage, but 17-23 spilled VGPRs
    // at the 128-register budget of four waves per SIMD made the step 6 % slower.)
    const unsigned c2 = stored ? (unsigned)(__mul24(gj, g.sj * (int)sizeof(real)) + (gi << REAL_SHIFT)) : 0u;
    const double dv_raw = *(const real*)((const char*)m.del6_v + c2);
    const double du_raw = *(const real*)((const char*)m.del6_u + c2);
    const double ra_raw = *(const real*)((const char*)m.rarea + c2);
    dv[t] = flx[t] ? dv_raw : 0.0;
    du[t] = flx[t] ? du_raw : 0.0;
    ra[t] = cel[t] ? ra_raw : 0.0;
    if (own[t]) sd[lidx[t]] = stored ? d0 * src[lidx[t]] : 0.0;
  }
  __syncthreads();

  for (int it = 0;; ++it) {
#pragma unroll
    for (int t = 0; t < DN_NE; ++t) {
      if (!own[t]) continue;
      const int l = lidx[t];
      double vx = 0.0, vy = 0.0;
      if (flx[t]) {
        if (rc) {
          const int gi = pgi[t], gj = pgj[t];
          {
            int ai = gi - 1, aj = gj, bi = gi, bj = gj;
            remap_agrid_x(g, ai, aj);
            remap_agrid_x(g, bi, bj);
            const int la = ai - ilo, lb = aj - jlo, lc = bi - ilo, ld = bj - jlo;
            double da = 0.0, db = 0.0;
            if (la >= 0 && la < DW && lb >= 0 && lb < DH) da = sd[lb * DWP + la];
            if (lc >= 0 && lc < DW && ld >= 0 && ld < DH) db = sd[ld * DWP + lc];
            const double tt = dv[t] * (da - db);
            vx = (it == 0) ? tt : -tt;
          }
          {
            int ai = gi, aj = gj - 1, bi = gi, bj = gj;
            remap_agrid_y(g, ai, aj);
            remap_agrid_y(g, bi, bj);
            const int la = ai - ilo, lb = aj - jlo, lc = bi - ilo, ld = bj - jlo;
            double da = 0.0, db = 0.0;
            if (la >= 0 && la < DW && lb >= 0 && lb < DH) da = sd[lb * DWP + la];
            if (lc >= 0 && lc < DW && ld >= 0 && ld < DH) db = sd[ld * DWP + lc];
            const double tt = du[t] * (da - db);
            vy = (it == 0) ? tt : -tt;
          }
        } else {
          const double d0v = sd[l];
          const double tx = dv[t] * (sd[l - 1] - d0v);
          const double ty = du[t] * (sd[l - DWP] - d0v);
          vx = (it == 0) ? tx : -tx;
          vy = (it == 0) ? ty : -ty;
        }
      }
      sfx[l] = vx;
      sfy[l] = vy;
    }
    __syncthreads();
    if (it == iters) break;
    // d2_highorder (delnflux.py:183-205)
#pragma unroll
    for (int t = 0; t < DN_NE; ++t) {
      if (!own[t]) continue;
      const int l = lidx[t];
      double v = 0.0;
      if (cel[t]) v = (sfx[l] - sfx[l + 1] + sfy[l] - sfy[l + DWP]) * ra[t];
      sd[l] = v;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same iteration with one batch of LDS reads per pass (round 3).  A pass of the reference is flux -> divergence
// (delnflux.py:183-205 between two flux stencils); above, the two halves are separate LDS rounds of five points per thread,
// each point waiting for its own reads: measured 17.5 k of the transport workgroup's 50 k cycles, 11.9 k with the CU to
// itself -- pure dependent-chain latency.  Here a thread owns a COLUMN RUN of DN_RC cells and evaluates the four face
// fluxes of each of its cells itself (the flux of a face is the same expression on the same operands from either side, so
// the bits are those of the two-step form).
// Register budget (the transport kernel lives at 128 VGPRs; a first version with the metrics of the run in registers
// spilled and was 2.3 x slower): del6_v and del6_u sit in LDS tiles staged together with the footprint of q, only rarea of
// the run is held in registers; the iterate has ONE plane -- the first pass reads q and writes the plane, a later pass reads
// the plane into registers, waits for everyone (barrier) and overwrites it.  The last flux evaluation happens in the
// consumer's registers (deln_face_x / deln_face_y).  Cells whose five-point stencil reaches into a corner region of the
// halo -- where the reference reads corner copies, copy_corners_x_nord / _y_nord, delnflux.py:1009-1047 -- are left out of the
// run code and recomputed by up to 64 threads with the general (index-mapped) form: 15 cells per corner of the tile domain.
// ---------------------------------------------------------------------------------------------------------------------
#define DN_RC 5
#define DN_NR ((DH + DN_RC - 1) / DN_RC)
static_assert(DW * DN_NR <= 256, "one column run per thread");

struct DelnMet {
  double ra[DN_RC];
  bool valid[DN_RC];
  bool active;
  int c, r0;
};

__device__ __forceinline__ bool deln_in_corner(const Geo& g, int i, int j) {
  return (i < g.is || i > g.ie) && (j < g.js || j > g.je);
}
__device__ __forceinline__ bool deln_affected(const Geo& g, int i, int j) {
  return deln_in_corner(g, i - 1, j) || deln_in_corner(g, i, j) || deln_in_corner(g, i + 1, j) || deln_in_corner(g, i, j - 1) ||
         deln_in_corner(g, i, j + 1);
}
__device__ __forceinline__ unsigned deln_off(const Geo& g, int i, int j) {  // byte offset in a 2-D metric, clamped into the storage
  const bool stored = i >= 0 && i < g.ni && j >= 0 && j < g.nj;
  return stored ? (unsigned)(__mul24(j, g.sj * (int)sizeof(real)) + (i << REAL_SHIFT)) : 0u;
}

// del6_v, del6_u on the footprint -> LDS tiles sdv, sdu (pitch DWP; zero outside the storage).  The caller's barrier after
// staging q covers them.  All 256 threads.
template <class MT>
__device__ __forceinline__ void deln_stage_metrics(const Geo& g, const MT& m, double* sdv, double* sdu, int i0, int j0) {
  const int tid = threadIdx.x;
  double v[DN_NE], u[DN_NE];
#pragma unroll
  for (int t = 0; t < DN_NE; ++t) {
    const int e = tid + 256 * t;
    const int jj = e / DW, ii = e - jj * DW;
    const int gi = i0 - 3 + ii, gj = j0 - 3 + jj;
    const bool stored = e < DW * DH && gi >= 0 && gi < g.ni && gj >= 0 && gj < g.nj;
    const unsigned o = deln_off(g, stored ? gi : 0, stored ? gj : 0);
    v[t] = *(const real*)((const char*)m.del6_v + o);
    u[t] = *(const real*)((const char*)m.del6_u + o);
    if (!stored) v[t] = 0.0, u[t] = 0.0;
  }
#pragma unroll
  for (int t = 0; t < DN_NE; ++t) {
    const int e = tid + 256 * t;
    const int jj = e / DW, ii = e - jj * DW;
    if (e < DW * DH) sdv[jj * DWP + ii] = v[t], sdu[jj * DWP + ii] = u[t];
  }
}

// rarea of this thread's run (global loads: issue early, the first pass consumes them) and the validity of its cells
template <class MT>
__device__ __forceinline__ void deln_load(const Geo& g, const MT& m, int i0, int j0, DelnMet& M) {
  const int tid = threadIdx.x;
  const int ilo = i0 - 3, jlo = j0 - 3;
  M.active = tid < DW * DN_NR;
  const int r = tid / DW;
  M.c = tid - r * DW;
  M.r0 = r * DN_RC;
  const int gi = ilo + M.c;
#pragma unroll
  for (int t = 0; t < DN_RC; ++t) {
    const int jj = M.r0 + t, gj = jlo + jj;
    M.ra[t] = *(const real*)((const char*)m.rarea + deln_off(g, gi, gj));
    M.valid[t] = M.active && M.c >= 1 && M.c <= DW - 2 && jj >= 1 && jj <= DH - 2 && gi >= 1 && gi + 1 < g.ni && gj >= 1 && gj + 1 < g.nj;
  }
}

// the run part of a pass: res[t] = divergence at cell (c, r0 + t) of the fluxes of src (zero where the cell is not valid)
template <bool FIRST>
__device__ __forceinline__ void deln_run(const DelnMet& M, const double* src, const double* __restrict__ sdv,
                                         const double* __restrict__ sdu, double d0, double* res) {
  auto val = [&](double x) { return FIRST ? d0 * x : x; };  // (d0 * x: the reference's first statement, d2 = damp * q)
  auto sgn = [&](double x) { return FIRST ? x : -x; };
  const int c = M.c, r0 = M.r0;
  constexpr bool kExact = DH % DN_RC == 0;  // every run has DN_RC rows inside the footprint
  // two chunks of the run (cells [0, DN_H) and [DN_H, DN_RC)), each one batch of LDS reads
  const int b0 = r0 * DWP + c;
  const double* p = src + b0;
  const double* pv = sdv + b0;
  const double* pu = sdu + b0;
  const int ow = c > 0 ? -1 : 0, oe = c < DW - 1 ? 1 : 0;
  // offset of row r0 + t relative to row r0, clamped into the footprint (t is a constant after unrolling: only the rows
  // above / below the run are clamped at run time when kExact)
  auto ro = [&](int t) {
    if (kExact && t >= 0 && t < DN_RC) return t * DWP;
    int jj = r0 + t;
    jj = jj < 0 ? 0 : (jj > DH - 1 ? DH - 1 : jj);
    return (jj - r0) * DWP;
  };
  auto chunk = [&](auto T0_, auto T1_) {
    constexpr int T0 = decltype(T0_)::value, T1 = decltype(T1_)::value, N = T1 - T0;
    double vc[N + 2], vw[N], ve[N], dv0[N], dv1[N], du[N + 1];
#pragma unroll
    for (int u = 0; u < N + 2; ++u) vc[u] = val(p[ro(T0 - 1 + u)]);
#pragma unroll
    for (int t = 0; t <= N; ++t) du[t] = pu[ro(T0 + t)];
#pragma unroll
    for (int t = 0; t < N; ++t) {
      const int o = ro(T0 + t);
      vw[t] = val(p[o + ow]);
      ve[t] = val(p[o + oe]);
      dv0[t] = pv[o];
      dv1[t] = pv[o + oe];
    }
    double fy[N + 1];
#pragma unroll
    for (int t = 0; t <= N; ++t) fy[t] = sgn(du[t] * (vc[t] - vc[t + 1]));
#pragma unroll
    for (int t = 0; t < N; ++t) {
      const double fw = sgn(dv0[t] * (vw[t] - vc[t + 1]));
      const double fe = sgn(dv1[t] * (vc[t + 1] - ve[t]));
      const double d = (fw - fe + fy[t] - fy[t + 1]) * M.ra[T0 + t];
      res[T0 + t] = M.valid[T0 + t] ? d : 0.0;
    }
  };
  constexpr int DN_H = (DN_RC + 1) / 2;
  chunk(std::integral_constant<int, 0>{}, std::integral_constant<int, DN_H>{});
#ifndef PACE_EMU
  __builtin_amdgcn_sched_barrier(0);
#endif
  chunk(std::integral_constant<int, DN_H>{}, std::integral_constant<int, DN_RC>{});
}

// The cells next to / inside the corner regions of the halo -- where the reference reads corner copies, copy_corners_x_nord /
// _y_nord (delnflux.py:1009-1047): 4 x 4 blocks (minus the one cell that touches no corner) at the corners of the tile domain --
// one candidate cell per thread (threads 0 .. 63): the divergence with the index-mapped reads.  fix_at: the cell's place in the
// plane, or -1.
template <bool FIRST, class MT>
__device__ __forceinline__ double deln_corner_fix(const Geo& g, const MT& m, const double* src, const double* sdv, const double* sdu,
                                                  int i0, int j0, double d0, int& fix_at) {
  const int ilo = i0 - 3, jlo = j0 - 3;
  auto val = [&](double x) { return FIRST ? d0 * x : x; };
  auto sgn = [&](double x) { return FIRST ? x : -x; };
  double fix = 0.0;
  fix_at = -1;
  const int tid = threadIdx.x;
  if (tid < 64) {
    const int q = tid >> 4, a = tid & 3, b = (tid >> 2) & 3;
    const int gi = (q & 1) ? g.ie + a : g.is - 3 + a;
    const int gj = (q & 2) ? g.je + b : g.js - 3 + b;
    const int ii = gi - ilo, jj = gj - jlo;
    if (ii >= 0 && ii < DW && jj >= 0 && jj < DH && deln_affected(g, gi, gj)) {
      fix_at = jj * DWP + ii;
      const bool valid = ii >= 1 && ii <= DW - 2 && jj >= 1 && jj <= DH - 2 && gi >= 1 && gi + 1 < g.ni && gj >= 1 && gj + 1 < g.nj;
      if (valid) {
        auto rd = [&](int i, int j) {
          const int la = i - ilo, lb = j - jlo;
          return (la >= 0 && la < DW && lb >= 0 && lb < DH) ? val(src[lb * DWP + la]) : 0.0;
        };
        auto X = [&](int i, int j) { remap_agrid_x(g, i, j); return rd(i, j); };
        auto Y = [&](int i, int j) { remap_agrid_y(g, i, j); return rd(i, j); };
        const double ra = *(const real*)((const char*)m.rarea + deln_off(g, gi, gj));
        const double xc = X(gi, gj), yc = Y(gi, gj);
        const double fw = sgn(sdv[fix_at] * (X(gi - 1, gj) - xc));
        const double fe = sgn(sdv[fix_at + 1] * (xc - X(gi + 1, gj)));
        const double fs = sgn(sdu[fix_at] * (Y(gi, gj - 1) - yc));
        const double fn = sgn(sdu[fix_at + DWP] * (yc - Y(gi, gj + 1)));
        fix = (fw - fe + fs - fn) * ra;
      }
    }
  }
  return fix;
}

// one pass: dst = divergence of the fluxes of src (FIRST: of d0 * src, fluxes as they are; later passes: fluxes negated,
// delnflux.py:1232-1254 "fx2 = -fx2").  RC: the footprint reaches a corner of the halo and the corner copies apply
// (block-uniform: the other instance carries none of that logic).  dst == src is allowed: every thread holds its results until
// all reads are done (one barrier inside, then).  The caller synchronises after the pass.
template <bool RC, bool FIRST, class MT>
__device__ __forceinline__ void deln_pass(const Geo& g, const MT& m, const DelnMet& M, const double* src, const double* __restrict__ sdv,
                                          const double* __restrict__ sdu, double* dst, int i0, int j0, double d0) {
  const int ilo = i0 - 3, jlo = j0 - 3;
  const bool inplace = dst == src;  // block-uniform
  double res[DN_RC];
  const int c = M.c, r0 = M.r0;
  constexpr bool kExact = DH % DN_RC == 0;
  if (M.active) deln_run<FIRST>(M, src, sdv, sdu, d0, res);
  double fix = 0.0;
  int fix_at = -1;
  if (RC) fix = deln_corner_fix<FIRST>(g, m, src, sdv, sdu, i0, j0, d0, fix_at);
  if (inplace) __syncthreads();
  if (M.active) {
#pragma unroll
    for (int t = 0; t < DN_RC; ++t) {
      const int jj = r0 + t;
      bool mine = kExact || jj < DH;
      if (RC) mine = mine && !deln_affected(g, ilo + c, jlo + jj);
      if (mine) dst[jj * DWP + c] = res[t];
    }
  }
  if (RC && fix_at >= 0) dst[fix_at] = fix;
}

// The passes.  `sq` holds q on the footprint (zero outside the storage), sdv / sdu the metrics (deln_stage_metrics), `plane` is
// the scratch plane of the iterate (pitch DWP).  On return (after a barrier if a pass ran) the damping flux through a face
// is given by deln_face_x / deln_face_y.
struct DelnResult {
  const double* plane;
  double d0;     // factor still to be applied to the plane's values (no pass ran: the plane is q itself)
  bool first;    // the face fluxes are the first flux evaluation (not negated)
};
template <class MT>
__device__ __forceinline__ DelnResult deln_iterate(const Geo& g, const MT& m, const DelnMet& M, const double* sq, const double* sdv,
                                                   const double* sdu, double* plane, int i0, int j0, double d0, bool hi_order,
                                                   int nmax) {
  const int iters = hi_order ? nmax : 0;
  const int ilo = i0 - 3, jlo = j0 - 3;
  const bool rc = hi_order && (ilo < g.is || ilo + DW - 1 > g.ie) && (jlo < g.js || jlo + DH - 1 > g.je);
  for (int it = 0; it < iters; ++it) {
    if (it == 0) {
      if (rc) deln_pass<true, true>(g, m, M, sq, sdv, sdu, plane, i0, j0, d0);
      else deln_pass<false, true>(g, m, M, sq, sdv, sdu, plane, i0, j0, d0);
    } else {
      if (rc) deln_pass<true, false>(g, m, M, plane, sdv, sdu, plane, i0, j0, d0);
      else deln_pass<false, false>(g, m, M, plane, sdv, sdu, plane, i0, j0, d0);
    }
    __syncthreads();
  }
  return DelnResult{iters == 0 ? sq : plane, d0, iters == 0};
}
// flux through the x-face (west side) / y-face (south side) of footprint cell (ii, jj); `metric` = del6_v / del6_u at the face
__device__ __forceinline__ double deln_face_x(const DelnResult& R, double metric, int ii, int jj) {
  const int l = jj * DWP + ii;
  if (R.first) return metric * (R.d0 * R.plane[l - 1] - R.d0 * R.plane[l]);
  return -(metric * (R.plane[l - 1] - R.plane[l]));
}
__device__ __forceinline__ double deln_face_y(const DelnResult& R, double metric, int ii, int jj) {
  const int l = jj * DWP + ii;
  if (R.first) return metric * (R.d0 * R.plane[l - DWP] - R.d0 * R.plane[l]);
  return -(metric * (R.plane[l - DWP] - R.plane[l]));
}
