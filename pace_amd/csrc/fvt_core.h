// (device code of k_fvt.hip; tools/census/ instantiates single tile variants of it for the instruction census)
// FiniteVolumeTransport for the production tilings -- the lean form of k_fvtp2d.hip (same arithmetic, same bits).
// Reference: fv3core/pace/fv3core/stencils/fvtp2d.py:262-345, xppm.py:148-181 / yppm.py, delnflux.py:1005-1261,
// d_sw.py:63-145.
//
// Why a second kernel.  Measured on MI355X (tools/ubench, profiles/r04_ubench.txt): ONE wave issues at most one instruction
// every ~8 cycles, whatever its kind and whether or not it depends on the previous one; four waves per SIMD share an fp64 pipe
// of ~3.6 cycles per wave-instruction, a 32-bit pipe of ~1.9 and a scalar unit of ~1 per cycle per CU.  The general kernel
// walks ~4 500 instructions per wave of which ~730 are fp64 arithmetic -- a lone workgroup takes exactly 8 cycles x 4 500 --
// so its time is its instruction COUNT, and five sixths of that count is position arithmetic, predicates (exec-mask
// bookkeeping on the scalar unit), selects and spilled-SGPR traffic.  This kernel removes those instead of tuning stages:
//  * only tilings that put every tile edge on a workgroup-tile boundary (N a multiple of TI and TJ: C96, C192, C384 with
//    32 x 24) -- the footprint of every tile lies inside the storage, so no load, store or value is predicated;
//  * a thread's place in every stage is fixed once per workgroup (one y-run and one x-run), LDS and global offsets are
//    compile-time immediates on two base registers each;
//  * three barrier intervals instead of seven: [inner y sweep + q_i | inner x sweep + q_j] -> [outer x | outer y] -> cell
//    update; a run is C = 4 cells with its C + 1 interfaces (neighbouring runs both evaluate the shared interface), so the
//    inner fluxes stay in the registers of the thread that needs them again in the outer sweep;
//  * the footprint is loaded 16 bytes per lane;
//  * w's damping fluxes are never formed at the faces: dw / heat_source / diss_est (heat_diss, d_sw.py:63-103) are one more
//    divergence of the damping iterate on the thread's own column run, stored from there.
// Other tilings, ord 8 and the float32 build take k_fvtp2d.hip.
#pragma once
#include "common.h"
#include "kernels.h"

#ifndef FV_TI
#define FV_TI 32
#define FV_TJ 24
#endif
#ifndef DN_TI
#define DN_TI FV_TI
#define DN_TJ FV_TJ
#endif
#include "delnflux_core.h"

#if !defined(PACE_REAL_FLOAT) && (FV_TI % 4 == 0) && (FV_TJ % 4 == 0) && (FV_TI >= 8) && (FV_TJ >= 8) && (DN_TI == FV_TI) && \
    (DN_TJ == FV_TJ) && ((FV_TI + 6) * (FV_TJ / 4) <= 256) && ((FV_TJ + 6) * (FV_TI / 4) <= 256)
#define FVT_AVAILABLE 1
#else
#define FVT_AVAILABLE 0
#endif

#if FVT_AVAILABLE
namespace {

constexpr int TI = FV_TI, TJ = FV_TJ;
constexpr int C = 4;                      // cells per run
constexpr int NF = C + 1;                 // interfaces per run
constexpr int GXN = TI / C, GYN = TJ / C;  // runs per row / column of the tile
constexpr int QW = TI + 6, QH = TJ + 6;    // footprint
constexpr int P = QW + 1;                  // pitch of sq, sqi and the damping planes
constexpr int PJ = TI + 1;                 // pitch of sqj and ax
constexpr int NYO = TI * GYN;              // threads 0 .. NYO-1: the y-runs of the tile's own columns (they run the outer y sweep too)
constexpr int NYH = 6 * GYN;               // then the y-runs of the six halo columns
constexpr int NXO = TJ * GXN;              // threads 0 .. NXO-1: the x-runs of the tile's own rows
constexpr int NXA = QH * GXN;              // ... NXA-1: all x-runs
static_assert(DW == QW && DH == QH && DWP == P, "the damping core shares the footprint");
static_assert(GXN >= 2 && GYN >= 2, "the first and the last run of a row / column must be different runs");

struct alignas(16) D2 {
  double x, y;
};

struct FvtLds {
  double pad0[P];     // (the damping runs read one row above / below a plane without clamping)
  double sq[QH * P];  // q on the footprint, corners copied in y; never modified
  union {
    struct {
      double sqi[TJ * P];        // q advected in y, tile rows x footprint columns
      double sqj[QH * PJ];       // q advected in x, footprint rows x tile columns
      double ax[TJ * PJ];        // final x-face fluxes (EPI 1)
      double ay[(TJ + 1) * TI];  // final y-face fluxes
    } s;
    double scratch[3 * QH * P];  // damping: iterate, del6_v, del6_u
  } u;
  double pad1[P];
  double sqc[9];  // the tile's corner block with corners copied in x (corner tiles)
};

#define LDG(p, off) (*(const real*)((const char*)(p) + (off)))
#define STG(p, off) (*(real*)((char*)(p) + (off)))

// A-grid spacings of the one-sided PPM forms (common.h EdgeSpacing): four values along the sweep axis at the start (lo) or at the
// end of the tile; `fixed` is the byte offset of the run's column (y sweeps) or row (x sweeps), `step` the byte stride along the axis
__device__ __forceinline__ EdgeSpacing fvt_spacing(const real* d, unsigned fixed, int step, int s, int e, bool lo, bool hi) {
  EdgeSpacing sp;
#pragma unroll
  for (int t = 0; t < 4; ++t) sp.S[t] = 0.0, sp.E[t] = 0.0;
  if (lo) {
#pragma unroll
    for (int t = 0; t < 4; ++t) sp.S[t] = LDG(d, fixed + (unsigned)((s - 2 + t) * step));
  }
  if (hi) {
#pragma unroll
    for (int t = 0; t < 4; ++t) sp.E[t] = LDG(d, fixed + (unsigned)((e - 1 + t) * step));
  }
  return sp;
}

// one run: the NF fluxes of the interfaces between cells Q[2 .. NF+2] (Q[u] = cell first_interface - 3 + u)
template <int MORD, bool EDGE>
__device__ __forceinline__ void fvt_run(const double* Q, const double* c, bool lane_lo, bool lane_hi, const EdgeSpacing& sp, double* out) {
  if constexpr (EDGE) ppm_run_canon<MORD, NF, C>(Q, c, lane_lo, lane_hi, sp, out);
  else ppm_run_p<MORD, false, NF>(Q, c, 0, 0, 0, sp, out);
}

// ---- del-n damping, lean form (delnflux.py:1209-1261; the arithmetic of delnflux_core.h deln_run, same bits) ----------------
// A thread owns the column run (column dc, rows dr0 .. dr0 + DN_RC - 1) of the footprint.  One pass = the divergence of the
// fluxes of the iterate, every cell of the footprint evaluated as if it were an interior cell: the outermost ring of the
// footprint (after the first pass; two rings after the second) holds garbage that no tile face ever reads -- the reference
// shrinks its compute domain by one cell per pass for the same reason -- so no cell is predicated and no index is clamped
// (rows -1 and QH of a plane are the neighbouring arrays / pad rows of FvtLds).
template <bool FIRST>
__device__ __forceinline__ void fvt_deln_run(const double* p, const double* pv, const double* pu, const double* ra, double d0, double* res) {
  auto val = [&](double x) { return FIRST ? d0 * x : x; };  // (d0 * x: the reference's first statement, d2 = damp * q)
  auto sgn = [&](double x) { return FIRST ? x : -x; };      // (later passes: fx2 = -fx2, delnflux.py:1232-1254)
  auto chunk = [&](auto T0_, auto T1_) {
    constexpr int T0 = decltype(T0_)::value, T1 = decltype(T1_)::value, N = T1 - T0;
    double vc[N + 2], vw[N], ve[N], dv0[N], dv1[N], du[N + 1];
#pragma unroll
    for (int u = 0; u < N + 2; ++u) vc[u] = val(p[(T0 - 1 + u) * P]);
#pragma unroll
    for (int t = 0; t <= N; ++t) du[t] = pu[(T0 + t) * P];
#pragma unroll
    for (int t = 0; t < N; ++t) {
      vw[t] = val(p[(T0 + t) * P - 1]);
      ve[t] = val(p[(T0 + t) * P + 1]);
      dv0[t] = pv[(T0 + t) * P];
      dv1[t] = pv[(T0 + t) * P + 1];
    }
    double fy[N + 1];
#pragma unroll
    for (int t = 0; t <= N; ++t) fy[t] = sgn(du[t] * (vc[t] - vc[t + 1]));
#pragma unroll
    for (int t = 0; t < N; ++t) {
      const double fw = sgn(dv0[t] * (vw[t] - vc[t + 1]));
      const double fe = sgn(dv1[t] * (vc[t + 1] - ve[t]));
      res[T0 + t] = (fw - fe + fy[t] - fy[t + 1]) * ra[T0 + t];
    }
  };
  constexpr int H = (DN_RC + 1) / 2;
  chunk(std::integral_constant<int, 0>{}, std::integral_constant<int, H>{});
#ifndef PACE_EMU
  __builtin_amdgcn_sched_barrier(0);
#endif
  chunk(std::integral_constant<int, H>{}, std::integral_constant<int, DN_RC>{});
}

// a footprint-sized plane of a 2-D field, 16 bytes per lane (the pieces stage 0 loads of q)
struct FvtPieces {
  static constexpr int HW = QW / 2;             // 16-byte pieces per row
  static constexpr int RPP = 256 / HW;          // rows per pass
  static constexpr int NP = (QH + RPP - 1) / RPP;
  int lc, lr0, row[NP];
  unsigned off[NP];
  // whether piece p of this thread is a piece of its own (not a clamped repeat of another thread's)
  __device__ __forceinline__ bool own(int p) const { return lr0 < RPP && lr0 + RPP * p < QH; }
  __device__ __forceinline__ FvtPieces(int tid, int ilo, int jlo, int sj8) {
    int lr = tid / HW;
    lc = tid - lr * HW;
    lr0 = lr;
    if (lr >= RPP) lr = RPP - 1;  // (the spare threads repeat the last piece: same values to the same place)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      row[p] = lr + RPP * p;
      if (row[p] >= QH) row[p] = QH - 1;
      off[p] = (unsigned)((jlo + row[p]) * sj8 + (ilo + 2 * lc) * 8);
    }
  }
  __device__ __forceinline__ void load(const real* src, D2* v) const {
#pragma unroll
    for (int p = 0; p < NP; ++p) v[p] = *(const D2*)((const char*)src + off[p]);
  }
  __device__ __forceinline__ void store(double* plane, const D2* v) const {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      plane[row[p] * P + 2 * lc] = v[p].x;
      plane[row[p] * P + 2 * lc + 1] = v[p].y;
    }
  }
};

// DMODE: -1 transport only; 0 damping fluxes -> dp.fx2o / fy2o (+ dp.add2d, u / v update: the vorticity call of d_sw);
// 1 damping fluxes added to the transport fluxes; 2 added mass-weighted; 3 damping of q -> dw / heat_s / diss_est only (w).
// EPI 0: fluxes stored (or accumulated / turned into winds); 1: flux-form update of the cell stored.
template <int MORD, int DMODE, int EPI, bool EX, bool EY>
__device__ __forceinline__ void fvt_tile(FvtLds& L, const Geo& g, const FvMet& m, const real* __restrict__ q,
                                         const real* __restrict__ crx, const real* __restrict__ cry,
                                         const real* __restrict__ xfx, const real* __restrict__ yfx, real* __restrict__ fx,
                                         real* __restrict__ fy, const real* __restrict__ xunit,
                                         const real* __restrict__ yunit, const FvDamp& dp, int bx, int by, int k) {
  const int tid = threadIdx.x;
  const int i0 = g.is + bx * TI, j0 = g.js + by * TJ;
  const int ilo = i0 - 3, jlo = j0 - 3;
  const int sj8 = g.sj * 8;
  const long kb = (long)k * g.sk;
  q += kb, crx += kb, cry += kb, xfx += kb, yfx += kb, xunit += kb, yunit += kb;
  // which edges of the cubed-sphere tile this workgroup tile holds (block-uniform; one per axis at most: >= 2 tiles each way)
  const bool west = EX && bx == 0, east = EX && !west;
  const bool south = EY && by == 0, north = EY && !south;

  // ---- the thread's places: one y-run (column ycol of the footprint, rows C*yg ..), one x-run (row xrow, columns C*xg ..) ----
  int ycol, yg;
  bool y_on = true;
  if (tid < NYO) {
    yg = tid / TI;
    ycol = 3 + (tid - yg * TI);
  } else {
    const int r = tid - NYO;
    y_on = r < NYH;
    yg = y_on ? r / 6 : 0;
    const int h = y_on ? r - yg * 6 : 0;
    ycol = h < 3 ? h : TI + h;
  }
  const bool y_outer = tid < NYO;
  const int xr = tid / GXN, xg = tid - xr * GXN;
  const bool x_on = xr < QH, x_outer = xr < TJ;
  int xrow = xr + 3;  // the tile's own rows first: footprint rows 3 .. TJ+2, then TJ+3 .. QH-1, then 0 .. 2
  if (xrow >= QH) xrow -= QH;
  if (!x_on) xrow = 0;
  const int ybase = (C * yg) * P + ycol;  // sq / sqi / damping planes: element (row C*yg, column ycol)
  const int xbase = xrow * P + C * xg;    // sq: element (row xrow, column C*xg)
  // byte offsets in a level: the y-run's first interface (ilo + ycol, j0 + C*yg), the x-run's (i0 + C*xg, jlo + xrow)
  const unsigned yoff = (unsigned)((j0 + C * yg) * sj8 + (ilo + ycol) * 8);
  const unsigned xoff = (unsigned)((jlo + xrow) * sj8 + (i0 + C * xg) * 8);

  // ---- stage 0: the footprint (and the damping metrics), 16 bytes per lane (rows start 16-byte aligned: ilo = TI * bx, sj even) ----
  const FvtPieces pc(tid, ilo, jlo, sj8);
  double* const plane = L.u.scratch;
  double* const sdv = plane + QH * P;
  double* const sdu = sdv + QH * P;
  {
    D2 v[FvtPieces::NP];
    pc.load(q, v);
    if (EX && EY) {  // corner tile: the corner block holds the values copy_corners_y puts there (corners.py:367-425)
#pragma unroll
      for (int p = 0; p < FvtPieces::NP; ++p) {
        const int gj = jlo + pc.row[p];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int gi = ilo + 2 * pc.lc + e;
          if ((gi < g.is || gi > g.ie) && (gj < g.js || gj > g.je)) {
            int ri = gi, rj = gj;
            remap_agrid_y(g, ri, rj);
            const double val = LDG(q, (unsigned)(rj * sj8 + ri * 8));
            if (e == 0) v[p].x = val;
            else v[p].y = val;
          }
        }
      }
      if (tid < 9) {  // ... and what copy_corners_x puts there, for the x sweeps
        const int b = tid / 3, a = tid - b * 3;
        int ri = (west ? g.is - 3 : g.ie + 1) + a, rj = (south ? g.js - 3 : g.je + 1) + b;
        remap_agrid_x(g, ri, rj);
        L.sqc[tid] = LDG(q, (unsigned)(rj * sj8 + ri * 8));
      }
    }
    pc.store(L.sq, v);
    if (DMODE >= 0) {
      D2 mv[FvtPieces::NP], mu[FvtPieces::NP];
      pc.load(m.del6_v, mv);
      pc.load(m.del6_u, mu);
      pc.store(sdv, mv);
      pc.store(sdu, mu);
    }
  }
  // the thread's column run of the damping: column dc, rows dr0 .. dr0 + DN_RC - 1 of the footprint
  const int dr = tid / QW, dc = tid - dr * QW;
  const bool dn_on = dr < DN_NR;
  const int dr0 = dn_on ? dr * DN_RC : 0;
  const int dbase = dr0 * P + dc;
  double dra[DN_RC];
  DelnMet DM;
  constexpr bool RC = EX && EY;  // the footprint reaches a corner of the halo: the corner copies apply (delnflux_core.h)
  if (DMODE >= 0) {
    if (RC) {
      deln_load(g, m, i0, j0, DM);
    } else {
#pragma unroll
      for (int t = 0; t < DN_RC; ++t) {
        int row = dr0 + t;
        if (QH % DN_RC != 0 && row >= QH) row = QH - 1;
        dra[t] = LDG(m.rarea, (unsigned)((jlo + row) * sj8 + (ilo + dc) * 8));
      }
    }
  }
  __syncthreads();

  // ---- the del-n damping of q on the footprint (delnflux.py:1209-1261); this thread's face values stay in registers ----
  double dvx[NF], dvy[NF];
  double damp = 0.0;
  if (DMODE >= 0) {
    damp = dp.damp_k[k];
    const double d0 = dp.mass_given ? 1.0 : damp;
    const bool hi_order = dp.nord_k[k] > 0.0;
    const int iters = hi_order ? dp.nmax : 0;
    if (RC) {
      (void)deln_iterate(g, m, DM, L.sq, sdv, sdu, plane, i0, j0, d0, hi_order, dp.nmax);
    } else {
      for (int it = 0; it < iters; ++it) {
        double res[DN_RC];
        if (dn_on) {
          if (it == 0) fvt_deln_run<true>(L.sq + dbase, sdv + dbase, sdu + dbase, dra, d0, res);
          else fvt_deln_run<false>(plane + dbase, sdv + dbase, sdu + dbase, dra, d0, res);
        }
        if (it > 0) __syncthreads();  // (in place: everyone has read the iterate)
        if (dn_on) {
#pragma unroll
          for (int t = 0; t < DN_RC; ++t)
            if (QH % DN_RC == 0 || dr0 + t < QH) plane[dbase + t * P] = res[t];
        }
        __syncthreads();
      }
    }
    // the damping flux through a face from the last iterate (first == no pass ran: the first flux evaluation, of d0 * q)
    const bool first = iters == 0;
    const double* const last = first ? L.sq : plane;
    auto face = [&](double metric, double a, double b) { return first ? metric * (d0 * a - d0 * b) : -(metric * (a - b)); };
    if (DMODE != 3) {
      if (x_outer) {
#pragma unroll
        for (int f = 0; f < NF; ++f) dvx[f] = face(sdv[xbase + f + 3], last[xbase + f + 2], last[xbase + f + 3]);
      }
      if (y_outer) {
#pragma unroll
        for (int f = 0; f < NF; ++f) dvy[f] = face(sdu[ybase + (f + 3) * P], last[ybase + (f + 2) * P], last[ybase + (f + 3) * P]);
      }
    } else if (dn_on) {
      // heat_diss (d_sw.py:63-103): dw = divergence of the damping fluxes / area -- one more divergence of the iterate on this
      // thread's column run (the same expression, in the same order, as the flux-difference form of the general kernel)
      double res[DN_RC];
      if (RC) {
#pragma unroll
        for (int t = 0; t < DN_RC; ++t) dra[t] = DM.ra[t];
      }
      if (first) fvt_deln_run<true>(L.sq + dbase, sdv + dbase, sdu + dbase, dra, d0, res);
      else fvt_deln_run<false>(plane + dbase, sdv + dbase, sdu + dbase, dra, d0, res);
      const bool on = dp.damp_w_k[k] > 1e-5;
      const double dd8 = dp.ke_bg_k[k] * fabs(dp.dt);
      if (dc >= 3 && dc < TI + 3) {
#pragma unroll
        for (int t = 0; t < DN_RC; ++t) {
          const int jj = dr0 + t;
          if (jj >= 3 && jj < TJ + 3) {
            const unsigned c = (unsigned)(kb * 8) + (unsigned)((jlo + jj) * sj8 + (ilo + dc) * 8);
            double hs = 0.0;
            if (on) {
              const double d = res[t];
              const double qv = L.sq[jj * P + dc];
              STG(dp.dw, c) = d;
              hs = dd8 - d * (qv + 0.5 * d);
            }
            STG(dp.heat_s, c) = hs;
            STG(dp.diss_est, c) = hs;
          }
        }
      }
    }
    __syncthreads();  // the sweeps overwrite the damping planes
  }
  if (DMODE == 0 && dp.add2d) {
    // the transported scalar is q + add2d (absolute vorticity), the damped one was q: every thread adds to the pieces it loaded
    D2 v[FvtPieces::NP];
    pc.load(dp.add2d, v);
#pragma unroll
    for (int p = 0; p < FvtPieces::NP; ++p) {
      if (pc.own(p)) {  // (each piece once: the clamped repeats would add twice)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          double add = e == 0 ? v[p].x : v[p].y;
          if (EX && EY) {
            int gi = ilo + 2 * pc.lc + e, gj = jlo + pc.row[p];
            if ((gi < g.is || gi > g.ie) && (gj < g.js || gj > g.je)) {
              remap_agrid_y(g, gi, gj);
              add = LDG(dp.add2d, (unsigned)(gj * sj8 + gi * 8));
            }
          }
          L.sq[pc.row[p] * P + 2 * pc.lc + e] = L.sq[pc.row[p] * P + 2 * pc.lc + e] + add;
        }
      }
    }
    if (EX && EY && tid < 9) {
      const int b = tid / 3, a = tid - b * 3;
      int ri = (west ? g.is - 3 : g.ie + 1) + a, rj = (south ? g.js - 3 : g.je + 1) + b;
      remap_agrid_x(g, ri, rj);
      L.sqc[tid] = L.sqc[tid] + LDG(dp.add2d, (unsigned)(rj * sj8 + ri * 8));
    }
    __syncthreads();
  }

  // ---- stage I: the inner sweeps and the advected fields (fvtp2d.py:34-77) ----
  double si_y[NF], si_x[NF], cy[NF], cx[NF];
  if (y_on) {  // YPiecewiseParabolic on the run's five interfaces, then q_i of the four cells between them
    double Q[NF + 5], yf[NF], ar[C];
#pragma unroll
    for (int u = 0; u < NF + 5; ++u) Q[u] = L.sq[ybase + u * P];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      cy[f] = LDG(cry, yoff + (unsigned)(f * sj8));
      yf[f] = LDG(yfx, yoff + (unsigned)(f * sj8));
    }
#pragma unroll
    for (int t = 0; t < C; ++t) ar[t] = LDG(m.area, yoff + (unsigned)(t * sj8));
    EdgeSpacing sp;
    if (EY) sp = fvt_spacing(m.dya, (unsigned)((ilo + ycol) * 8), sj8, g.js, g.je, south && yg == 0, north && yg == GYN - 1);
    fvt_run<MORD, EY>(Q, cy, south && yg == 0, north && yg == GYN - 1, sp, si_y);
#pragma unroll
    for (int t = 0; t < C; ++t)
      L.u.s.sqi[ybase + t * P] = (Q[t + 3] * ar[t] + yf[t] * si_y[t] - yf[t + 1] * si_y[t + 1]) / (ar[t] + yf[t] - yf[t + 1]);
  }
  if (x_on) {  // XPiecewiseParabolic, then q_j
    double Q[NF + 5], xf[NF], ar[C];
#pragma unroll
    for (int u = 0; u < NF + 5; ++u) Q[u] = L.sq[xbase + u];
    if (EX && EY) {  // halo rows of a corner tile: the three corner columns hold the x-direction copies
      const bool halo_row = south ? xrow < 3 : xrow >= TJ + 3;
      const int b = south ? xrow : xrow - (TJ + 3);
      if (halo_row && west && xg == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) Q[a] = L.sqc[b * 3 + a];
      }
      if (halo_row && east && xg == GXN - 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a) Q[NF + 2 + a] = L.sqc[b * 3 + a];
      }
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      cx[f] = LDG(crx, xoff + (unsigned)(f * 8));
      xf[f] = LDG(xfx, xoff + (unsigned)(f * 8));
    }
#pragma unroll
    for (int t = 0; t < C; ++t) ar[t] = LDG(m.area, xoff + (unsigned)(t * 8));
    EdgeSpacing sp;
    if (EX) sp = fvt_spacing(m.dxa, (unsigned)((jlo + xrow) * sj8), 8, g.is, g.ie, west && xg == 0, east && xg == GXN - 1);
    fvt_run<MORD, EX>(Q, cx, west && xg == 0, east && xg == GXN - 1, sp, si_x);
#pragma unroll
    for (int t = 0; t < C; ++t)
      L.u.s.sqj[xrow * PJ + C * xg + t] = (Q[t + 3] * ar[t] + xf[t] * si_x[t] - xf[t + 1] * si_x[t + 1]) / (ar[t] + xf[t] - xf[t + 1]);
  }
  __syncthreads();

  // ---- stage II: the outer sweeps and the final fluxes (fvtp2d.py:80-119) ----
  const unsigned kb8 = (unsigned)(kb * 8);
  if (x_outer) {  // outer x on q_i, tile row xr (= footprint row xrow), faces i0 + C*xg + f
    double Q[NF + 5], out[NF], xu[NF], v[NF];
#pragma unroll
    for (int u = 0; u < NF + 5; ++u) Q[u] = L.u.s.sqi[xr * P + C * xg + u];
#pragma unroll
    for (int f = 0; f < NF; ++f) xu[f] = LDG(xunit, xoff + (unsigned)(f * 8));
    double ms[NF + 1];
    if (DMODE == 2) {
#pragma unroll
      for (int t = 0; t <= NF; ++t) ms[t] = LDG(dp.mass, kb8 + xoff + (unsigned)((t - 1) * 8));
    }
    EdgeSpacing sp;
    if (EX) sp = fvt_spacing(m.dxa, (unsigned)((jlo + xrow) * sj8), 8, g.is, g.ie, west && xg == 0, east && xg == GXN - 1);
    fvt_run<MORD, EX>(Q, cx, west && xg == 0, east && xg == GXN - 1, sp, out);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      v[f] = 0.5 * (out[f] + si_x[f]) * xu[f];
      if (DMODE == 1) v[f] = v[f] + dvx[f];
      if (DMODE == 2) v[f] = v[f] + 0.5 * damp * (ms[f] + ms[f + 1]) * dvx[f];
    }
    if (EPI == 0) {
      // a face is stored by the run it opens; the last face of the row (ie + 1) by the last run of the east-edge tile
      const bool last = east && xg == GXN - 1;
      double w0[NF], w1[NF], w2[NF], w3[NF], wa[NF];
      if (dp.v_upd) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          w0[f] = LDG(dp.v_upd, kb8 + xoff + (unsigned)(f * 8));
          w1[f] = LDG(m.dy, xoff + (unsigned)(f * 8));
          w2[f] = LDG(dp.ke, kb8 + xoff + (unsigned)(f * 8));
          w3[f] = LDG(dp.ke, kb8 + xoff + (unsigned)(f * 8 + sj8));
        }
      }
      if (dp.accx) {
#pragma unroll
        for (int f = 0; f < NF; ++f) wa[f] = LDG(dp.accx, kb8 + xoff + (unsigned)(f * 8));
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        if (f < C || last) {
          const unsigned c = kb8 + xoff + (unsigned)(f * 8);
          if (DMODE == 0) STG(dp.fx2o, c) = dvx[f];
          if (dp.v_upd) STG(dp.v_out ? dp.v_out : dp.v_upd, c) = w0[f] * w1[f] + w2[f] - w3[f] - v[f];  // v_from_ke (d_sw.py:423-436)
          else STG(fx, c) = v[f];
          if (dp.accx) STG(dp.accx, c) = wa[f] + v[f];
        }
      }
    } else {
#pragma unroll
      for (int f = 0; f < NF; ++f) L.u.s.ax[xr * PJ + C * xg + f] = v[f];
    }
  }
  if (y_outer) {  // outer y on q_j, tile column ycol - 3, faces j0 + C*yg + f
    double Q[NF + 5], out[NF], yu[NF], v[NF];
#pragma unroll
    for (int u = 0; u < NF + 5; ++u) Q[u] = L.u.s.sqj[(C * yg + u) * PJ + ycol - 3];
#pragma unroll
    for (int f = 0; f < NF; ++f) yu[f] = LDG(yunit, yoff + (unsigned)(f * sj8));
    double ms[NF + 1];
    if (DMODE == 2) {
#pragma unroll
      for (int t = 0; t <= NF; ++t) ms[t] = LDG(dp.mass, kb8 + yoff + (unsigned)((t - 1) * sj8));
    }
    EdgeSpacing sp;
    if (EY) sp = fvt_spacing(m.dya, (unsigned)((ilo + ycol) * 8), sj8, g.js, g.je, south && yg == 0, north && yg == GYN - 1);
    fvt_run<MORD, EY>(Q, cy, south && yg == 0, north && yg == GYN - 1, sp, out);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      v[f] = 0.5 * (out[f] + si_y[f]) * yu[f];
      if (DMODE == 1) v[f] = v[f] + dvy[f];
      if (DMODE == 2) v[f] = v[f] + 0.5 * damp * (ms[f] + ms[f + 1]) * dvy[f];
    }
    if (EPI == 0) {
      const bool last = north && yg == GYN - 1;
      double w0[NF], w1[NF], w2[NF], w3[NF], wa[NF];
      if (dp.u_upd) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          w0[f] = LDG(dp.u_upd, kb8 + yoff + (unsigned)(f * sj8));
          w1[f] = LDG(m.dx, yoff + (unsigned)(f * sj8));
          w2[f] = LDG(dp.ke, kb8 + yoff + (unsigned)(f * sj8));
          w3[f] = LDG(dp.ke, kb8 + yoff + (unsigned)(f * sj8 + 8));
        }
      }
      if (dp.accy) {
#pragma unroll
        for (int f = 0; f < NF; ++f) wa[f] = LDG(dp.accy, kb8 + yoff + (unsigned)(f * sj8));
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        if (f < C || last) {
          const unsigned c = kb8 + yoff + (unsigned)(f * sj8);
          if (DMODE == 0) STG(dp.fy2o, c) = dvy[f];
          if (dp.u_upd) STG(dp.u_out ? dp.u_out : dp.u_upd, c) = w0[f] * w1[f] + w2[f] - w3[f] + v[f];  // u_from_ke (d_sw.py:406-420)
          else STG(fy, c) = v[f];
          if (dp.accy) STG(dp.accy, c) = wa[f] + v[f];
        }
      }
    } else {
#pragma unroll
      for (int f = 0; f < NF; ++f) L.u.s.ay[(C * yg + f) * TI + ycol - 3] = v[f];
    }
  }
  if (EPI == 1) {
    // apply_fluxes (d_sw.py:122-145): q * mass + the flux increment, one cell per lane, lanes along i
    __syncthreads();
    constexpr int NEC = (TI * TJ + 255) / 256;
    double ra[NEC], am[NEC];
    int jj[NEC], ii[NEC];
    unsigned c2[NEC];
#pragma unroll
    for (int t = 0; t < NEC; ++t) {
      int e = tid + 256 * t;
      if (e >= TI * TJ) e = TI * TJ - 1;  // (spare lanes repeat the last cell)
      jj[t] = e / TI, ii[t] = e - jj[t] * TI;
      c2[t] = (unsigned)((j0 + jj[t]) * sj8 + (i0 + ii[t]) * 8);
      ra[t] = LDG(m.rarea, c2[t]);
      am[t] = LDG(dp.amass, kb8 + c2[t]);
    }
#pragma unroll
    for (int t = 0; t < NEC; ++t) {
      const double qv = L.sq[(jj[t] + 3) * P + ii[t] + 3];
      const double* ax = L.u.s.ax + jj[t] * PJ + ii[t];
      const double* ay = L.u.s.ay + jj[t] * TI + ii[t];
      STG(dp.qout, kb8 + c2[t]) = qv * am[t] + (ax[0] - ax[1] + ay[0] - ay[TI]) * ra[t];
    }
  }
}

}  // namespace
#endif  // FVT_AVAILABLE
