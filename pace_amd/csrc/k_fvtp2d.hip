// FiniteVolumeTransport (Fortran fv_tp_2d): Putman-Lin 2-D transport fluxes of one scalar.
// Reference: fv3core/pace/fv3core/stencils/fvtp2d.py:262-345 -- 9 stencil launches (copy_corners_y,
// yppm inner, q_i, xppm outer, copy_corners_x, xppm inner, q_j, yppm outer, final_fluxes) through 6
// full 3-D temporaries.  Here: ONE kernel.  A workgroup owns a TI x TJ tile of one level, stages the
// (TI+6) x (TJ+6) footprint of q in LDS, runs the five sweeps tile-locally and writes only the two
// flux fields.  HBM-bound: algorithmic traffic = 5-7 reads + 2 writes of 3-D fields.
#include <cstdlib>

#include "common.h"
#include "kernels.h"

#ifndef FV_TI
#define FV_TI 32
#define FV_TJ 24  // 32 x 24 cells, runs of 5 interfaces: best of the tile shapes measured at C192
#endif
#define TI FV_TI
#define TJ FV_TJ
#ifndef DN_TI
#define DN_TI FV_TI
#define DN_TJ FV_TJ
#endif
#include "delnflux_core.h"
static_assert(DN_TI == FV_TI && DN_TJ == FV_TJ, "the fused damping shares the transport tile");
// Global accesses: uniform field base (SGPR pair) + 32-bit byte offset per lane, so that loads / stores take the
// `global_load ... v_off, s[base:base+1]` form and the offset arithmetic is two full-rate 32-bit instructions
// (v_mad_i32_i24 + shift) instead of a quarter-rate 64-bit multiply-add and a 64-bit add per access.  (A field is < 4 GB.)
#define OFF2(gi, gj) ((unsigned)(__mul24((gj), sj8) + ((gi) << REAL_SHIFT)))
#define LD(p, off) (*(const real*)((const char*)(p) + (off)))
#define ST(p, off) (*(real*)((char*)(p) + (off)))
#define QW (TI + 6)
#define QH (TJ + 6)


// EX / EY: whether any x- (y-) interface this workgroup evaluates lies within two cells of a tile edge, where the
// PPM interface values switch to the one-sided forms (xppm.py:148-181).  Block-uniform, so interior workgroups
// (25 of 35 at C192) run straight-line code without the per-interface position tests.
// LDS of one workgroup.  Occupancy is what this kernel lives on (a workgroup's time is set by its ~15 barrier-separated,
// latency-bound stages; measured: throughput is proportional to resident workgroups per CU), so the arrays are kept
// to a minimum: q advected in x (q_j) overwrites q in place, and the three scratch planes of the fused damping alias
// the sweep arrays (padded only as far as needed).  32 x 24 tile: 32.6 KB (5 workgroups / CU) plain, 37.4 KB (4) fused.
template <int DMODE, int EPI>
struct FvLds {
  double sq[QH][QW + 1];        // q on [i0-3, i0+TI+3) x [j0-3, j0+TJ+3); columns 3 .. TI+2 become q_j in stage 4
  double syin[TJ + 1][QW + 1];  // inner y sweep: mean advected value on y-interfaces
  double sqi[TJ][QW + 1];       // q advected in y (fvtp2d.py:34-56)
  double sxin[QH][TI + 1];      // inner x sweep on x-interfaces
  static constexpr int kSweep = (TJ + 1) * (QW + 1) + TJ * (QW + 1) + QH * (TI + 1);
  static constexpr int kDamp = DMODE >= 0 ? 3 * (TJ + 6) * (TI + 7) : 0;
  static constexpr int kEpi = EPI > 0 ? 2 * (TJ * (TI + 2) + (TJ + 1) * (TI + 1)) : 0;
  static constexpr int kNeed = kDamp > kEpi ? kDamp : kEpi;
  static constexpr int kPad = kNeed > kSweep ? kNeed - kSweep : 1;
  double pad[kPad];
};

#ifndef FV_EPI0_BATCH
#define FV_EPI0_BATCH 2
#endif

#ifndef FV_RF
#define FV_RF 5
#endif
#define RF FV_RF                          // interfaces per thread in a PPM run
#define GY ((TJ + 1 + RF - 1) / RF)       // runs per column for the TJ+1 y-interfaces
#define GX ((TI + 1 + RF - 1) / RF)       // runs per row for the TI+1 x-interfaces
// whether a y-run can reach past the last footprint row (then its row index is clamped); not at the production shape
constexpr bool kClampY = (GY - 1) * RF + RF + 4 >= QH;
static_assert((GX - 1) * RF + RF + 4 < 2 * (QW + 1), "an x-run may overrun its row by less than one row");
static_assert(QW * GY <= 256 && QH * GX <= 256 && (TJ + 3) * GX <= 256, "one PPM run per thread: the tile is too large for 256 threads");

// the A-grid spacings the one-sided PPM forms need (common.h EdgeSpacing), for a sweep along x in row gj (AXIS 0: dxa) or along y
// in column gi (AXIS 1: dya): unconditional loads (every position is inside the storage), only in workgroups next to that edge
template <int AXIS>
__device__ __forceinline__ EdgeSpacing load_edge_spacing(const Geo& g, const real* d, int fixed, bool near_s, bool near_e, int sj8) {
  EdgeSpacing sp;
#pragma unroll
  for (int t = 0; t < 4; ++t) sp.S[t] = 0.0, sp.E[t] = 0.0;
  const int s0 = (AXIS == 0 ? g.is : g.js) - 2, e0 = (AXIS == 0 ? g.ie : g.je) - 1;
  if (near_s) {
#pragma unroll
    for (int t = 0; t < 4; ++t) sp.S[t] = AXIS == 0 ? LD(d, OFF2(s0 + t, fixed)) : LD(d, OFF2(fixed, s0 + t));
  }
  if (near_e) {
#pragma unroll
    for (int t = 0; t < 4; ++t) sp.E[t] = AXIS == 0 ? LD(d, OFF2(e0 + t, fixed)) : LD(d, OFF2(fixed, e0 + t));
  }
  return sp;
}

template <int MORD, bool EX, bool EY, int DMODE, int EPI, bool CANON>
__device__ __forceinline__ void fvtp2d_tile(FvLds<DMODE, EPI>& L, const Geo& g, const FvMet& m, const real* __restrict__ q,
                                            const real* __restrict__ crx, const real* __restrict__ cry,
                                            const real* __restrict__ xfx, const real* __restrict__ yfx,
                                            real* __restrict__ fx, real* __restrict__ fy,
                                            const real* __restrict__ xunit, const real* __restrict__ yunit,
                                            const FvDamp& dp, const FvTile wg) {
  auto& sq = L.sq;
  auto& syin = L.syin;
  auto& sqi = L.sqi;
  auto& sxin = L.sxin;

  const int tid = threadIdx.x;
  const int i0 = g.is + wg.bx * TI;
  const int j0 = g.js + wg.by * TJ;
  const int k = wg.bz;
  const long kb = (long)k * g.sk;
  const int ilo = i0 - 3, jlo = j0 - 3;
  const int sj = g.sj;
  const int sj8 = sj * (int)sizeof(real);  // byte strides (the names date from the fp64-only kernel)
  constexpr int E8 = (int)sizeof(real);
  const unsigned kb8 = (unsigned)(kb * (long)sizeof(real));
  // Thread maps of the outer sweeps (stage 5): the thread that ran the inner x-run of footprint row jj+3 / the inner y-run
  // of footprint column ii+3 also runs the outer run of tile row jj / tile column ii, so the Courant numbers it loaded
  // for the inner sweep are reused from registers instead of being fetched a second time (by then evicted from L2).
  const int x5_row = tid / GX - 3, x5_grp = tid % GX;
  const bool x5_on = x5_row >= 0 && x5_row < TJ;
  const int y5_grp = tid / QW, y5_col = tid % QW - 3;
  const bool y5_on = y5_grp < GY && y5_col >= 0 && y5_col < TI;
  double cx_keep[RF], cy_keep[RF];
  // which tile edges the one-sided PPM forms of this workgroup can touch (block-uniform; see the EX / EY tests in k_fvtp2d)
  const bool near_w = i0 - 1 <= g.is + 1, near_e = i0 + TI + 1 >= g.ie, near_s = j0 - 1 <= g.js + 1, near_n = j0 + TJ + 1 >= g.je;
  // ... and whether they sit where ppm_run_canon expects them: the tile starts at the edge (first run, A = 0) / ends exactly at
  // it (last run, A = AHI).  True for every edge tile when N is a multiple of the tile size (C48 ... C384 with 32 x 24: in x;
  // in y when 24 divides N); other tilings take the general form.
  constexpr int AHI_X = TI - (GX - 1) * RF, AHI_Y = TJ - (GY - 1) * RF;
  // (AHI >= 2: the run before the last one does not reach the interface at e; RF >= 3: the second run does not reach s + 1 --
  // the interface values of neighbouring runs overlap by two)
  constexpr bool kCanonX = MORD != 8 && RF >= 3 && AHI_X >= 2 && AHI_X <= RF - 1 && GX >= 2;
  constexpr bool kCanonY = MORD != 8 && RF >= 3 && AHI_Y >= 2 && AHI_Y <= RF - 1 && GY >= 2;
  // (CANON: the launcher has checked that the tiling puts every edge there -- fv_canonical_tiling)
  constexpr bool canon_x = CANON && kCanonX, canon_y = CANON && kCanonY;

  DelnMet DM;
  if (DMODE >= 0) {
    deln_stage_metrics(g, m, &L.syin[0][0] + DH * DWP, &L.syin[0][0] + 2 * DH * DWP, i0, j0);
  }
  // stage 0: q with corners copied in the y direction (copy_corners_y, corners.py:367-425).
  // All of a thread's loads are issued before the first of them is consumed (the straightforward loop waits for every
  // iteration's load before issuing the next one: 4.5 exposed memory latencies per workgroup, measured with FV_PROF).
  {
    constexpr int NE0 = (QW * QH + 255) / 256;
    double v0[NE0];
#pragma unroll
    for (int t = 0; t < NE0; ++t) {
      const int e = tid + 256 * t;
      const int jj = e / QW, ii = e - jj * QW;
      int gi = ilo + ii, gj = jlo + jj;
      // (a workgroup with EX false has its whole footprint inside the compute domain in x, likewise EY in y; corner cells
      // -- outside in both -- exist only for EX && EY)
      const bool ok = e < QW * QH && (!EX || (gi >= 0 && gi < g.ni)) && (!EY || (gj >= 0 && gj < g.nj));
      if (EX && EY) remap_agrid_y(g, gi, gj);
      v0[t] = ok ? LD(q, kb8 + OFF2(gi, gj)) : 0.0;
    }
#pragma unroll
    for (int t = 0; t < NE0; ++t) {
      const int e = tid + 256 * t;
      const int jj = e / QW, ii = e - jj * QW;
      if (e < QW * QH) sq[jj][ii] = v0[t];
    }
  }
  __syncthreads();

  // fused damping: the passes run in the LDS space the sweeps will use afterwards; this thread's face values stay in registers
  double dvx[RF], dvy[RF];
  double damp = 0.0;
  if (DMODE >= 0) {
    static_assert(DMODE < 0 || sizeof(L.syin) + sizeof(L.sqi) + sizeof(L.sxin) + sizeof(L.pad) >= 3 * DH * DWP * sizeof(double),
                  "scratch");
    double* pa = &L.syin[0][0];
    const double* sdv = pa + DH * DWP;
    const double* sdu = sdv + DH * DWP;
    damp = dp.damp_k[k];
    deln_load(g, m, i0, j0, DM);
    const DelnResult R = deln_iterate(g, m, DM, &sq[0][0], sdv, sdu, pa, i0, j0, dp.mass_given ? 1.0 : damp, dp.nord_k[k] > 0.0, dp.nmax);
    if (x5_on) {
#pragma unroll
      for (int f = 0; f < RF; ++f) dvx[f] = (x5_grp * RF + f <= TI) ? deln_face_x(R, sdv[(x5_row + 3) * DWP + x5_grp * RF + f + 3], x5_grp * RF + f + 3, x5_row + 3) : 0.0;
    }
    if (y5_on) {
#pragma unroll
      for (int f = 0; f < RF; ++f) dvy[f] = (y5_grp * RF + f <= TJ) ? deln_face_y(R, sdu[(y5_grp * RF + f + 3) * DWP + y5_col + 3], y5_col + 3, y5_grp * RF + f + 3) : 0.0;
    }
    __syncthreads();
  }
  if (dp.add2d) {  // block-uniform
    for (int e = tid; e < QW * QH; e += 256) {
      const int jj = e / QW, ii = e - jj * QW;
      int gi = ilo + ii, gj = jlo + jj;
      if ((!EX || (gi >= 0 && gi < g.ni)) && (!EY || (gj >= 0 && gj < g.nj))) {
        if (EX && EY) remap_agrid_y(g, gi, gj);
        sq[jj][ii] = sq[jj][ii] + LD(dp.add2d, OFF2(gi, gj));
      }
    }
    __syncthreads();
  }

  // stage 1: inner y sweep (YPiecewiseParabolic, origin (is-3, js), domain (N+7, N+1)): one run of RF interfaces of one
  // column per thread, lanes along i
  if (tid < QW * GY) {
    const int grp = tid / QW, ii = tid - grp * QW;
    const int jj0 = grp * RF;
    const int gi = ilo + ii, gj0 = j0 + jj0;
    const bool col_ok = !EX || (gi >= 0 && gi <= g.ni - 1);
    double Q[RF + 5], cc[RF], out[RF];
#pragma unroll
    for (int u = 0; u < RF + 5; ++u) Q[u] = sq[kClampY ? ((jj0 + u < QH) ? jj0 + u : QH - 1) : jj0 + u][ii];  // rows gj0-3 ..
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      const int gj = gj0 + f;
      // (unconditional load from a clamped address + select: a load under a condition is a branch per interface)
      const bool okf = col_ok && jj0 + f <= TJ && (!EY || (gj >= g.js && gj <= g.je + 1));
      const double cv = LD(cry, kb8 + (okf ? OFF2(gi, gj) : OFF2(g.is, g.js)));
      cc[f] = okf ? cv : 0.0;
      cy_keep[f] = cc[f];
    }
    const int gic = gi < 0 ? 0 : (gi < g.ni ? gi : g.ni - 1);  // (a tile may stick out of the storage: the metric column must exist)
    EdgeSpacing sp;
    if (EY) sp = load_edge_spacing<1>(g, m.dya, gic, near_s, near_n, sj8);
    if constexpr (EY && canon_y) ppm_run_canon<MORD == 8 ? 6 : MORD, RF, kCanonY ? AHI_Y : 0>(Q, cc, near_s && grp == 0, near_n && grp == GY - 1, sp, out);
    else ppm_run_p<MORD, EY, RF>(Q, cc, gj0, g.js, g.je, sp, out);
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      const int gj = gj0 + f;
      if (jj0 + f <= TJ) syin[jj0 + f][ii] = (col_ok && (!EY || (gj >= g.js && gj <= g.je + 1))) ? out[f] : 0.0;
    }
  }
  __syncthreads();

  // stage 2: q_i; and re-stage the corner cells of q for the x direction (copy_corners_x)
  {
    // One thread per column and group of RPG consecutive rows (round 3; before: one cell per thread and pass, every cell loading
    // both of its faces' area fluxes and reading both advected values from LDS): a face between two cells of the run is loaded
    // / read once, and the thread's position is computed once.  Same expression per cell.
    constexpr int NG2 = (256 / QW) < TJ ? (256 / QW) : TJ;  // row groups per column (6 in production: 228 threads)
    constexpr int RPG = (TJ + NG2 - 1) / NG2;                // rows per thread (4)
    const int grp2 = tid / QW, ii = tid - grp2 * QW;
    const int jj0 = grp2 * RPG;
    const int gi = ilo + ii;
    const bool act = grp2 < NG2;
    const bool col_ok = act && (!EX || (gi >= 0 && gi <= g.ni - 1));
    double y_[RPG + 1], a_[RPG], sy_[RPG + 1];
#pragma unroll
    for (int t = 0; t <= RPG; ++t) {  // loads first ...
      const int gj = j0 + jj0 + t;
      const bool row_ok = jj0 + t <= TJ && (!EY || (gj >= g.js && gj <= g.je + 1));
      const unsigned c2 = (col_ok && row_ok) ? OFF2(gi, gj) : OFF2(g.is, g.js);
      y_[t] = LD(yfx, kb8 + c2);
      if (t < RPG) a_[t] = LD(m.area, c2);
    }
    if (act) {
#pragma unroll
      for (int t = 0; t <= RPG; ++t) sy_[t] = (jj0 + t <= TJ) ? syin[jj0 + t][ii] : 0.0;
#pragma unroll
      for (int t = 0; t < RPG; ++t) {  // ... then the arithmetic
        const int jj = jj0 + t, gj = j0 + jj;
        if (jj < TJ) {
          const bool ok = col_ok && (!EY || (gj >= g.js && gj <= g.je));
          const double y0 = y_[t], y1 = y_[t + 1], a = a_[t];
          const double val = (sq[jj + 3][ii] * a + y0 * sy_[t] - y1 * sy_[t + 1]) / (a + y0 - y1);
          sqi[jj][ii] = ok ? val : 0.0;
        }
      }
    }
  }
  {
    const bool icorner_tile = (i0 == g.is) || (i0 + TI + 3 > g.ie + 1);
    const bool jcorner_tile = (j0 == g.js) || (j0 + TJ + 3 > g.je + 1);
    if (icorner_tile && jcorner_tile) {
      for (int e = tid; e < QW * QH; e += 256) {
        const int jj = e / QW, ii = e - jj * QW;
        int gi = ilo + ii, gj = jlo + jj;
        if (gi >= 0 && gi < g.ni && gj >= 0 && gj < g.nj && (gi < g.is || gi > g.ie) && (gj < g.js || gj > g.je)) {
          remap_agrid_x(g, gi, gj);
          double v = LD(q, kb8 + OFF2(gi, gj));
          if (dp.add2d) v = v + LD(dp.add2d, OFF2(gi, gj));
          sq[jj][ii] = v;
        }
      }
    }
  }
  __syncthreads();

  // stage 3: inner x sweep (XPiecewiseParabolic, origin (is, js-3), domain (N+1, N+7)): one run of RF interfaces of one
  // row per thread
  if (tid < QH * GX) {
    const int jj = tid / GX, grp = tid - jj * GX;
    const int ii0 = grp * RF;
    const int gi0 = i0 + ii0, gj = jlo + jj;
    const bool row_ok = !EY || (gj >= 0 && gj <= g.nj - 1);
    double Q[RF + 5], cc[RF], out[RF];
    // (no clamp of the column index: a run that sticks out of the row reads the pad column / the first element of the next
    // row -- LDS that exists -- and only interfaces that are masked below see it)
    const double* qrow = &sq[0][0] + jj * (QW + 1) + ii0;
#pragma unroll
    for (int u = 0; u < RF + 5; ++u) Q[u] = qrow[u];  // columns gi0-3 ..
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      const int gi = gi0 + f;
      const bool okf = row_ok && ii0 + f <= TI && (!EX || (gi >= g.is && gi <= g.ie + 1));
      const double cv = LD(crx, kb8 + (okf ? OFF2(gi, gj) : OFF2(g.is, g.js)));
      cc[f] = okf ? cv : 0.0;
      cx_keep[f] = cc[f];
    }
    const int gjc = gj < 0 ? 0 : (gj < g.nj ? gj : g.nj - 1);  // (a tile may stick out of the storage: the metric row must exist)
    EdgeSpacing sp;
    if (EX) sp = load_edge_spacing<0>(g, m.dxa, gjc, near_w, near_e, sj8);
    if constexpr (EX && canon_x) ppm_run_canon<MORD == 8 ? 6 : MORD, RF, kCanonX ? AHI_X : 0>(Q, cc, near_w && grp == 0, near_e && grp == GX - 1, sp, out);
    else ppm_run_p<MORD, EX, RF>(Q, cc, gi0, g.is, g.ie, sp, out);
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      const int gi = gi0 + f;
      if (ii0 + f <= TI) sxin[jj][ii0 + f] = (row_ok && (!EX || (gi >= g.is && gi <= g.ie + 1))) ? out[f] : 0.0;
    }
  }
  __syncthreads();

  // stage 4: q_j
  {
    // the same along x: one thread per row and group of CPG consecutive columns (QH * TI / CPG = 240 threads in production)
    constexpr int CPG = 4;
    static_assert(TI % CPG == 0 && QH * (TI / CPG) <= 256, "q_j run mapping");
    constexpr int NGX4 = TI / CPG;
    const int jj = tid / NGX4, ii0 = (tid - jj * NGX4) * CPG;
    const int gj = jlo + jj;
    const bool act = jj < QH;
    const bool row_ok = act && (!EY || (gj >= 0 && gj <= g.nj - 1));
    double x_[CPG + 1], a_[CPG], sx_[CPG + 1];
#pragma unroll
    for (int t = 0; t <= CPG; ++t) {
      const int gi = i0 + ii0 + t;
      const bool col_ok = !EX || (gi >= g.is && gi <= g.ie + 1);
      const unsigned c2 = (row_ok && col_ok) ? OFF2(gi, gj) : OFF2(g.is, g.js);
      x_[t] = LD(xfx, kb8 + c2);
      if (t < CPG) a_[t] = LD(m.area, c2);
    }
    if (act) {
#pragma unroll
      for (int t = 0; t <= CPG; ++t) sx_[t] = sxin[jj][ii0 + t];
#pragma unroll
      for (int t = 0; t < CPG; ++t) {
        const int ii = ii0 + t, gi = i0 + ii;
        const bool ok = row_ok && (!EX || (gi >= g.is && gi <= g.ie));
        const double x0 = x_[t], x1 = x_[t + 1], a = a_[t];
        const double val = (sq[jj][ii + 3] * a + x0 * sx_[t] - x1 * sx_[t + 1]) / (a + x0 - x1);
        sq[jj][ii + 3] = ok ? val : 0.0;  // q_j in place: this thread is the only one that reads or writes this cell in this stage
      }
    }
  }
  __syncthreads();

  // stage 5: outer sweeps + final_fluxes (fvtp2d.py:80-119).  The grid has ceil(N / TI) x ceil(N / TJ) workgroups; the
  // N+1-th face row / column (ie+1, je+1) is produced by the workgroup that owns cell ie / je, not by an extra,
  // almost empty row of workgroups.
  double vxf[RF], vyf[RF];  // this thread's final face fluxes (kept for the epilogue)
  if (x5_on) {  // outer x on q_i: rows of the tile, runs of x-interfaces
    const int jj = x5_row, grp = x5_grp;
    const int ii0 = grp * RF;
    const int gi0 = i0 + ii0, gj = j0 + jj;
    double Q[RF + 5], cc[RF], out[RF];
    bool calc[RF];
    const double* qirow = &sqi[0][0] + jj * (QW + 1) + ii0;
#pragma unroll
    for (int u = 0; u < RF + 5; ++u) Q[u] = qirow[u];  // q_i at gi0-3 .. (unclamped, as in stage 3)
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      const int ii = ii0 + f, gi = gi0 + f;
      calc[f] = (!EY || gj <= g.je) && (!EX || gi <= g.ie + 1) && ii <= TI;
      if (EPI == 0) calc[f] = calc[f] && (ii < TI || gi == g.ie + 1);  // a neighbour stores its own west face
      cc[f] = calc[f] ? cx_keep[f] : 0.0;  // = crx[c], loaded by this thread for the inner sweep of the same row
    }
    const int gjc = gj < g.nj ? gj : g.nj - 1;  // (a tile may stick out of the storage: the metric row must exist)
    EdgeSpacing sp;
    if (EX) sp = load_edge_spacing<0>(g, m.dxa, gjc, near_w, near_e, sj8);
    if constexpr (EX && canon_x) ppm_run_canon<MORD == 8 ? 6 : MORD, RF, kCanonX ? AHI_X : 0>(Q, cc, near_w && grp == 0, near_e && grp == GX - 1, sp, out);
    else ppm_run_p<MORD, EX, RF>(Q, cc, gi0, g.is, g.ie, sp, out);
#pragma unroll
    for (int f = 0; f < RF; ++f) vxf[f] = 0.0;
    // The face loop.  All operands of a batch of faces are loaded first, from clamped addresses and without conditions: a load
    // inside `if (calc[f])` is a branch and a full wait per face (the ISA showed 5 + 5 serialized L2 round trips per tile in
    // these two loops).  Modes that only need the unit flux (+ mass) take all faces in one batch; EPI == 0 with its wind /
    // accumulator operands (up to five more loads per face) takes two, for the registers.
    auto faces = [&](auto F0_, auto F1_) {
      constexpr int F0 = decltype(F0_)::value, F1 = decltype(F1_)::value, NF = F1 - F0;
      unsigned cf[NF];
      double xu[NF], si[NF], ms[NF + 1], w0[NF], w1[NF], w2[NF], w3[NF], wa[NF];
#pragma unroll
      for (int t = 0; t < NF; ++t) {
        const int f = F0 + t;
        cf[t] = kb8 + (calc[f] ? OFF2(gi0 + f, gj) : OFF2(g.is, g.js));
        xu[t] = LD(xunit, cf[t]);
        si[t] = sxin[jj + 3][ii0 + f];
      }
      if (DMODE == 2) {
#pragma unroll
        for (int t = 0; t <= NF; ++t) {
          const int gim = gi0 + F0 - 1 + t;
          ms[t] = LD(dp.mass, kb8 + OFF2(gim < g.ni ? gim : g.ni - 1, gjc));
        }
      }
      if (EPI == 0) {
        if (dp.v_upd) {
#pragma unroll
          for (int t = 0; t < NF; ++t) {
            w0[t] = LD(dp.v_upd, cf[t]);
            w1[t] = LD(m.dy, cf[t] - kb8);
            w2[t] = LD(dp.ke, cf[t]);
            w3[t] = LD(dp.ke, cf[t] + sj8);
          }
        }
        if (dp.accx) {
#pragma unroll
          for (int t = 0; t < NF; ++t) wa[t] = LD(dp.accx, cf[t]);
        }
      }
#pragma unroll
      for (int t = 0; t < NF; ++t) {
        const int f = F0 + t;
        if (calc[f]) {
          const unsigned c = cf[t];
          double v = 0.5 * (out[f] + si[t]) * xu[t];
          if (DMODE == 0 && EPI == 0) ST(dp.fx2o, c) = dvx[f];
          if (DMODE == 1) v = v + dvx[f];
          if (DMODE == 2) v = v + 0.5 * damp * (ms[t] + ms[t + 1]) * dvx[f];
          if (EPI == 0) {
            if (dp.v_upd) {  // v_from_ke (d_sw.py:423-436): same expression, same order as the stand-alone kernel
              ST(dp.v_out ? dp.v_out : dp.v_upd, c) = w0[t] * w1[t] + w2[t] - w3[t] - v;
            } else {
              ST(fx, c) = v;
            }
            if (dp.accx && (ii0 + f < TI || gi0 + f == g.ie + 1)) ST(dp.accx, c) = wa[t] + v;
          }
          vxf[f] = v;
        }
      }
    };
    if (EPI == 0) {  // batches of FV_EPI0_BATCH faces (three of them at RF = 5: 114 spilled VGPRs with two larger ones)
      constexpr int B = FV_EPI0_BATCH;
      faces(std::integral_constant<int, 0>{}, std::integral_constant<int, (B < RF ? B : RF)>{});
      if constexpr (B < RF) faces(std::integral_constant<int, B>{}, std::integral_constant<int, (2 * B < RF ? 2 * B : RF)>{});
      if constexpr (2 * B < RF) faces(std::integral_constant<int, 2 * B>{}, std::integral_constant<int, (3 * B < RF ? 3 * B : RF)>{});
      if constexpr (3 * B < RF) faces(std::integral_constant<int, 3 * B>{}, std::integral_constant<int, (4 * B < RF ? 4 * B : RF)>{});
      if constexpr (4 * B < RF) faces(std::integral_constant<int, 4 * B>{}, std::integral_constant<int, RF>{});
    } else {
      faces(std::integral_constant<int, 0>{}, std::integral_constant<int, RF>{});
    }
  }
  if (y5_on) {  // outer y on q_j: columns of the tile, runs of y-interfaces, lanes along i
    const int grp = y5_grp, ii = y5_col;
    const int jj0 = grp * RF;
    const int gi = i0 + ii, gj0 = j0 + jj0;
    double Q[RF + 5], cc[RF], out[RF];
    bool calc[RF];
#pragma unroll
    for (int u = 0; u < RF + 5; ++u) Q[u] = sq[kClampY ? ((jj0 + u < QH) ? jj0 + u : QH - 1) : jj0 + u][ii + 3];  // q_j at gj0-3 ..
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      const int jj = jj0 + f, gj = gj0 + f;
      calc[f] = (!EX || gi <= g.ie) && (!EY || gj <= g.je + 1) && jj <= TJ;
      if (EPI == 0) calc[f] = calc[f] && (jj < TJ || gj == g.je + 1);
      cc[f] = calc[f] ? cy_keep[f] : 0.0;  // = cry[c]
    }
    const int gic = gi < g.ni ? gi : g.ni - 1;  // (a tile may stick out of the storage: the metric column must exist)
    EdgeSpacing sp;
    if (EY) sp = load_edge_spacing<1>(g, m.dya, gic, near_s, near_n, sj8);
    if constexpr (EY && canon_y) ppm_run_canon<MORD == 8 ? 6 : MORD, RF, kCanonY ? AHI_Y : 0>(Q, cc, near_s && grp == 0, near_n && grp == GY - 1, sp, out);
    else ppm_run_p<MORD, EY, RF>(Q, cc, gj0, g.js, g.je, sp, out);
#pragma unroll
    for (int f = 0; f < RF; ++f) vyf[f] = 0.0;
    auto faces = [&](auto F0_, auto F1_) {  // (see the x loop)
      constexpr int F0 = decltype(F0_)::value, F1 = decltype(F1_)::value, NF = F1 - F0;
      unsigned cf[NF];
      double yu[NF], si[NF], ms[NF + 1], w0[NF], w1[NF], w2[NF], w3[NF], wa[NF];
#pragma unroll
      for (int t = 0; t < NF; ++t) {
        const int f = F0 + t;
        cf[t] = kb8 + (calc[f] ? OFF2(gi, gj0 + f) : OFF2(g.is, g.js));
        yu[t] = LD(yunit, cf[t]);
        si[t] = syin[jj0 + f][ii + 3];
      }
      if (DMODE == 2) {
#pragma unroll
        for (int t = 0; t <= NF; ++t) {
          const int gjm = gj0 + F0 - 1 + t;
          ms[t] = LD(dp.mass, kb8 + OFF2(gic, gjm < g.nj ? gjm : g.nj - 1));
        }
      }
      if (EPI == 0) {
        if (dp.u_upd) {
#pragma unroll
          for (int t = 0; t < NF; ++t) {
            w0[t] = LD(dp.u_upd, cf[t]);
            w1[t] = LD(m.dx, cf[t] - kb8);
            w2[t] = LD(dp.ke, cf[t]);
            w3[t] = LD(dp.ke, cf[t] + E8);
          }
        }
        if (dp.accy) {
#pragma unroll
          for (int t = 0; t < NF; ++t) wa[t] = LD(dp.accy, cf[t]);
        }
      }
#pragma unroll
      for (int t = 0; t < NF; ++t) {
        const int f = F0 + t;
        if (calc[f]) {
          const unsigned c = cf[t];
          double v = 0.5 * (out[f] + si[t]) * yu[t];
          if (DMODE == 0 && EPI == 0) ST(dp.fy2o, c) = dvy[f];
          if (DMODE == 1) v = v + dvy[f];
          if (DMODE == 2) v = v + 0.5 * damp * (ms[t] + ms[t + 1]) * dvy[f];
          if (EPI == 0) {
            if (dp.u_upd) {  // u_from_ke (d_sw.py:406-420)
              ST(dp.u_out ? dp.u_out : dp.u_upd, c) = w0[t] * w1[t] + w2[t] - w3[t] + v;
            } else {
              ST(fy, c) = v;
            }
            if (dp.accy) ST(dp.accy, c) = wa[t] + v;
          }
          vyf[f] = v;
        }
      }
    };
    if (EPI == 0) {  // batches of FV_EPI0_BATCH faces (three of them at RF = 5: 114 spilled VGPRs with two larger ones)
      constexpr int B = FV_EPI0_BATCH;
      faces(std::integral_constant<int, 0>{}, std::integral_constant<int, (B < RF ? B : RF)>{});
      if constexpr (B < RF) faces(std::integral_constant<int, B>{}, std::integral_constant<int, (2 * B < RF ? 2 * B : RF)>{});
      if constexpr (2 * B < RF) faces(std::integral_constant<int, 2 * B>{}, std::integral_constant<int, (3 * B < RF ? 3 * B : RF)>{});
      if constexpr (3 * B < RF) faces(std::integral_constant<int, 3 * B>{}, std::integral_constant<int, (4 * B < RF ? 4 * B : RF)>{});
      if constexpr (4 * B < RF) faces(std::integral_constant<int, 4 * B>{}, std::integral_constant<int, RF>{});
    } else {
      faces(std::integral_constant<int, 0>{}, std::integral_constant<int, RF>{});
    }
  }
  if (EPI > 0) {
    // epilogue: put the face fluxes of the tile (both sides of every cell) into LDS, then update the cells
    constexpr int AXP = TI + 2, AYP = TI + 1;
    double* ax = &L.syin[0][0];          // [TJ][AXP]
    double* ay = ax + TJ * AXP;          // [TJ + 1][AYP]
    double* ax2 = ay + (TJ + 1) * AYP;   // damping fluxes (EPI == 2, 3)
    double* ay2 = ax2 + TJ * AXP;
    static_assert(EPI == 0 || 2 * (TJ * AXP + (TJ + 1) * AYP) <= FvLds<DMODE, EPI>::kSweep + FvLds<DMODE, EPI>::kPad, "epilogue scratch");
    __syncthreads();
    if (x5_on) {
      const int jj = x5_row, grp = x5_grp;
#pragma unroll
      for (int f = 0; f < RF; ++f) {
        if (grp * RF + f <= TI) {
          ax[jj * AXP + grp * RF + f] = vxf[f];
          if (EPI >= 2) ax2[jj * AXP + grp * RF + f] = dvx[f];
        }
      }
    }
    if (y5_on) {
      const int grp = y5_grp, ii = y5_col;
#pragma unroll
      for (int f = 0; f < RF; ++f) {
        if (grp * RF + f <= TJ) {
          ay[(grp * RF + f) * AYP + ii] = vyf[f];
          if (EPI >= 2) ay2[(grp * RF + f) * AYP + ii] = dvy[f];
        }
      }
    }
    __syncthreads();
    constexpr int NEC = (TI * TJ + 255) / 256;
    if (EPI == 3) {
      // apply_height_fluxes (updatedzd.py:70-126): the advected height from the transport's own fluxes over the area the cell
      // has after the step, plus the damping increment -- written to dp.qout (the height field itself is still being read by
      // neighbouring tiles); same expressions, same order as the stand-alone kernel it replaces
      double ar_[NEC], qv_[NEC], x0_[NEC], x1_[NEC], y0_[NEC], y1_[NEC];
#pragma unroll
      for (int t = 0; t < NEC; ++t) {
        const int e = tid + 256 * t;
        const int jj = e / TI, ii = e - jj * TI;
        const int gi = i0 + ii, gj = j0 + jj;
        const bool ok = e < TI * TJ && !((EX && gi > g.ie) || (EY && gj > g.je));
        const unsigned c2 = ok ? OFF2(gi, gj) : OFF2(g.is, g.js);
        ar_[t] = LD(m.area, c2);
        qv_[t] = LD(q, kb8 + c2);
        x0_[t] = LD(xfx, kb8 + c2);
        x1_[t] = LD(xfx, kb8 + c2 + E8);
        y0_[t] = LD(yfx, kb8 + c2);
        y1_[t] = LD(yfx, kb8 + c2 + sj8);
      }
#pragma unroll
      for (int t = 0; t < NEC; ++t) {
        const int e = tid + 256 * t;
        const int jj = e / TI, ii = e - jj * TI;
        const int gi = i0 + ii, gj = j0 + jj;
        if (e >= TI * TJ || (EX && gi > g.ie) || (EY && gj > g.je)) continue;
        const unsigned c = kb8 + OFF2(gi, gj);
        const double area = ar_[t];
        const double area_after = (area + x0_[t] - x1_[t]) + (area + y0_[t] - y1_[t]) - area;
        const double adv = (qv_[t] * area + ax[jj * AXP + ii] - ax[jj * AXP + ii + 1] + ay[jj * AYP + ii] - ay[(jj + 1) * AYP + ii]) /
                           area_after;
        ST(dp.qout, c) = adv + (ax2[jj * AXP + ii] - ax2[jj * AXP + ii + 1] + ay2[jj * AYP + ii] - ay2[(jj + 1) * AYP + ii]) / area;
      }
      return;
    }
    double ra_[NEC], qv_[NEC], am_[NEC];
#pragma unroll
    for (int t = 0; t < NEC; ++t) {  // all loads first (see stage 0)
      const int e = tid + 256 * t;
      const int jj = e / TI, ii = e - jj * TI;
      const int gi = i0 + ii, gj = j0 + jj;
      const bool ok = e < TI * TJ && !((EX && gi > g.ie) || (EY && gj > g.je));
      const unsigned c2 = ok ? OFF2(gi, gj) : OFF2(g.is, g.js);
      ra_[t] = LD(m.rarea, c2);
      qv_[t] = LD(q, kb8 + c2);  // (the LDS copy of q has become q_j)
      am_[t] = LD(dp.amass, kb8 + c2);
    }
#pragma unroll
    for (int t = 0; t < NEC; ++t) {
      const int e = tid + 256 * t;
      const int jj = e / TI, ii = e - jj * TI;
      const int gi = i0 + ii, gj = j0 + jj;
      if (e >= TI * TJ || (EX && gi > g.ie) || (EY && gj > g.je)) continue;
      const unsigned c = kb8 + OFF2(gi, gj);
      const double ra = ra_[t];
      const double qv = qv_[t];
      ST(dp.qout, c) = qv * am_[t] + (ax[jj * AXP + ii] - ax[jj * AXP + ii + 1] + ay[jj * AYP + ii] - ay[(jj + 1) * AYP + ii]) * ra;
      if (EPI == 2) {
        double hs = 0.0;
        if (dp.damp_w_k[k] > 1e-5) {
          const double dd8 = dp.ke_bg_k[k] * fabs(dp.dt);
          const double d = (ax2[jj * AXP + ii] - ax2[jj * AXP + ii + 1] + ay2[jj * AYP + ii] - ay2[(jj + 1) * AYP + ii]) * ra;
          ST(dp.dw, c) = d;
          hs = dd8 - d * (qv + 0.5 * d);
        }
        ST(dp.heat_s, c) = hs;
        ST(dp.diss_est, c) = hs;
      }
    }
  }
}

#ifndef FV_WAVES
#define FV_WAVES 4
#endif
// CANON: every tile edge coincides with a tile boundary of the workgroup tiling (fv_canonical_tiling), so that the one-sided
// PPM forms sit at compile-time positions of the first / last run of a row or column (common.h ppm_run_canon)
template <int MORD, int DMODE, int EPI, bool CANON>
__global__ void __launch_bounds__(256, FV_WAVES) k_fvtp2d(Geo g, FvMet m, const real* __restrict__ q,
                                                       const real* __restrict__ crx, const real* __restrict__ cry,
                                                       const real* __restrict__ xfx, const real* __restrict__ yfx,
                                                       real* __restrict__ fx, real* __restrict__ fy,
                                                       const real* __restrict__ xunit, const real* __restrict__ yunit,
                                                       FvDamp dp) {
  // x-interfaces evaluated: i0 .. i0+TI (their al's reach one further each way); special forms at is-1 .. is+1 and
  // ie .. ie+2
  __shared__ FvLds<DMODE, EPI> L;
  const FvTile wg = fv_tile_of_workgroup();
  const int i0 = g.is + wg.bx * TI, j0 = g.js + wg.by * TJ;
  // (ord 8: the special CELLS are s-1 .. s+1 and e-1 .. e+1; the cells evaluated are i0-1 .. i0+TI -- the same test)
  const bool ex = (i0 - 1 <= g.is + 1) || (i0 + TI + 1 >= g.ie);
  const bool ey = (j0 - 1 <= g.js + 1) || (j0 + TJ + 1 >= g.je);
  if (ex && ey) fvtp2d_tile<MORD, true, true, DMODE, EPI, CANON>(L, g, m, q, crx, cry, xfx, yfx, fx, fy, xunit, yunit, dp, wg);
  else if (ex) fvtp2d_tile<MORD, true, false, DMODE, EPI, CANON>(L, g, m, q, crx, cry, xfx, yfx, fx, fy, xunit, yunit, dp, wg);
  else if (ey) fvtp2d_tile<MORD, false, true, DMODE, EPI, CANON>(L, g, m, q, crx, cry, xfx, yfx, fx, fy, xunit, yunit, dp, wg);
  else fvtp2d_tile<MORD, false, false, DMODE, EPI, CANON>(L, g, m, q, crx, cry, xfx, yfx, fx, fy, xunit, yunit, dp, wg);
}

// d_sw's three mass-weighted scalars -- w (damping fluxes + heat_diss: DMODE 0, EPI 2), q_con and pt (mass-weighted damping:
// DMODE 2, EPI 1) -- in ONE launch (round 3): the grid is three tile planes high, a workgroup's plane says which scalar it
// transports.  The three read the same Courant numbers, area fluxes and mass fluxes level by level (a level belongs to one XCD:
// tile_of_workgroup), their workgroups fill each other's partly filled rounds, and nothing has to fork to side streams and
// join again before k_finish_scalars.  The instances are the stand-alone kernels' (same code, same results).
template <int MORD, bool CANON>
__global__ void __launch_bounds__(256, FV_WAVES) k_fvtp2d_scalars3(Geo g, FvMet m, const real* __restrict__ crx, const real* __restrict__ cry,
                                                                const real* __restrict__ xfx, const real* __restrict__ yfx,
                                                                const real* __restrict__ xunit, const real* __restrict__ yunit,
                                                                const real* __restrict__ q0, FvDamp dp0, const real* __restrict__ q1,
                                                                FvDamp dp1, const real* __restrict__ q2, FvDamp dp2, int gy) {
  constexpr size_t kBytes = sizeof(FvLds<0, 2>) > sizeof(FvLds<2, 1>) ? sizeof(FvLds<0, 2>) : sizeof(FvLds<2, 1>);
  __shared__ double raw[(kBytes + 7) / 8];
  FvTile wg = fv_tile_of_workgroup();
  const int which = wg.by / gy;  // block-uniform
  wg.by -= which * gy;
  const int i0 = g.is + wg.bx * TI, j0 = g.js + wg.by * TJ;
  const bool ex = (i0 - 1 <= g.is + 1) || (i0 + TI + 1 >= g.ie);
  const bool ey = (j0 - 1 <= g.js + 1) || (j0 + TJ + 1 >= g.je);
  if (which == 0) {
    FvLds<0, 2>& L = *reinterpret_cast<FvLds<0, 2>*>(raw);
    if (ex && ey) fvtp2d_tile<MORD, true, true, 0, 2, CANON>(L, g, m, q0, crx, cry, xfx, yfx, nullptr, nullptr, xunit, yunit, dp0, wg);
    else if (ex) fvtp2d_tile<MORD, true, false, 0, 2, CANON>(L, g, m, q0, crx, cry, xfx, yfx, nullptr, nullptr, xunit, yunit, dp0, wg);
    else if (ey) fvtp2d_tile<MORD, false, true, 0, 2, CANON>(L, g, m, q0, crx, cry, xfx, yfx, nullptr, nullptr, xunit, yunit, dp0, wg);
    else fvtp2d_tile<MORD, false, false, 0, 2, CANON>(L, g, m, q0, crx, cry, xfx, yfx, nullptr, nullptr, xunit, yunit, dp0, wg);
  } else {
    FvLds<2, 1>& L = *reinterpret_cast<FvLds<2, 1>*>(raw);
    const real* q = which == 1 ? q1 : q2;
    const FvDamp& dp = which == 1 ? dp1 : dp2;
    if (ex && ey) fvtp2d_tile<MORD, true, true, 2, 1, CANON>(L, g, m, q, crx, cry, xfx, yfx, nullptr, nullptr, xunit, yunit, dp, wg);
    else if (ex) fvtp2d_tile<MORD, true, false, 2, 1, CANON>(L, g, m, q, crx, cry, xfx, yfx, nullptr, nullptr, xunit, yunit, dp, wg);
    else if (ey) fvtp2d_tile<MORD, false, true, 2, 1, CANON>(L, g, m, q, crx, cry, xfx, yfx, nullptr, nullptr, xunit, yunit, dp, wg);
    else fvtp2d_tile<MORD, false, false, 2, 1, CANON>(L, g, m, q, crx, cry, xfx, yfx, nullptr, nullptr, xunit, yunit, dp, wg);
  }
}

// whether every edge tile of the launch holds its edge at the canonical place: N a multiple of the tile in both directions,
// at least two tiles each way (so that no run holds both ends)
static inline bool fv_canonical_tiling(const Geo& g) {
  return g.n % TI == 0 && g.n % TJ == 0 && g.n >= 2 * TI && g.n >= 2 * TJ;
}

// w, q_con and pt of d_sw in one launch (k_fvtp2d_scalars3); ord 6 for all three, damping orders <= 2
int launch_transport_scalars3(const Geo& g, const Met& m, const real* w, const real* q_con, const real* pt, const real* crx,
                              const real* cry, const real* xfx, const real* yfx, const real* xmf, const real* ymf, int nlev,
                              const FvDamp& dpw, const FvDamp& dpq, const FvDamp& dpt, hipStream_t st) {
  if (dpw.nmax > 2 || dpq.nmax > 2 || dpt.nmax > 2) return PACE_ERR_UNSUPPORTED;
  const int gy = (g.n + TJ - 1) / TJ;
  const dim3 grid((g.n + TI - 1) / TI, 3 * gy, nlev);
  if (fv_canonical_tiling(g))
    hipLaunchKernelGGL((k_fvtp2d_scalars3<6, true>), grid, dim3(256), 0, st, g, fv_met(m), crx, cry, xfx, yfx, xmf, ymf, w, dpw, q_con, dpq, pt,
                       dpt, gy);
  else
    hipLaunchKernelGGL((k_fvtp2d_scalars3<6, false>), grid, dim3(256), 0, st, g, fv_met(m), crx, cry, xfx, yfx, xmf, ymf, w, dpw, q_con, dpq, pt,
                       dpt, gy);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}

#define FV_LAUNCH(D, E)                                                                                                          \
  do {                                                                                                                           \
    if (MORD != 8 && fv_canonical_tiling(g))                                                                                     \
      hipLaunchKernelGGL((k_fvtp2d<MORD, D, E, MORD != 8>), grid, block, 0, st, g, fv_met(m), q, crx, cry, xfx, yfx, fx, fy, \
                         xu, yu, dp);                                                                                            \
    else                                                                                                                         \
      hipLaunchKernelGGL((k_fvtp2d<MORD, D, E, false>), grid, block, 0, st, g, fv_met(m), q, crx, cry, xfx, yfx, fx, fy, xu, \
                         yu, dp);                                                                                                \
  } while (0)

template <int MORD>
static int launch_mode(int dmode, int epi, dim3 grid, hipStream_t st, const Geo& g, const Met& m, const real* q,
                       const real* crx, const real* cry, const real* xfx, const real* yfx, real* fx, real* fy,
                       const real* xu, const real* yu, const FvDamp& dp) {
  const dim3 block(256);
  if (epi == 0) {
    switch (dmode) {
      case 0: FV_LAUNCH(0, 0); break;
      case 1: FV_LAUNCH(1, 0); break;
      case 2: FV_LAUNCH(2, 0); break;
      default: FV_LAUNCH(-1, 0); break;
    }
  } else if (epi == 1) {
    switch (dmode) {
      case 1: FV_LAUNCH(1, 1); break;
      case 2: FV_LAUNCH(2, 1); break;
      case -1: FV_LAUNCH(-1, 1); break;
      default: return PACE_ERR_UNSUPPORTED;
    }
  } else if (epi == 2) {
    if (dmode != 0) return PACE_ERR_UNSUPPORTED;
    FV_LAUNCH(0, 2);
  } else {
    if (dmode != 0 || epi != 3) return PACE_ERR_UNSUPPORTED;
    FV_LAUNCH(0, 3);
  }
  return PACE_OK;
}

// The general launcher.  dmode -1: transport only; otherwise the del-n damping of q is fused (see FvDamp).  epi 0:
// fluxes are written; 1 / 2: the flux-form update of the cell (and heat_diss) is written instead.
int launch_transport(const Geo& g, const Met& m, const real* q, const real* crx, const real* cry, const real* xfx,
                     const real* yfx, real* fx, real* fy, const real* xmf, const real* ymf, int hord, int nlev,
                     int dmode, int epi, const FvDamp& dp, hipStream_t st) {
  if (dmode >= 0 && dp.nmax > 2) return PACE_ERR_UNSUPPORTED;
  const dim3 grid((g.n + TI - 1) / TI, (g.n + TJ - 1) / TJ, nlev);
  const real* xu = xmf ? xmf : xfx;
  const real* yu = ymf ? ymf : yfx;
  int rc = launch_transport_lean(g, m, q, crx, cry, xfx, yfx, fx, fy, xu, yu, hord, nlev, dmode, epi, dp, st);
  if (rc != PACE_ERR_UNSUPPORTED) return rc;
  if (hord == 5) rc = launch_mode<5>(dmode, epi, grid, st, g, m, q, crx, cry, xfx, yfx, fx, fy, xu, yu, dp);
  else if (hord == 6) rc = launch_mode<6>(dmode, epi, grid, st, g, m, q, crx, cry, xfx, yfx, fx, fy, xu, yu, dp);
  else if (hord == 8 && dmode == -1 && epi == 0) {  // monotone PPM: tracer advection (plain transport only)
    hipLaunchKernelGGL((k_fvtp2d<8, -1, 0, false>), grid, dim3(256), 0, st, g, fv_met(m), q, crx, cry, xfx, yfx, fx, fy, xu, yu, dp);
    rc = PACE_OK;
  }
  else return PACE_ERR_UNSUPPORTED;
  if (rc) return rc;
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}

int launch_fvtp2d(const Geo& g, const Met& m, const real* q, const real* crx, const real* cry,
                  const real* xfx, const real* yfx, real* fx, real* fy, const real* xmf,
                  const real* ymf, int hord, int nlev, hipStream_t st) {
  FvDamp dp{};
  return launch_transport(g, m, q, crx, cry, xfx, yfx, fx, fy, xmf, ymf, hord, nlev, -1, 0, dp, st);
}
