// FiniteVolumeTransport for the production tilings -- the lean form of k_fvtp2d.hip (same arithmetic, same bits) -- and the
// scalar phase of d_sw (delp, w, q_con, pt) as ONE kernel built from the same stages.
// (device code of k_fvt.hip; tools/census/ instantiates single tile variants of it for the instruction census)
// Reference: fv3core/pace/fv3core/stencils/fvtp2d.py:262-345, xppm.py:148-181 / yppm.py, delnflux.py:1005-1261,
// d_sw.py:63-201,331-350.
//
// What the measurements of round 4 say (tools/ubench, profiles/r04_ubench.txt):
//  * ONE wave issues at most one instruction every ~8 cycles, whatever its kind; four waves per SIMD share an fp64 pipe of
//    ~3.6 cycles per wave-instruction, a 32-bit pipe of ~1.9, a scalar unit of ~1 per cycle per CU.  The general kernel walks
//    ~4 500 instructions per wave of which ~730 are fp64 arithmetic; the rest is position arithmetic, predicates (exec-mask
//    bookkeeping on the scalar unit), selects and spilled-SGPR traffic.
//  * Removing 60 % of those instructions (this file's single-scalar kernel: ~1 900 per wave) did NOT shorten the launch: the
//    transport launches move their L2-miss bytes at 3.5 - 4 TB/s, three quarters of what a device copy reaches on this chip
//    (4.8 - 5.0 TB/s) -- they are bound by the bytes they really move (1.5 x the algorithmic ones per launch, and every scalar
//    of d_sw moves the Courant numbers, area fluxes, mass fluxes and the mass again), not by the algorithmic bytes.
// So: (1) the lean stages below -- fixed thread places, no predication (only tilings that put every tile edge on a
// workgroup-tile boundary: N a multiple of TI and TJ, every footprint inside the storage), three barrier intervals per
// scalar instead of seven, 16-byte footprint loads -- and (2) fvt_scalars_tile: one workgroup takes its tile through delp,
// w, q_con and pt back to back, the mass fluxes of delp stay in the registers of the threads that need them as unit fluxes,
// the new delp divides the other three in the same workgroup (apply_pt_delp_fluxes / adjust_w_and_qcon, d_sw.py:148-201,
// 331-350: no k_finish_scalars pass, no flux-form intermediates), results go to buffers of their own (the neighbouring tiles
// still read the old values).  Other tilings, ord 8 and the float32 build take k_fvtp2d.hip.
#pragma once
#include <type_traits>

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

#if (FV_TI % 4 == 0) && (FV_TJ % 4 == 0) && (FV_TI >= 8) && (FV_TJ >= 8) && (DN_TI == FV_TI) && \
    (DN_TJ == FV_TJ) && ((FV_TI + 6) * (FV_TJ / 4) <= 256) && ((FV_TJ + 6) * (FV_TI / 4) <= 256)
#define FVT_AVAILABLE 1
#else
#define FVT_AVAILABLE 0
#endif

#ifndef FVT_NS
#define FVT_NS fvt  // (k_fvt16.hip compiles the same source for a second tile shape in a namespace of its own)
#endif
#if FVT_AVAILABLE
namespace FVT_NS {

constexpr int TI = FV_TI, TJ = FV_TJ;
constexpr int C = 4;                       // cells per run
constexpr int NF = C + 1;                  // interfaces per run
constexpr int GXN = TI / C, GYN = TJ / C;  // runs per row / column of the tile
constexpr int QW = TI + 6, QH = TJ + 6;    // footprint
constexpr int P = QW + 1;                  // pitch of sq, sqi and the damping planes
constexpr int PJ = TI + 1;                 // pitch of sqj and ax
constexpr int NYO = TI * GYN;              // threads 0 .. NYO-1: the y-runs of the tile's own columns (they run the outer y sweep too)
constexpr int NYH = 6 * GYN;               // then the y-runs of the six halo columns
constexpr int NEC = (TI * TJ + 255) / 256;  // cells per thread in the cell update
static_assert(DW == QW && DH == QH && DWP == P, "the damping core shares the footprint");
static_assert(GXN >= 2 && GYN >= 2, "the first and the last run of a row / column must be different runs");

constexpr int RB = (int)sizeof(real);  // bytes per stored element (8; 4 in the float32-storage build: arithmetic is double in both)

struct alignas(16) D2 {
  double x, y;
};
struct alignas(2 * sizeof(real)) RealPair {  // two neighbouring elements as they are stored: one 16-byte (float32: 8-byte) load
  real x, y;
};

struct FvtLds {
  double pad0[P];     // (the damping runs read one row above / below a plane without clamping)
  double sq[QH * P];  // q on the footprint, corners copied in y; never modified
  union {
    struct {
      double sqi[TJ * P];        // q advected in y, tile rows x footprint columns
      double sqj[QH * PJ];       // q advected in x, footprint rows x tile columns
      double ax[TJ * PJ];        // final x-face fluxes (cell update)
      double ay[(TJ + 1) * TI];  // final y-face fluxes
    } s;
    double scratch[3 * QH * P];  // damping: iterate, del6_v, del6_u
  } u;
  double pad1[P];
  double sqc[9];  // the tile's corner block with corners copied in x (corner tiles)
};

// The layout of the scalar-phase kernel since round 6: the damping's iterate and its two metric planes have places of their own
// instead of aliasing the sweeps' arrays.  del6_v / del6_u are then staged ONCE per tile (they used to be fetched again at the start
// of every pass -- 0.35 GB of a launch's 1.06 GB, L2 misses every time: 50 us of 420, experiment x25 of round 5), and the iterate
// lives until the outer sweep has formed its fluxes, so the damping's face values are read from it where they are used (no
// per-thread store of them: 20 KB) and the damping's passes share barrier intervals with the sweeps, which read q only.
struct FvtLdsRes {
  double pad0[P];          // (the damping runs read one row above / below a plane without clamping)
  double sq[QH * P];       // q on the footprint, corners copied in y; never modified
  double it[QH * P];       // the damping's iterate
  double sdv[QH * P];      // del6_v, del6_u on the footprint
  double sdu[QH * P];
  double pad1[P];
  struct {
    struct {
      double sqi[TJ * P];        // q advected in y, tile rows x footprint columns (w: dw on the tile; the winds: fy)
      double sqj[QH * PJ];       // q advected in x, footprint rows x tile columns (w: its heating term; the winds: fx)
      double ax[TJ * PJ];        // inner, then final x-face fluxes
      double ay[(TJ + 1) * TI];  // ... y-face fluxes
    } s;
  } u;
  double sqc[9];  // the tile's corner block with corners copied in x (corner tiles)
  // edge tiles: the four A-grid spacings of the one-sided PPM forms along the tile's edge, per footprint row (x sweeps) and column
  // (y sweeps): staged once per tile -- every sweep of every pass used to load them from memory right in front of its arithmetic
  double spx[QH * 4], spy[QW * 4];
  int fxs[64 * 7];  // corner tiles: the places of the damping's corner cells (FvtTile fx_at, fx_src), derived once per tile
};

// order pin for the instruction scheduler (register pressure: operands that are only needed after a PPM run are loaded after it)
#ifdef PACE_EMU
#define FVT_FENCE()
#define FVT_LAUNDER(x) (x)
#else
#define FVT_FENCE() __builtin_amdgcn_sched_barrier(0)
// the same value through an empty asm: what is derived from it is derived AGAIN, not kept in registers from the last time
// (the thread places of the scalar-phase kernel, re-derived per scalar instead of living through the whole kernel)
__device__ __forceinline__ int fvt_launder(int x) {
  asm volatile("" : "+v"(x));
  return x;
}
#define FVT_LAUNDER(x) fvt_launder(x)
#endif
// ... of the scalar-phase kernel (two workgroups per CU: 80 KB each): the transport's arrays, the mass on the tile and one cell
// around it (the mass-weighted damping of q_con and pt reads it at the faces, the cell update at the cells), and ten doubles
// per thread that would otherwise be spilled registers (the damping fluxes of the thread's faces while the inner sweeps run)
constexpr int MP = TI + 3;  // pitch of the mass tile
struct FvtLdsScalars {
  FvtLds t;
  double mass[(TJ + 2) * MP];
  double priv[2 * NF][256];
  double newmass[TI * TJ];  // (512-thread form: the new delp of the tile's cells; the 256-thread form keeps it in registers)
  double heat[TI * TJ];     // (512-thread form with the winds: w's heating term, heat_diss -> the dissipative heating at the end)
};
static_assert(2 * sizeof(FvtLdsScalars) <= 160 * 1024, "two workgroups per CU");

#define LDG(p, off) (*(const real*)((const char*)(p) + (off)))
#define STG(p, off) (*(real*)((char*)(p) + (off)))
// *p += v for an accumulator that exactly one thread of the launch touches: the hardware's fp64 atomic add without a return value
// -- one instruction, nothing to wait for -- instead of load, add, store (one rounding either way: the same bits).  float32
// storage: the sum is formed in double and rounded once, as the load-add-store form does.
__device__ __forceinline__ void fvt_accumulate(real* p, double v) {
#if defined(PACE_EMU) || defined(PACE_REAL_FLOAT)
  *p = (real)((double)*p + v);
#else
  (void)unsafeAtomicAdd(p, v);
#endif
}

// A-grid spacings of the one-sided PPM forms (the provider interface of common.h ppm_patch_edge): four values along the sweep
// axis, at the start OR at the end of the tile -- a workgroup tile holds at most one edge per axis.  `fixed` is the byte offset of
// the run's column (y sweeps) or row (x sweeps), `step` the byte stride along the axis; loaded only by the runs that hold the edge.
struct FvtSpacing {
  double d[4];
  __device__ __forceinline__ double operator()(bool, int idx, int) const { return d[idx]; }
};
__device__ __forceinline__ FvtSpacing fvt_spacing(const real* d, unsigned fixed, int step, int s, int e, bool lo, bool hi) {
  FvtSpacing sp;
#pragma unroll
  for (int t = 0; t < 4; ++t) sp.d[t] = 0.0;
  if (lo || hi) {
    const int first = lo ? s - 2 : e - 1;
#pragma unroll
    for (int t = 0; t < 4; ++t) sp.d[t] = LDG(d, fixed + (unsigned)((first + t) * step));
  }
  return sp;
}

// one run: the NF fluxes of the interfaces between cells Q[2 .. NF+2] (Q[u] = cell first_interface - 3 + u)
template <int MORD, bool EDGE>
__device__ __forceinline__ void fvt_run(const double* Q, const double* c, bool lane_lo, bool lane_hi, const FvtSpacing& sp, double* out) {
  if constexpr (EDGE) ppm_run_canon<MORD, NF, C>(Q, c, lane_lo, lane_hi, sp, out);
  else ppm_run_p<MORD, false, NF>(Q, c, 0, 0, 0, sp, out);
}

// ---- del-n damping, lean form (delnflux.py:1209-1261; the arithmetic of delnflux_core.h deln_run, same bits) ----------------
// A thread owns the column run (column dc, rows dr0 .. dr0 + DN_RC - 1) of the footprint.  One pass = the divergence of the
// fluxes of the iterate, every cell of the footprint evaluated as if it were an interior cell: the outermost ring of the
// footprint (after the first pass; two rings after the second) holds garbage that no tile face ever reads -- the reference
// shrinks its compute domain by one cell per pass for the same reason -- so no cell is predicated and no index is clamped
// (rows -1 and QH of a plane are the neighbouring arrays / pad rows of FvtLds).
template <bool FIRST, int RC = DN_RC>
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
  constexpr int H = (RC + 1) / 2;
  chunk(std::integral_constant<int, 0>{}, std::integral_constant<int, H>{});
#ifndef PACE_EMU
  __builtin_amdgcn_sched_barrier(0);
#endif
  chunk(std::integral_constant<int, H>{}, std::integral_constant<int, RC>{});
}

// a footprint-sized plane of a field, 16 bytes per lane
template <int NT>
struct FvtPiecesT {
  static constexpr int HW = QW / 2;             // 16-byte pieces per row
  static constexpr int RPP = NT / HW;           // rows per pass
  static constexpr int NP = (QH + RPP - 1) / RPP;
  int lc, lr0, row[NP];
  unsigned off[NP];
  // whether piece p of this thread is a piece of its own (not a clamped repeat of another thread's)
  __device__ __forceinline__ bool own(int p) const { return lr0 < RPP && lr0 + RPP * p < QH; }
  __device__ __forceinline__ void init(int tid, int ilo, int jlo, int sj8) {
    int lr = tid / HW;
    lc = tid - lr * HW;
    lr0 = lr;
    if (lr >= RPP) lr = RPP - 1;  // (the spare threads repeat the last piece: same values to the same place)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      row[p] = lr + RPP * p;
      if (row[p] >= QH) row[p] = QH - 1;
      off[p] = (unsigned)((jlo + row[p]) * sj8 + (ilo + 2 * lc) * RB);
    }
  }
  __device__ __forceinline__ void load(const real* src, D2* v) const {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const RealPair r = *(const RealPair*)((const char*)src + off[p]);
      v[p].x = r.x, v[p].y = r.y;
    }
  }
  __device__ __forceinline__ void store(double* plane, const D2* v) const {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      plane[row[p] * P + 2 * lc] = v[p].x;
      plane[row[p] * P + 2 * lc + 1] = v[p].y;
    }
  }
};
using FvtPieces = FvtPiecesT<256>;

// The stages of one tile.  EX / EY: the tile holds a west or east / south or north edge of the cubed-sphere tile (block-uniform).
// NT = 256: every thread owns one y-run AND one x-run.  NT = 512 (the scalar-phase kernel): the first four waves own the x-runs,
// the last four the y-runs -- half the persistent state per thread, twice the waves per SIMD.
template <int MORD, bool EX, bool EY, int NT = 256, class LT = FvtLds>
struct FvtTile {
  static constexpr bool RES = std::is_same<LT, FvtLdsRes>::value;  // the damping's planes have places of their own
  static constexpr bool RC = EX && EY;  // the footprint reaches a corner of the halo: the corner copies apply
  static constexpr bool SPLIT = NT == 512;
  static_assert(NT == 256 || NT == 512, "thread places of one or two roles");
  static constexpr int DRC = SPLIT ? 3 : DN_RC;       // cells per column run of the damping
  static constexpr int DNR = (QH + DRC - 1) / DRC;   // runs per column
  static_assert(QW * DNR <= NT, "one column run per thread");
  static constexpr int NCU = (TI * TJ + NT - 1) / NT;  // cells per thread in the cell update
  using Pieces = FvtPiecesT<NT>;
  LT& L;
  const Geo& g;
  const FvMet& m;
  int tid, i0, j0, ilo, jlo, sj8, k;
  unsigned kb8;  // byte offset of the level
  bool west, east, south, north;
  // the thread's places: one y-run (column ycol of the footprint, rows C*yg ..), one x-run (row xrow, columns C*xg ..)
  int ycol, yg, xr, xg, xrow;
  bool y_on, y_outer, x_on, x_outer;
  int ybase, xbase;     // sq / sqi / damping planes: element (row C*yg, column ycol); sq: element (row xrow, column C*xg)
  unsigned yoff, xoff;  // byte offsets in a level: the y-run's first interface (ilo + ycol, j0 + C*yg), the x-run's (i0 + C*xg, jlo + xrow)
  Pieces pc;
  // the thread's column run of the damping: column dc, rows dr0 .. dr0 + DRC - 1 of the footprint
  int dc, dr0, dbase;
  bool dn_on;
  double dra[DRC];
  // corner tiles: a cell whose damping stencil reaches into a corner region of the halo (delnflux_core.h deln_affected), one
  // candidate per thread of the first wave: its place in the plane (or -1) and the places its six operands come from once the
  // corner copies are applied -- X(i-1), X(i), X(i+1) with copy_corners_x, Y(j-1), Y(j), Y(j+1) with copy_corners_y
  // (delnflux.py:1009-1047); -1: outside the footprint, the value is zero
  int fx_at, fx_src[6];
  double *plane, *sdv, *sdu;

  // fx_cached (resident form, corner tiles): the corner cells' places are read from L.fxs (put there by publish_corner_places)
  __device__ __forceinline__ FvtTile(LT& L_, const Geo& g_, const FvMet& m_, int bx, int by, int k_, int tid_, bool fx_cached = false)
      : L(L_), g(g_), m(m_) {
    tid = tid_;
    k = k_;
    i0 = g.is + bx * TI, j0 = g.js + by * TJ;
    ilo = i0 - 3, jlo = j0 - 3;
    sj8 = g.sj * RB;
    kb8 = (unsigned)((long)k * g.sk * RB);
    // which edges of the cubed-sphere tile this workgroup tile holds (one per axis at most: >= 2 tiles each way)
    west = EX && bx == 0, east = EX && !west;
    south = EY && by == 0, north = EY && !south;
    const bool xrole = !SPLIT || tid < 256, yrole = !SPLIT || tid >= 256;  // (wave-uniform)
    const int tr = SPLIT ? (tid & 255) : tid;                               // the thread's number within its role
    y_on = yrole;
    if (tr < NYO) {
      yg = tr / TI;
      ycol = 3 + (tr - yg * TI);
    } else {
      const int r = tr - NYO;
      y_on = yrole && r < NYH;
      yg = y_on ? r / 6 : 0;
      const int h = y_on ? r - yg * 6 : 0;
      ycol = h < 3 ? h : TI + h;
    }
    y_outer = yrole && tr < NYO;
    if (GXN == 8) {
      // Sixteen lanes that the LDS serves together take 4 rows x 4 runs, not 2 rows x 8 runs: a run starts every C = 4 values, so
      // runs g and g + 4 of a row are 16 values apart -- the same pair of banks for 8-byte accesses (two-way conflicts on every
      // read of the x-runs: 30 M conflict cycles per launch, a third of the LDS's busy time; SQ_LDS_BANK_CONFLICT).  With the
      // pitches of these planes (41, 33, 35: odd and != 1 mod 4 apart) four consecutive rows shift by distinct amounts mod 16.
      const int l = tr & 63, w = tr >> 6;
      xr = (w << 3) + ((l >> 5) << 2) + ((l & 15) >> 2);
      xg = (((l >> 4) & 1) << 2) + (l & 3);
    } else {
      xr = tr / GXN, xg = tr - xr * GXN;
    }
    x_on = xrole && xr < QH, x_outer = xrole && xr < TJ;
    xrow = xr + 3;  // the tile's own rows first: footprint rows 3 .. TJ+2, then TJ+3 .. QH-1, then 0 .. 2
    if (xrow >= QH) xrow -= QH;
    if (!x_on) xrow = 0;
    ybase = (C * yg) * P + ycol;
    xbase = xrow * P + C * xg;
    yoff = (unsigned)((j0 + C * yg) * sj8 + (ilo + ycol) * RB);
    xoff = (unsigned)((jlo + xrow) * sj8 + (i0 + C * xg) * RB);
    pc.init(tid, ilo, jlo, sj8);
    const int dr = tid / QW;
    dc = tid - dr * QW;
    dn_on = dr < DNR;
    dr0 = dn_on ? dr * DRC : 0;
    dbase = dr0 * P + dc;
    if constexpr (RES) {
      plane = L.it, sdv = L.sdv, sdu = L.sdu;
    } else {
      plane = L.u.scratch;
      sdv = plane + QH * P;
      sdu = sdv + QH * P;
    }
    fx_at = -1;
    if (RC && tid < 64 && fx_cached) {
      if constexpr (RES) {
        fx_at = L.fxs[tid * 7];
#pragma unroll
        for (int d = 0; d < 6; ++d) fx_src[d] = L.fxs[tid * 7 + 1 + d];
      }
    } else if (RC && tid < 64) {
      const int q = tid >> 4, a = tid & 3, b = (tid >> 2) & 3;
      const int gi = (q & 1) ? g.ie + a : g.is - 3 + a;
      const int gj = (q & 2) ? g.je + b : g.js - 3 + b;
      const int ci = gi - ilo, cj = gj - jlo;
      const bool valid = ci >= 1 && ci <= QW - 2 && cj >= 1 && cj <= QH - 2 && gi >= 1 && gi + 1 < g.ni && gj >= 1 && gj + 1 < g.nj;
      if (ci >= 0 && ci < QW && cj >= 0 && cj < QH && deln_affected(g, gi, gj) && valid) {
        fx_at = cj * P + ci;
        auto place = [&](int i, int j) {
          const int la = i - ilo, lb = j - jlo;
          return (la >= 0 && la < QW && lb >= 0 && lb < QH) ? lb * P + la : -1;
        };
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          int xi = gi - 1 + d, xj = gj, yi = gi, yj = gj - 1 + d;
          remap_agrid_x(g, xi, xj);
          remap_agrid_y(g, yi, yj);
          fx_src[d] = place(xi, xj);
          fx_src[3 + d] = place(yi, yj);
        }
      }
    }
  }
  // the divergence at that cell (the expressions of delnflux_core.h deln_corner_fix)
  template <bool FIRST>
  __device__ __forceinline__ double corner_fix(const double* src, double d0) const {
    auto rd = [&](int o) { return o >= 0 ? (FIRST ? d0 * src[o] : src[o]) : 0.0; };
    auto sgn = [&](double x) { return FIRST ? x : -x; };
    const double ra = LDG(m.rarea, (unsigned)((jlo + fx_at / P) * sj8 + (ilo + fx_at % P) * RB));
    const double xm = rd(fx_src[0]), xc = rd(fx_src[1]), xp = rd(fx_src[2]), ym = rd(fx_src[3]), yc = rd(fx_src[4]), yp = rd(fx_src[5]);
    const double fw = sgn(sdv[fx_at] * (xm - xc));
    const double fe = sgn(sdv[fx_at + 1] * (xc - xp));
    const double fs = sgn(sdu[fx_at] * (ym - yc));
    const double fn = sgn(sdu[fx_at + P] * (yc - yp));
    return (fw - fe + fs - fn) * ra;
  }

  // ---- stage 0: the footprint of q (level base applied), 16 bytes per lane (rows start 16-byte aligned: ilo = TI * bx, sj even).
  // No barrier here.
  // halo_out (edge tiles of the scalar phase): the footprint's cells outside the compute domain are copied there as they are, so
  // that the output buffer ends up with the halo the input has (the reference updates its fields in place).
  __device__ __forceinline__ void load_footprint(const real* __restrict__ q, real* __restrict__ halo_out = nullptr) {
    D2 v[Pieces::NP];
    pc.load(q, v);
    place_footprint(q, v, halo_out);
  }
  // the two halves of it: the loads (the scalar-phase kernel issues those of all four scalars at its start: one memory latency
  // per tile instead of one per scalar) ...
  __device__ __forceinline__ void fetch_footprint(const real* __restrict__ q, D2* v) const { pc.load(q, v); }
  // ... and the pieces' way into the LDS (+ the halo copy and the corner values, which read q again)
  // DEFER (corner tiles of the resident form): the corner block's cells are filled by corners_from_lds() behind the caller's next
  // barrier, from the footprint itself -- every source of a corner copy lies in the tile's own footprint -- instead of by dependent
  // global loads here (a corner tile's footprint stage took 10 k cycles per pass against 1.3 k of an interior one)
  template <bool DEFER = false>
  __device__ __forceinline__ void place_footprint(const real* __restrict__ q, const D2* v, real* __restrict__ halo_out = nullptr) {
    if ((EX || EY) && halo_out) {
#pragma unroll
      for (int p = 0; p < Pieces::NP; ++p) {
        const int gj = jlo + pc.row[p];
        const bool rowout = gj < g.js || gj > g.je;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int gi = ilo + 2 * pc.lc + e;
          // (a corner tile's 3 x 3 corner block is written below: what the transport's in-place corner copies leave there)
          const bool corner_cell = RC && rowout && (gi < g.is || gi > g.ie);
          if (pc.own(p) && (rowout || gi < g.is || gi > g.ie) && !corner_cell) STG(halo_out, pc.off[p] + (unsigned)(RB * e)) = e == 0 ? v[p].x : v[p].y;
        }
      }
    }
    pc.store(L.sq, v);
    if (RC && !DEFER) {  // corner tile: the corner block holds the values copy_corners_y puts there (corners.py:367-425); the thread that
               // stored a piece overwrites its corner cells (same thread, same address: program order)
#pragma unroll
      for (int p = 0; p < Pieces::NP; ++p) {
        const int gj = jlo + pc.row[p];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int gi = ilo + 2 * pc.lc + e;
          if ((gi < g.is || gi > g.ie) && (gj < g.js || gj > g.je)) {
            int ri = gi, rj = gj;
            remap_agrid_y(g, ri, rj);
            L.sq[pc.row[p] * P + 2 * pc.lc + e] = LDG(q, (unsigned)(rj * sj8 + ri * RB));
          }
        }
      }
      if (tid < 9) {  // ... and what copy_corners_x puts there, for the x sweeps
        const int b = tid / 3, a = tid - b * 3;
        const int di = (west ? g.is - 3 : g.ie + 1) + a, dj = (south ? g.js - 3 : g.je + 1) + b;
        int ri = di, rj = dj;
        remap_agrid_x(g, ri, rj);
        const real cv = LDG(q, (unsigned)(rj * sj8 + ri * RB));
        L.sqc[tid] = cv;
        // ... which is also what the reference LEAVES in q's corner block: the y copy, then the x copy, in place (fvtp2d.py:262-345),
        // nothing after it -- TranslateD_SW compares the scalars over the whole storage (translate_d_sw.py:36-65)
        if (halo_out) STG(halo_out, (unsigned)(dj * sj8 + di * RB)) = cv;
      }
    }
  }
  // del6_v, del6_u on the footprint -> sdv, sdu; rarea of the thread's damping run
  __device__ __forceinline__ void stage_damping_metrics() {
    stage_damping_planes();
    load_damping_rarea();
  }
  __device__ __forceinline__ void stage_damping_planes() {
    D2 mv[Pieces::NP], mu[Pieces::NP];
    pc.load(m.del6_v, mv);
    pc.load(m.del6_u, mu);
    pc.store(sdv, mv);
    pc.store(sdu, mu);
  }
  __device__ __forceinline__ void load_damping_rarea() {
#pragma unroll
    for (int t = 0; t < DRC; ++t) {
      int row = dr0 + t;
      if (QH % DRC != 0 && row >= QH) row = QH - 1;
      dra[t] = LDG(m.rarea, (unsigned)((jlo + row) * sj8 + (ilo + dc) * RB));
    }
  }
  // the transported scalar is q + add2d (absolute vorticity), the damped one was q: every thread adds to the pieces it loaded.
  // Ends with a barrier.
  __device__ __forceinline__ void add_2d(const real* __restrict__ add2d) {
    D2 v[Pieces::NP];
    pc.load(add2d, v);
#pragma unroll
    for (int p = 0; p < Pieces::NP; ++p) {
      if (pc.own(p)) {  // (each piece once: the clamped repeats would add twice)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          double add = e == 0 ? v[p].x : v[p].y;
          if (RC) {
            int gi = ilo + 2 * pc.lc + e, gj = jlo + pc.row[p];
            if ((gi < g.is || gi > g.ie) && (gj < g.js || gj > g.je)) {
              remap_agrid_y(g, gi, gj);
              add = LDG(add2d, (unsigned)(gj * sj8 + gi * RB));
            }
          }
          L.sq[pc.row[p] * P + 2 * pc.lc + e] = L.sq[pc.row[p] * P + 2 * pc.lc + e] + add;
        }
      }
    }
    if (RC && tid < 9) {
      const int b = tid / 3, a = tid - b * 3;
      int ri = (west ? g.is - 3 : g.ie + 1) + a, rj = (south ? g.js - 3 : g.je + 1) + b;
      remap_agrid_x(g, ri, rj);
      L.sqc[tid] = L.sqc[tid] + LDG(add2d, (unsigned)(rj * sj8 + ri * RB));
    }
    __syncthreads();
  }

  __device__ __forceinline__ void publish_corner_places() {
    if constexpr (RES) {
      if (RC && tid < 64) {
        L.fxs[tid * 7] = fx_at;
#pragma unroll
        for (int d = 0; d < 6; ++d) L.fxs[tid * 7 + 1 + d] = fx_src[d];
      }
    }
  }
  // edge tiles of the resident form: the spacings of the one-sided PPM forms into L.spx / L.spy (the caller's next barrier publishes)
  __device__ __forceinline__ void stage_spacings() {
    if constexpr (RES) {
      if (EX && tid < QH) {
        const int first = west ? g.is - 2 : g.ie - 1;
#pragma unroll
        for (int t = 0; t < 4; ++t) L.spx[tid * 4 + t] = LDG(m.dxa, (unsigned)((jlo + tid) * sj8 + (first + t) * RB));
      }
      if (EY && tid >= 64 && tid < 64 + QW) {
        const int cidx = tid - 64, first = south ? g.js - 2 : g.je - 1;
#pragma unroll
        for (int t = 0; t < 4; ++t) L.spy[cidx * 4 + t] = LDG(m.dya, (unsigned)((first + t) * sj8 + (ilo + cidx) * RB));
      }
    }
  }
  // Corner tile, after the barrier that publishes the footprint: the corner block <- what copy_corners_y puts there, sqc <- what
  // copy_corners_x puts there (and, if given, to the output buffer's corner block: what the reference leaves in q), all from the
  // footprint in LDS.  The caller passes a barrier before anything reads the block.
  __device__ __forceinline__ void corners_from_lds(real* __restrict__ halo_out) {
    if (!RC) return;
    const int t = tid < 9 ? tid : tid - 16;
    if (t < 0 || t >= 9) return;
    const int b = t / 3, a = t - b * 3;
    const int di = (west ? g.is - 3 : g.ie + 1) + a, dj = (south ? g.js - 3 : g.je + 1) + b;
    int ri = di, rj = dj;
    if (tid < 9) {
      remap_agrid_y(g, ri, rj);
      L.sq[(dj - jlo) * P + (di - ilo)] = L.sq[(rj - jlo) * P + (ri - ilo)];
    } else {
      remap_agrid_x(g, ri, rj);
      const double cv = L.sq[(rj - jlo) * P + (ri - ilo)];
      L.sqc[t] = cv;
      if (halo_out) STG(halo_out, (unsigned)(dj * sj8 + di * RB)) = (real)cv;
    }
  }
  // add_2d for the resident form: every thread adds to the pieces it loaded, a corner tile's corner cells are copied again from
  // their (now absolute) sources -- the same sums, formed once.  Ends with a barrier.
  __device__ __forceinline__ void add_2d_lds(const real* __restrict__ add2d) {
    D2 v[Pieces::NP];
    pc.load(add2d, v);
#pragma unroll
    for (int p = 0; p < Pieces::NP; ++p) {
      if (pc.own(p)) {  // (each piece once: the clamped repeats would add twice)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int gi = ilo + 2 * pc.lc + e, gj = jlo + pc.row[p];
          if (RC && (gi < g.is || gi > g.ie) && (gj < g.js || gj > g.je)) continue;
          L.sq[pc.row[p] * P + 2 * pc.lc + e] = L.sq[pc.row[p] * P + 2 * pc.lc + e] + (e == 0 ? v[p].x : v[p].y);
        }
      }
    }
    if (RC) {
      __syncthreads();
      corners_from_lds(nullptr);
    }
    __syncthreads();
  }

  // ---- the del-n damping of q on the footprint (delnflux.py:1209-1261).  Call after the barrier that follows stage 0; on return
  // (after a barrier if a pass ran) the last iterate is in `last` (q itself if no pass ran: `first`).
  struct Damped {
    const double* last;
    double d0;
    bool first;
  };
  __device__ __forceinline__ Damped damp(double d0, bool hi_order, int nmax) {
    const int iters = hi_order ? nmax : 0;
    for (int it = 0; it < iters; ++it) {
      double res[DRC];
      double fix = 0.0;
      if (dn_on) {
        if (it == 0) fvt_deln_run<true, DRC>(L.sq + dbase, sdv + dbase, sdu + dbase, dra, d0, res);
        else fvt_deln_run<false, DRC>(plane + dbase, sdv + dbase, sdu + dbase, dra, d0, res);
      }
      if (RC && fx_at >= 0) {  // corner tile: the cells whose stencil reaches into a corner region, with the corner copies
        if (it == 0) fix = corner_fix<true>(L.sq, d0);
        else fix = corner_fix<false>(plane, d0);
      }
      if (it > 0) __syncthreads();  // (in place: everyone has read the iterate)
      if (dn_on) {
#pragma unroll
        for (int t = 0; t < DRC; ++t)
          if (QH % DRC == 0 || dr0 + t < QH) plane[dbase + t * P] = res[t];
      }
      if (RC) {
        __syncthreads();  // (the runs have written those cells as if they were interior cells: overwrite)
        if (fx_at >= 0) plane[fx_at] = fix;
      }
      __syncthreads();
    }
    return Damped{iters == 0 ? L.sq : plane, d0, iters == 0};
  }
  // the damping flux through a face from the last iterate (first: the first flux evaluation, of d0 * q)
  __device__ __forceinline__ static double face(const Damped& D, double metric, double a, double b) {
    return D.first ? metric * (D.d0 * a - D.d0 * b) : -(metric * (a - b));
  }
  __device__ __forceinline__ void damping_faces(const Damped& D, double* dvx, double* dvy) const {
    if (x_outer) {
#pragma unroll
      for (int f = 0; f < NF; ++f) dvx[f] = face(D, sdv[xbase + f + 3], D.last[xbase + f + 2], D.last[xbase + f + 3]);
    }
    if (y_outer) {
#pragma unroll
      for (int f = 0; f < NF; ++f) dvy[f] = face(D, sdu[ybase + (f + 3) * P], D.last[ybase + (f + 2) * P], D.last[ybase + (f + 3) * P]);
    }
  }
  // ---- the same damping in pieces, for a caller that runs other stages between its barriers (the resident layout) ----
  // one pass on the thread's column run (and a corner tile's fixed cell), result in registers: from q (FIRST) or from the iterate
  template <bool FIRST>
  __device__ __forceinline__ void deln_compute(double d0, double* res, double& fix) const {
    fix = 0.0;
    if (dn_on) fvt_deln_run<FIRST, DRC>((FIRST ? L.sq : plane) + dbase, sdv + dbase, sdu + dbase, dra, d0, res);
    if (RC && fx_at >= 0) fix = FIRST ? corner_fix<true>(L.sq, d0) : corner_fix<false>(plane, d0);
  }
  // ... put down into the iterate (the caller has passed a barrier since the last read of it).  On a corner tile the cells whose
  // stencil reaches a corner region are overwritten behind a barrier; the caller's next barrier publishes everything.
  __device__ __forceinline__ void deln_store(const double* res, double fix) {
    if (dn_on) {
#pragma unroll
      for (int t = 0; t < DRC; ++t)
        if (QH % DRC == 0 || dr0 + t < QH) plane[dbase + t * P] = res[t];
    }
    if (RC) {
      __syncthreads();
      if (fx_at >= 0) plane[fx_at] = fix;
    }
  }
  // heat_diss (d_sw.py:63-103): dw = divergence of the damping fluxes / area -- one more divergence of the iterate on this thread's
  // column run (the same expression, in the same order, as the flux-difference form of the general kernel) --, heat_source and
  // diss_est from it; stored for the tile's cells of the run (level offsets applied by the caller)
  // dw_tile / heat_tile: if given, dw / heat_s go to these LDS tiles ([TJ][TI]; read back by the caller after a barrier) instead
  // of memory
  __device__ __forceinline__ void heat_diss(const Damped& D, real* __restrict__ dw, real* __restrict__ heat_s,
                                            real* __restrict__ diss_est, bool on, double dd8, double* dw_tile = nullptr,
                                            double* heat_tile = nullptr) {
    if (!dn_on) return;
    double res[DRC];
    if (D.first) fvt_deln_run<true, DRC>(L.sq + dbase, sdv + dbase, sdu + dbase, dra, D.d0, res);
    else fvt_deln_run<false, DRC>(plane + dbase, sdv + dbase, sdu + dbase, dra, D.d0, res);
    if (dc >= 3 && dc < TI + 3) {
#pragma unroll
      for (int t = 0; t < DRC; ++t) {
        const int jj = dr0 + t;
        if (jj >= 3 && jj < TJ + 3) {
          const unsigned c = (unsigned)((jlo + jj) * sj8 + (ilo + dc) * RB);
          double hs = 0.0;
          if (on) {
            const double d = res[t];
            const double qv = L.sq[jj * P + dc];
            if (dw_tile) dw_tile[(jj - 3) * TI + dc - 3] = d;
            else STG(dw, c) = d;
            hs = dd8 - d * (qv + 0.5 * d);
          }
          if (heat_tile) heat_tile[(jj - 3) * TI + dc - 3] = hs;
          else STG(heat_s, c) = hs;
          STG(diss_est, c) = hs;
        }
      }
    }
  }

  // ---- stage I: the inner sweeps and the advected fields (fvtp2d.py:34-77); pointers with the level applied.  Ends with a barrier.
  __device__ __forceinline__ void inner(const real* __restrict__ crx, const real* __restrict__ cry, const real* __restrict__ xfx,
                                        const real* __restrict__ yfx, double* si_x, double* si_y, double* cx, double* cy) {
    if (y_on) {  // YPiecewiseParabolic on the run's five interfaces, then q_i of the four cells between them
      double Q[NF + 5], yf[NF], ar[C];
#pragma unroll
      for (int u = 0; u < NF + 5; ++u) Q[u] = L.sq[ybase + u * P];
#pragma unroll
      for (int f = 0; f < NF; ++f) cy[f] = LDG(cry, yoff + (unsigned)(f * sj8));
      FvtSpacing sp;
      if (EY) sp = fvt_spacing(m.dya, (unsigned)((ilo + ycol) * RB), sj8, g.js, g.je, south && yg == 0, north && yg == GYN - 1);
      fvt_run<MORD, EY>(Q, cy, south && yg == 0, north && yg == GYN - 1, sp, si_y);
      FVT_FENCE();
#pragma unroll
      for (int f = 0; f < NF; ++f) yf[f] = LDG(yfx, yoff + (unsigned)(f * sj8));
#pragma unroll
      for (int t = 0; t < C; ++t) ar[t] = LDG(m.area, yoff + (unsigned)(t * sj8));
#pragma unroll
      for (int t = 0; t < C; ++t)
        L.u.s.sqi[ybase + t * P] = (Q[t + 3] * ar[t] + yf[t] * si_y[t] - yf[t + 1] * si_y[t + 1]) / (ar[t] + yf[t] - yf[t + 1]);
    }
    FVT_FENCE();
    if (x_on) {  // XPiecewiseParabolic, then q_j
      double Q[NF + 5], xf[NF], ar[C];
#pragma unroll
      for (int u = 0; u < NF + 5; ++u) Q[u] = L.sq[xbase + u];
      if (RC) {  // halo rows of a corner tile: the three corner columns hold the x-direction copies
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
      for (int f = 0; f < NF; ++f) cx[f] = LDG(crx, xoff + (unsigned)(f * RB));
      FvtSpacing sp;
      if (EX) sp = fvt_spacing(m.dxa, (unsigned)((jlo + xrow) * sj8), RB, g.is, g.ie, west && xg == 0, east && xg == GXN - 1);
      fvt_run<MORD, EX>(Q, cx, west && xg == 0, east && xg == GXN - 1, sp, si_x);
      FVT_FENCE();
#pragma unroll
      for (int f = 0; f < NF; ++f) xf[f] = LDG(xfx, xoff + (unsigned)(f * RB));
#pragma unroll
      for (int t = 0; t < C; ++t) ar[t] = LDG(m.area, xoff + (unsigned)(t * RB));
#pragma unroll
      for (int t = 0; t < C; ++t)
        L.u.s.sqj[xrow * PJ + C * xg + t] = (Q[t + 3] * ar[t] + xf[t] * si_x[t] - xf[t + 1] * si_x[t + 1]) / (ar[t] + xf[t] - xf[t + 1]);
    }
    __syncthreads();
  }

  // The operands of the inner sweeps that do not depend on the scalar: the Courant numbers and area fluxes of the thread's five
  // y- and five x-faces, the areas of its cells (pointers with the level applied).  The scalar-phase kernel loads them once.
  struct SweepOperands {
    double cx[NF], cy[NF], xf[NF], yf[NF], arx[C], ary[C];
  };
  __device__ __forceinline__ void load_sweep_operands(const real* __restrict__ crx, const real* __restrict__ cry,
                                                      const real* __restrict__ xfx, const real* __restrict__ yfx, SweepOperands& o) const {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      o.cy[f] = LDG(cry, yoff + (unsigned)(f * sj8));
      o.yf[f] = LDG(yfx, yoff + (unsigned)(f * sj8));
      o.cx[f] = LDG(crx, xoff + (unsigned)(f * RB));
      o.xf[f] = LDG(xfx, xoff + (unsigned)(f * RB));
    }
#pragma unroll
    for (int t = 0; t < C; ++t) {
      o.ary[t] = LDG(m.area, yoff + (unsigned)(t * sj8));
      o.arx[t] = LDG(m.area, xoff + (unsigned)(t * RB));
    }
  }
  // stage I with the operands given.  Ends with a barrier.
  template <bool DO_Y = true, bool DO_X = true>
  __device__ __forceinline__ void inner_with(const SweepOperands& o, double* si_x, double* si_y) {
    if (DO_Y && y_on) {
      double Q[NF + 5];
#pragma unroll
      for (int u = 0; u < NF + 5; ++u) Q[u] = L.sq[ybase + u * P];
      FvtSpacing sp;
      if (EY) sp = fvt_spacing(m.dya, (unsigned)((ilo + ycol) * RB), sj8, g.js, g.je, south && yg == 0, north && yg == GYN - 1);
      fvt_run<MORD, EY>(Q, o.cy, south && yg == 0, north && yg == GYN - 1, sp, si_y);
#pragma unroll
      for (int t = 0; t < C; ++t)
        L.u.s.sqi[ybase + t * P] = (Q[t + 3] * o.ary[t] + o.yf[t] * si_y[t] - o.yf[t + 1] * si_y[t + 1]) / (o.ary[t] + o.yf[t] - o.yf[t + 1]);
    }
    if (DO_X && x_on) {
      double Q[NF + 5];
#pragma unroll
      for (int u = 0; u < NF + 5; ++u) Q[u] = L.sq[xbase + u];
      if (RC) {  // halo rows of a corner tile: the three corner columns hold the x-direction copies
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
      FvtSpacing sp;
      if (EX) sp = fvt_spacing(m.dxa, (unsigned)((jlo + xrow) * sj8), RB, g.is, g.ie, west && xg == 0, east && xg == GXN - 1);
      fvt_run<MORD, EX>(Q, o.cx, west && xg == 0, east && xg == GXN - 1, sp, si_x);
#pragma unroll
      for (int t = 0; t < C; ++t)
        L.u.s.sqj[xrow * PJ + C * xg + t] = (Q[t + 3] * o.arx[t] + o.xf[t] * si_x[t] - o.xf[t + 1] * si_x[t + 1]) / (o.arx[t] + o.xf[t] - o.xf[t + 1]);
    }
    __syncthreads();
  }

  // ---- stage II: the outer sweeps (fvtp2d.py:80-119): the mean advected value through the run's five faces, 0.5 * (outer + inner).
  // x: on q_i, tile row xr (= footprint row xrow), faces i0 + C*xg + f.  Only for x_outer threads.
  __device__ __forceinline__ void outer_x(const double* cx, const double* si_x, double* mean) const {
    double Q[NF + 5], out[NF];
#pragma unroll
    for (int u = 0; u < NF + 5; ++u) Q[u] = L.u.s.sqi[xr * P + C * xg + u];
    FvtSpacing sp;
    if (EX) sp = fvt_spacing(m.dxa, (unsigned)((jlo + xrow) * sj8), RB, g.is, g.ie, west && xg == 0, east && xg == GXN - 1);
    fvt_run<MORD, EX>(Q, cx, west && xg == 0, east && xg == GXN - 1, sp, out);
#pragma unroll
    for (int f = 0; f < NF; ++f) mean[f] = 0.5 * (out[f] + si_x[f]);
  }
  // y: on q_j, tile column ycol - 3, faces j0 + C*yg + f.  Only for y_outer threads.
  __device__ __forceinline__ void outer_y(const double* cy, const double* si_y, double* mean) const {
    double Q[NF + 5], out[NF];
#pragma unroll
    for (int u = 0; u < NF + 5; ++u) Q[u] = L.u.s.sqj[(C * yg + u) * PJ + ycol - 3];
    FvtSpacing sp;
    if (EY) sp = fvt_spacing(m.dya, (unsigned)((ilo + ycol) * RB), sj8, g.js, g.je, south && yg == 0, north && yg == GYN - 1);
    fvt_run<MORD, EY>(Q, cy, south && yg == 0, north && yg == GYN - 1, sp, out);
#pragma unroll
    for (int f = 0; f < NF; ++f) mean[f] = 0.5 * (out[f] + si_y[f]);
  }
  __device__ __forceinline__ void put_ax(const double* v) const {
#pragma unroll
    for (int f = 0; f < NF; ++f) L.u.s.ax[xr * PJ + C * xg + f] = v[f];
  }
  __device__ __forceinline__ void put_ay(const double* v) const {
#pragma unroll
    for (int f = 0; f < NF; ++f) L.u.s.ay[(C * yg + f) * TI + ycol - 3] = v[f];
  }
  // the cells of the cell update: NEC per thread, lanes along i
  __device__ __forceinline__ void cell_places(int* jj, int* ii, unsigned* c2) const {
#pragma unroll
    for (int t = 0; t < NCU; ++t) {
      int e = tid + NT * t;
      if (e >= TI * TJ) e = TI * TJ - 1;  // (spare lanes repeat the last cell)
      jj[t] = e / TI, ii[t] = e - jj[t] * TI;
      c2[t] = (unsigned)((j0 + jj[t]) * sj8 + (i0 + ii[t]) * RB);
    }
  }
  // q * mass + the flux increment (apply_fluxes, d_sw.py:122-145) of cell t from the fluxes in ax / ay
  __device__ __forceinline__ double flux_form(int jj, int ii, double am, double ra) const {
    const double qv = L.sq[(jj + 3) * P + ii + 3];
    const double* ax = L.u.s.ax + jj * PJ + ii;
    const double* ay = L.u.s.ay + jj * TI + ii;
    return qv * am + (ax[0] - ax[1] + ay[0] - ay[TI]) * ra;
  }
  __device__ __forceinline__ double flux_increment(int jj, int ii, double ra) const {
    const double* ax = L.u.s.ax + jj * PJ + ii;
    const double* ay = L.u.s.ay + jj * TI + ii;
    return (ax[0] - ax[1] + ay[0] - ay[TI]) * ra;
  }
};

// DMODE: -1 transport only; 0 damping fluxes -> dp.fx2o / fy2o (+ dp.add2d, u / v update: the vorticity call of d_sw);
// 1 damping fluxes added to the transport fluxes; 2 added mass-weighted; 3 damping of q -> dw / heat_s / diss_est only (w).
// EPI 0: fluxes stored (or accumulated / turned into winds); 1: flux-form update of the cell stored; 3 (with DMODE 0): the
// height update of updatedzd stored (apply_height_fluxes, updatedzd.py:70-126).
#ifndef FVT_ARRIVE
#define FVT_ARRIVE(n)  // (tools/census/fvt_prof.hip: when the first x-run wave and the first y-run wave reach barrier n of a pass)
#endif
#ifndef FVT_STAMP
#define FVT_STAMP(n)  // (tools/census/fvt_prof.hip: shader-clock stamps of one workgroup per level)
#endif
template <int MORD, int DMODE, int EPI, bool EX, bool EY>
__device__ __forceinline__ void fvt_tile(FvtLds& L, const Geo& g, const FvMet& m, const real* __restrict__ q,
                                         const real* __restrict__ crx, const real* __restrict__ cry,
                                         const real* __restrict__ xfx, const real* __restrict__ yfx, real* __restrict__ fx,
                                         real* __restrict__ fy, const real* __restrict__ xunit,
                                         const real* __restrict__ yunit, const FvDamp& dp, int bx, int by, int k) {
  FvtTile<MORD, EX, EY> T(L, g, m, bx, by, k, (int)threadIdx.x);
  const long kb = (long)k * g.sk;
  const unsigned kb8 = T.kb8;
  const int sj8 = T.sj8;
  FVT_STAMP(20);
  T.load_footprint(q + kb);
  if (DMODE >= 0) T.stage_damping_metrics();
  __syncthreads();
  FVT_STAMP(21);

  double dvx[NF], dvy[NF];
  double damp = 0.0;
  if (DMODE >= 0) {
    damp = dp.damp_k[k];
    const auto D = T.damp(dp.mass_given ? 1.0 : damp, dp.nord_k[k] > 0.0, dp.nmax);
    if (DMODE != 3) T.damping_faces(D, dvx, dvy);
    else T.heat_diss(D, dp.dw + kb, dp.heat_s + kb, dp.diss_est + kb, dp.damp_w_k[k] > 1e-5, dp.ke_bg_k[k] * fabs(dp.dt));
    __syncthreads();  // the sweeps overwrite the damping planes
  }
  if (DMODE == 0 && dp.add2d) T.add_2d(dp.add2d);
  FVT_STAMP(22);

  double si_y[NF], si_x[NF], cy[NF], cx[NF];
  T.inner(crx + kb, cry + kb, xfx + kb, yfx + kb, si_x, si_y, cx, cy);
  FVT_STAMP(23);

  if (T.x_outer) {
    double v[NF], xu[NF], ms[NF + 1];
#pragma unroll
    for (int f = 0; f < NF; ++f) xu[f] = LDG(xunit, kb8 + T.xoff + (unsigned)(f * RB));
    if (DMODE == 2) {
#pragma unroll
      for (int t = 0; t <= NF; ++t) ms[t] = LDG(dp.mass, kb8 + T.xoff + (unsigned)((t - 1) * RB));
    }
    T.outer_x(cx, si_x, v);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      v[f] = v[f] * xu[f];
      if (DMODE == 1) v[f] = v[f] + dvx[f];
      if (DMODE == 2) v[f] = v[f] + 0.5 * damp * (ms[f] + ms[f + 1]) * dvx[f];
    }
    if (EPI == 0) {
      // a face is stored by the run it opens; the last face of the row (ie + 1) by the last run of the east-edge tile
      const bool last = T.east && T.xg == GXN - 1;
      double w0[NF], w1[NF], w2[NF], w3[NF], wa[NF];
      if (dp.v_upd) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          w0[f] = LDG(dp.v_upd, kb8 + T.xoff + (unsigned)(f * RB));
          w1[f] = LDG(m.dy, T.xoff + (unsigned)(f * RB));
          w2[f] = LDG(dp.ke, kb8 + T.xoff + (unsigned)(f * RB));
          w3[f] = LDG(dp.ke, kb8 + T.xoff + (unsigned)(f * RB + sj8));
        }
      }
      if (dp.accx) {
#pragma unroll
        for (int f = 0; f < NF; ++f) wa[f] = LDG(dp.accx, kb8 + T.xoff + (unsigned)(f * RB));
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        if (f < C || last) {
          const unsigned c = kb8 + T.xoff + (unsigned)(f * RB);
          if (DMODE == 0) STG(dp.fx2o, c) = dvx[f];
          if (dp.v_upd) STG(dp.v_out ? dp.v_out : dp.v_upd, c) = w0[f] * w1[f] + w2[f] - w3[f] - v[f];  // v_from_ke (d_sw.py:423-436)
          else STG(fx, c) = v[f];
          if (dp.accx) STG(dp.accx, c) = wa[f] + v[f];
        }
      }
    } else {
      T.put_ax(v);
    }
  }
  FVT_STAMP(24);
  if (T.y_outer) {
    double v[NF], yu[NF], ms[NF + 1];
#pragma unroll
    for (int f = 0; f < NF; ++f) yu[f] = LDG(yunit, kb8 + T.yoff + (unsigned)(f * sj8));
    if (DMODE == 2) {
#pragma unroll
      for (int t = 0; t <= NF; ++t) ms[t] = LDG(dp.mass, kb8 + T.yoff + (unsigned)((t - 1) * sj8));
    }
    T.outer_y(cy, si_y, v);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      v[f] = v[f] * yu[f];
      if (DMODE == 1) v[f] = v[f] + dvy[f];
      if (DMODE == 2) v[f] = v[f] + 0.5 * damp * (ms[f] + ms[f + 1]) * dvy[f];
    }
    if (EPI == 0) {
      const bool last = T.north && T.yg == GYN - 1;
      double w0[NF], w1[NF], w2[NF], w3[NF], wa[NF];
      if (dp.u_upd) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          w0[f] = LDG(dp.u_upd, kb8 + T.yoff + (unsigned)(f * sj8));
          w1[f] = LDG(m.dx, T.yoff + (unsigned)(f * sj8));
          w2[f] = LDG(dp.ke, kb8 + T.yoff + (unsigned)(f * sj8));
          w3[f] = LDG(dp.ke, kb8 + T.yoff + (unsigned)(f * sj8 + RB));
        }
      }
      if (dp.accy) {
#pragma unroll
        for (int f = 0; f < NF; ++f) wa[f] = LDG(dp.accy, kb8 + T.yoff + (unsigned)(f * sj8));
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        if (f < C || last) {
          const unsigned c = kb8 + T.yoff + (unsigned)(f * sj8);
          if (DMODE == 0) STG(dp.fy2o, c) = dvy[f];
          if (dp.u_upd) STG(dp.u_out ? dp.u_out : dp.u_upd, c) = w0[f] * w1[f] + w2[f] - w3[f] + v[f];  // u_from_ke (d_sw.py:406-420)
          else STG(fy, c) = v[f];
          if (dp.accy) STG(dp.accy, c) = wa[f] + v[f];
        }
      }
    } else {
      T.put_ay(v);
    }
  }
  FVT_STAMP(25);
  if (EPI == 3) {
    // apply_height_fluxes (updatedzd.py:70-126): the advected height from the transport's own fluxes over the area the cell has
    // after the step, plus the damping increment -- to dp.qout (the neighbouring tiles still read the height field); the same
    // expressions in the same order as k_fvtp2d.hip's epilogue.  The damping fluxes of the tile's faces go where the sweeps'
    // inputs were (sqi as [TJ][PJ], sqj as [TJ + 1][TI]) once every sweep has read them.
    static_assert(EPI != 3 || DMODE == 0, "the height epilogue takes the damping fluxes unmixed");
    __syncthreads();
    double* const ax2 = L.u.s.sqi;
    double* const ay2 = L.u.s.sqj;
    if (T.x_outer) {
#pragma unroll
      for (int f = 0; f < NF; ++f) ax2[T.xr * PJ + C * T.xg + f] = dvx[f];
    }
    if (T.y_outer) {
#pragma unroll
      for (int f = 0; f < NF; ++f) ay2[(C * T.yg + f) * TI + T.ycol - 3] = dvy[f];
    }
    __syncthreads();
    int jj[NEC], ii[NEC];
    unsigned c2[NEC];
    double ar[NEC], x0[NEC], x1[NEC], y0[NEC], y1[NEC];
    T.cell_places(jj, ii, c2);
#pragma unroll
    for (int t = 0; t < NEC; ++t) {
      ar[t] = LDG(m.area, c2[t]);
      x0[t] = LDG(xfx, kb8 + c2[t]);
      x1[t] = LDG(xfx, kb8 + c2[t] + (unsigned)RB);
      y0[t] = LDG(yfx, kb8 + c2[t]);
      y1[t] = LDG(yfx, kb8 + c2[t] + (unsigned)sj8);
    }
#pragma unroll
    for (int t = 0; t < NEC; ++t) {
      const double area = ar[t];
      const double qv = L.sq[(jj[t] + 3) * P + ii[t] + 3];
      const double* ax = L.u.s.ax + jj[t] * PJ + ii[t];
      const double* ay = L.u.s.ay + jj[t] * TI + ii[t];
      const double* bx = ax2 + jj[t] * PJ + ii[t];
      const double* by = ay2 + jj[t] * TI + ii[t];
      const double area_after = (area + x0[t] - x1[t]) + (area + y0[t] - y1[t]) - area;
      const double adv = (qv * area + ax[0] - ax[1] + ay[0] - ay[TI]) / area_after;
      STG(dp.qout, kb8 + c2[t]) = adv + (bx[0] - bx[1] + by[0] - by[TI]) / area;
    }
  }
  if (EPI == 1) {
    // apply_fluxes (d_sw.py:122-145): q * mass + the flux increment, one cell per lane, lanes along i
    __syncthreads();
    int jj[NEC], ii[NEC];
    unsigned c2[NEC];
    double ra[NEC], am[NEC];
    T.cell_places(jj, ii, c2);
#pragma unroll
    for (int t = 0; t < NEC; ++t) {
      ra[t] = LDG(m.rarea, c2[t]);
      am[t] = LDG(dp.amass, kb8 + c2[t]);
    }
#pragma unroll
    for (int t = 0; t < NEC; ++t) STG(dp.qout, kb8 + c2[t]) = T.flux_form(jj[t], ii[t], am[t], ra[t]);
  }
  FVT_STAMP(26);
}

// ---- the scalar phase of d_sw in one kernel -------------------------------------------------------------------------------
// delp: transport + del-n damping of the mass fluxes (FiniteVolumeTransport with DelnFlux, d_sw.py:1040-1052), mfx / mfy
// accumulated (flux_capacitor, :33-60), new delp (apply_pt_delp_fluxes, :148-201).  w: DelnFluxNoSG -> heat_diss (:63-103),
// transport with the mass fluxes, flux-form update / new delp + dw (adjust_w_and_qcon, :331-350).  q_con, pt: transport with
// the mass fluxes + mass-weighted damping, flux-form update / new delp.  All outputs to buffers of their own.
struct FvtScalars {
  // in the order delp, w, q_con, pt (and, with the winds, the relative vorticity): input (never written), output, damping factor /
  // order columns (device, dsw_prepare)
  const real* q[5];
  real* qout[4];
  const real* fac[5];
  const real* nord[5];
  int nmax[5];
  const real *crx, *cry, *xfx, *yfx;
  real *mfx, *mfy;  // accumulated mass fluxes
  real *dw;         // one workspace field
  real *heat_s, *diss_est;
  const real *damp_w, *ke_bg;
  double dt;
  // ---- the winds (512-thread form only; winds != 0): the vorticity transport, u / v from it and the kinetic energy, the
  // dissipative heating and the final winds as a fifth pass of the tile (d_sw.py:406-477,493-608) ----
  int winds, do_skeb, copy_wind_halo;
  int ke_plus_vort;          // ke is the plain kinetic energy: ke + vort_b (the divergence damping's increment) is formed here
  double d_con;
  const real *u, *v;         // the winds before d_sw (read at the tile's faces only)
  real *u_out, *v_out;       // the winds after it: buffers of their own (a tile reads the old wind on the face its neighbour writes)
  const real *ke, *vort_b;   // kinetic energy (+ divergence damping) and damped vorticity at the B-grid points
  real* heat_source;         // += the dissipative heating
  const real *damp_vt, *d_con_k;
  const real *fC, *rdx, *rdy, *rsin2, *cosa_s;  // metric fields FvMet does not carry
};


template <int MORD, bool EX, bool EY>
__device__ __forceinline__ void fvt_scalars_tile(FvtLdsScalars& LS, const Geo& g, const FvMet& m, const FvtScalars& S, int bx, int by, int k) {
  // Register budget: 256 VGPRs = TWO workgroups per CU.  What a thread needs again for the next scalar and nobody else needs --
  // the Courant numbers and area fluxes of its runs, the mass fluxes through its faces, the new mass of its cells -- stays in
  // registers; the mass itself sits in the LDS.  At the 128 registers of four workgroups per CU the same kernel spilled (every
  // spilled register is 8 MB of scratch traffic per launch on a kernel that runs at the speed of its L2 misses: 2.1 GB per
  // launch against 1.7 GB for the four separate launches), and re-reading the operands per scalar misses the L2 every time (an
  // XCD's 4 MB turn over in ~6 us, a scalar takes ~20).  Measured: DESIGN.md section 4.
  FvtLds& L = LS.t;
  FvtTile<MORD, EX, EY> T(L, g, m, bx, by, k, (int)threadIdx.x);
  const int tid = T.tid;
  const long kb = (long)k * g.sk;
  const int sj8 = T.sj8;
  const bool w_on = S.damp_w[k] > 1e-5;
  double cx[NF], cy[NF], xf[NF], yf[NF];  // Courant numbers and area fluxes of the thread's faces
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    cy[f] = LDG(S.cry + kb, T.yoff + (unsigned)(f * sj8));
    yf[f] = LDG(S.yfx + kb, T.yoff + (unsigned)(f * sj8));
    cx[f] = LDG(S.crx + kb, T.xoff + (unsigned)(f * RB));
    xf[f] = LDG(S.xfx + kb, T.xoff + (unsigned)(f * RB));
  }
  double mfx[NF], mfy[NF];  // the mass fluxes through them (unit fluxes of w, q_con, pt)
  double dn[NEC];           // the new mass of the thread's cells

#pragma unroll
  for (int s = 0; s < 4; ++s) {  // delp, w, q_con, pt
    FVT_STAMP(4 * s);
    const real* const q = S.q[s] + kb;
    real* const qout = S.qout[s] + kb;
    const bool is_delp = s == 0, is_w = s == 1;
    const bool mass_weighted = s >= 2;  // DelnFlux with mass (q_con, pt); delp: plain DelnFlux; w: DelnFluxNoSG -> heat_diss
    const double damp = S.fac[s][k];
    T.load_footprint(q, qout);
    T.stage_damping_metrics();
    __syncthreads();
    FVT_STAMP(4 * s + 1);
    if (is_delp) {  // the mass on the tile and one cell around it, from the footprint while it is there
      for (int e = tid; e < (TJ + 2) * (TI + 2); e += 256) {
        const int r = e / (TI + 2), c = e - r * (TI + 2);
        LS.mass[r * MP + c] = L.sq[(r + 2) * P + c + 2];
      }
    }
    {
      double dvx[NF], dvy[NF];
      const auto D = T.damp(mass_weighted ? 1.0 : damp, S.nord[s][k] > 0.0, S.nmax[s]);
      if (is_w) {
        T.heat_diss(D, S.dw + kb, S.heat_s + kb, S.diss_est + kb, w_on, S.ke_bg[k] * fabs(S.dt));
      } else {
        T.damping_faces(D, dvx, dvy);
        if (T.x_outer) {
#pragma unroll
          for (int f = 0; f < NF; ++f) LS.priv[f][tid] = dvx[f];
        }
        if (T.y_outer) {
#pragma unroll
          for (int f = 0; f < NF; ++f) LS.priv[NF + f][tid] = dvy[f];
        }
      }
      __syncthreads();  // the sweeps overwrite the damping planes
    }
    FVT_STAMP(4 * s + 2);
    double si_y[NF], si_x[NF];
    {
      typename FvtTile<MORD, EX, EY>::SweepOperands O;
#pragma unroll
      for (int f = 0; f < NF; ++f) O.cx[f] = cx[f], O.cy[f] = cy[f], O.xf[f] = xf[f], O.yf[f] = yf[f];
#pragma unroll
      for (int t = 0; t < C; ++t) {
        O.ary[t] = LDG(m.area, T.yoff + (unsigned)(t * sj8));
        O.arx[t] = LDG(m.area, T.xoff + (unsigned)(t * RB));
      }
      T.inner_with(O, si_x, si_y);
    }
    FVT_STAMP(4 * s + 3);
    if (T.x_outer) {
      double v[NF];
      T.outer_x(cx, si_x, v);
      FVT_FENCE();
      if (is_delp) {
        double wa[NF];
        const bool last = T.east && T.xg == GXN - 1;
#pragma unroll
        for (int f = 0; f < NF; ++f) wa[f] = LDG(S.mfx + kb, T.xoff + (unsigned)(f * RB));
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          mfx[f] = v[f] * xf[f] + LS.priv[f][tid];
          v[f] = mfx[f];
          if (f < C || last) STG(S.mfx + kb, T.xoff + (unsigned)(f * RB)) = wa[f] + mfx[f];  // flux_capacitor (d_sw.py:33-60)
        }
      } else {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          v[f] = v[f] * mfx[f];
          if (mass_weighted) {
            const double* ms = LS.mass + (T.xr + 1) * MP + C * T.xg + f;  // the cells on either side of the face
            v[f] = v[f] + 0.5 * damp * (ms[0] + ms[1]) * LS.priv[f][tid];
          }
        }
      }
      T.put_ax(v);
    }
    FVT_FENCE();
    if (T.y_outer) {
      double v[NF];
      T.outer_y(cy, si_y, v);
      FVT_FENCE();
      if (is_delp) {
        double wa[NF];
        const bool last = T.north && T.yg == GYN - 1;
#pragma unroll
        for (int f = 0; f < NF; ++f) wa[f] = LDG(S.mfy + kb, T.yoff + (unsigned)(f * sj8));
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          mfy[f] = v[f] * yf[f] + LS.priv[NF + f][tid];
          v[f] = mfy[f];
          if (f < C || last) STG(S.mfy + kb, T.yoff + (unsigned)(f * sj8)) = wa[f] + mfy[f];
        }
      } else {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          v[f] = v[f] * mfy[f];
          if (mass_weighted) {
            const double* ms = LS.mass + (C * T.yg + f) * MP + T.ycol - 2;
            v[f] = v[f] + 0.5 * damp * (ms[0] + ms[MP]) * LS.priv[NF + f][tid];
          }
        }
      }
      T.put_ay(v);
    }
    __syncthreads();
    {
      int jj[NEC], ii[NEC];
      unsigned c2[NEC];
      double ra[NEC], dwv[NEC];
      T.cell_places(jj, ii, c2);
#pragma unroll
      for (int t = 0; t < NEC; ++t) ra[t] = LDG(m.rarea, c2[t]);
      if (is_w && w_on) {
#pragma unroll
        for (int t = 0; t < NEC; ++t) dwv[t] = LDG(S.dw + kb, c2[t]);
      }
#pragma unroll
      for (int t = 0; t < NEC; ++t) {
        const double am = LS.mass[(jj[t] + 1) * MP + ii[t] + 1];
        double val;
        if (is_delp) {
          // the new delp (apply_pt_delp_fluxes, d_sw.py:148-201)
          dn[t] = am + T.flux_increment(jj[t], ii[t], ra[t]);
          val = dn[t];
        } else {
          val = T.flux_form(jj[t], ii[t], am, ra[t]) / dn[t];
          if (is_w && w_on) val = val + dwv[t];  // adjust_w_and_qcon (d_sw.py:331-350)
        }
        STG(qout, c2[t]) = val;
      }
    }
    if (s < 3) __syncthreads();  // (the cell update read sq / ax / ay)
  }
  FVT_STAMP(16);
}


// ---- the same with 512 threads: x-runs and y-runs owned by different waves ----------------------------------------------------
// Round 5.  The 256-thread form holds the Courant numbers, area fluxes and mass fluxes of ten faces per thread (its y-run and
// its x-run): ~250 VGPRs = two waves per SIMD, and the counters showed the issue slots two thirds empty (one wave issues at most
// every ~8 cycles; 54 % of wave time in memory / LDS / barrier waits with nothing else to issue).  Here waves 0-3 own the x-runs
// (inner sweep in x on q, outer sweep in x on q_i) and waves 4-7 the y-runs, so a thread carries the operands of FIVE faces:
// <= 128 VGPRs = four waves per SIMD at the same LDS per workgroup, the same tile and the same arithmetic (same bits).  The
// stages every thread shares (footprint, damping passes, cell update) are spread over twice the threads.
template <int MORD, bool EX, bool EY>
__device__ __forceinline__ void fvt_scalars_tile_split(FvtLdsScalars& LS, const Geo& g, const FvMet& m, const FvtScalars& S, int bx, int by, int k) {
  constexpr int NT = 512;
  using Tile = FvtTile<MORD, EX, EY, NT>;
  constexpr int NCU = Tile::NCU;
  FvtLds& L = LS.t;
  const int tid = (int)threadIdx.x;
  const bool xrole = tid < 256;  // wave-uniform
  const long kb = (long)k * g.sk;
  const int sj8 = g.sj * RB;
  const bool w_on = S.damp_w[k] > 1e-5;
  double* const priv = &LS.priv[0][0];  // [NF][NT]: the damping fluxes of the thread's faces while the sweeps run
  double c[NF], af[NF], mf[NF];    // Courant numbers, area fluxes, mass fluxes (unit fluxes of w, q_con, pt) of the run's faces
  D2 fp[5][Tile::Pieces::NP];      // the thread's pieces of the footprints: every load of the tile's inputs is in flight at once
  {
    Tile T(L, g, m, bx, by, k, tid);
#pragma unroll
    for (int s = 0; s < 4; ++s) T.fetch_footprint(S.q[s] + kb, fp[s]);
    if (S.winds) T.fetch_footprint(S.q[4] + kb, fp[4]);
    const unsigned roff = xrole ? T.xoff : T.yoff;
  if (xrole) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      c[f] = LDG(S.crx + kb, roff + (unsigned)(f * RB));
      af[f] = LDG(S.xfx + kb, roff + (unsigned)(f * RB));
    }
  } else {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      c[f] = LDG(S.cry + kb, roff + (unsigned)(f * sj8));
      af[f] = LDG(S.yfx + kb, roff + (unsigned)(f * sj8));
    }
  }
  }

  // one pass of the tile: s = 0 .. 4 = delp, w, q_con, pt, (the winds:) relative vorticity -- a compile-time constant
  auto pass = [&](auto s_) {
    constexpr int s = decltype(s_)::value;
    constexpr bool is_delp = s == 0, is_w = s == 1, is_vort = s == 4;
    FVT_STAMP(4 * s);
    const real* const q = S.q[s] + kb;
    real* const qout = is_vort ? nullptr : S.qout[is_vort ? 0 : s] + kb;
    // DelnFlux with mass (q_con, pt); delp: plain DelnFlux; w: DelnFluxNoSG -> heat_diss; vorticity: DelnFluxNoSG -> the winds
    const bool mass_weighted = s == 2 || s == 3;
    const double damp = S.fac[s][k];
    // the thread's places, derived again for every scalar: only the operands above live through the whole kernel
    Tile T(L, g, m, bx, by, k, FVT_LAUNDER(tid));
    // the thread's run: its five faces in memory (first face, step), whether it takes part in the inner / the outer sweep
    const unsigned roff = xrole ? T.xoff : T.yoff;
    const bool run_outer = xrole ? T.x_outer : T.y_outer;
    const bool lane_lo = xrole ? (T.west && T.xg == 0) : (T.south && T.yg == 0);
    const bool lane_hi = xrole ? (T.east && T.xg == GXN - 1) : (T.north && T.yg == GYN - 1);
    const bool last_face = lane_hi;  // the run that stores the face past the end of the tile (ie + 1 / je + 1)
    // the one-sided forms' spacings (edge tiles), loaded by the runs that hold the edge where a sweep needs them
    auto spacing = [&]() {
      FvtSpacing sp;
      if (EX && xrole) sp = fvt_spacing(m.dxa, (unsigned)((T.jlo + T.xrow) * sj8), RB, g.is, g.ie, lane_lo, lane_hi);
      if (EY && !xrole) sp = fvt_spacing(m.dya, (unsigned)((T.ilo + T.ycol) * RB), sj8, g.js, g.je, lane_lo, lane_hi);
      return sp;
    };
    T.stage_damping_planes();  // (its loads first: the metric planes are cache hits, the footprint may still be on its way)
    T.load_damping_rarea();
    T.place_footprint(q, fp[s], qout);
    // the winds: the kinetic energy and the damped vorticity at the tile's (TI + 1) x (TJ + 1) B-grid points, on their way now,
    // into the LDS after the inner sweeps (where q's footprint and the mass tile are dead)
    constexpr int BW = TI + 1, NBP = (BW * (TJ + 1) + NT - 1) / NT;
    double bke[NBP], bvb[NBP];
    if (is_vort) {
#pragma unroll
      for (int t = 0; t < NBP; ++t) {
        int e = tid + NT * t;
        if (e >= BW * (TJ + 1)) e = BW * (TJ + 1) - 1;
        const int r = e / BW, cc = e - r * BW;
        const unsigned o = (unsigned)((T.j0 + r) * sj8 + (T.i0 + cc) * RB);
        bke[t] = LDG(S.ke + kb, o);
        bvb[t] = LDG(S.vort_b + kb, o);
      }
    }
    FVT_ARRIVE(4 * s);
    __syncthreads();
    FVT_STAMP(4 * s + 1);
    if (is_delp) {  // the mass on the tile and one cell around it, from the footprint while it is there
      for (int e = tid; e < (TJ + 2) * (TI + 2); e += NT) {
        const int r = e / (TI + 2), cc = e - r * (TI + 2);
        LS.mass[r * MP + cc] = L.sq[(r + 2) * P + cc + 2];
      }
    }
    {
      const auto D = T.damp(mass_weighted ? 1.0 : damp, S.nord[s][k] > 0.0, S.nmax[s]);
      if (is_w) {
        // (w has no face values: priv is free for dw; with the winds its heating term stays in the LDS as well)
        T.heat_diss(D, S.dw + kb, S.heat_s + kb, S.diss_est + kb, w_on, S.ke_bg[k] * fabs(S.dt), priv, S.winds ? LS.heat : nullptr);
      } else if (run_outer) {
        if (xrole) {
#pragma unroll
          for (int f = 0; f < NF; ++f)
            priv[f * NT + tid] = Tile::face(D, T.sdv[T.xbase + f + 3], D.last[T.xbase + f + 2], D.last[T.xbase + f + 3]);
        } else {
#pragma unroll
          for (int f = 0; f < NF; ++f)
            priv[f * NT + tid] = Tile::face(D, T.sdu[T.ybase + (f + 3) * P], D.last[T.ybase + (f + 2) * P], D.last[T.ybase + (f + 3) * P]);
        }
      }
      FVT_ARRIVE(4 * s + 1);
      __syncthreads();  // the sweeps overwrite the damping planes
    }
    if (is_vort) T.add_2d(S.fC);  // the damped scalar was the relative vorticity, the transported one is the absolute (d_sw.py:389-402)
    FVT_STAMP(4 * s + 2);
    // stage I: the inner sweep of the thread's run on q, and the field advected along it (fvtp2d.py:34-77).  The inner fluxes of
    // the runs that take part in the outer sweep wait in ax / ay, each run using the places of the C faces it opens (the places
    // its final fluxes go to; the face it shares with the next run is that run's place) and a register for the last one,
    // instead of ten registers across the barrier.
    double si_last = 0.0;
    double* const slot = xrole ? L.u.s.ax + T.xr * PJ + C * T.xg : L.u.s.ay + (C * T.yg) * TI + T.ycol - 3;  // of the run's first face
    constexpr int XS = 1, YS = TI;  // from face to face in ax / ay
    if (xrole) {
      if (T.x_on) {
        double Q[NF + 5], ar[C];
#pragma unroll
        for (int u = 0; u < NF + 5; ++u) Q[u] = L.sq[T.xbase + u];
        if (Tile::RC) {  // halo rows of a corner tile: the three corner columns hold the x-direction copies
          const bool halo_row = T.south ? T.xrow < 3 : T.xrow >= TJ + 3;
          const int b = T.south ? T.xrow : T.xrow - (TJ + 3);
          if (halo_row && T.west && T.xg == 0) {
#pragma unroll
            for (int a = 0; a < 3; ++a) Q[a] = L.sqc[b * 3 + a];
          }
          if (halo_row && T.east && T.xg == GXN - 1) {
#pragma unroll
            for (int a = 0; a < 3; ++a) Q[NF + 2 + a] = L.sqc[b * 3 + a];
          }
        }
        double si[NF];
        fvt_run<MORD, EX>(Q, c, lane_lo, lane_hi, spacing(), si);
        FVT_FENCE();
#pragma unroll
        for (int t = 0; t < C; ++t) ar[t] = LDG(m.area, roff + (unsigned)(t * RB));
        if (T.x_outer) {
#pragma unroll
          for (int f = 0; f < C; ++f) slot[f * XS] = si[f];
          si_last = si[C];
        }
#pragma unroll
        for (int t = 0; t < C; ++t)
          L.u.s.sqj[T.xrow * PJ + C * T.xg + t] = (Q[t + 3] * ar[t] + af[t] * si[t] - af[t + 1] * si[t + 1]) / (ar[t] + af[t] - af[t + 1]);
      }
    } else {
      if (T.y_on) {
        double Q[NF + 5], ar[C];
#pragma unroll
        for (int u = 0; u < NF + 5; ++u) Q[u] = L.sq[T.ybase + u * P];
        double si[NF];
        fvt_run<MORD, EY>(Q, c, lane_lo, lane_hi, spacing(), si);
        FVT_FENCE();
#pragma unroll
        for (int t = 0; t < C; ++t) ar[t] = LDG(m.area, roff + (unsigned)(t * sj8));
        if (T.y_outer) {
#pragma unroll
          for (int f = 0; f < C; ++f) slot[f * YS] = si[f];
          si_last = si[C];
        }
#pragma unroll
        for (int t = 0; t < C; ++t)
          L.u.s.sqi[T.ybase + t * P] = (Q[t + 3] * ar[t] + af[t] * si[t] - af[t + 1] * si[t + 1]) / (ar[t] + af[t] - af[t + 1]);
      }
    }
    FVT_ARRIVE(4 * s + 2);
    __syncthreads();
    FVT_STAMP(4 * s + 3);
    double* const tke = L.sq;     // the winds: kinetic energy / damped vorticity at the tile's B-grid points, pitch BW
    double* const tvb = LS.mass;
    if (is_vort) {
#pragma unroll
      for (int t = 0; t < NBP; ++t) {
        const int e = tid + NT * t;
        if (e < BW * (TJ + 1)) tke[e] = S.ke_plus_vort ? bke[t] + bvb[t] : bke[t], tvb[e] = bvb[t];
      }
    }
    // stage II: the outer sweep on the field advected across the run (fvtp2d.py:80-119), the fluxes through the run's faces
    double v[NF];
    double wind[NF];  // the winds: the old wind on the run's faces, on its way while the sweep runs
    if (is_vort && run_outer) {
      if (xrole) {
#pragma unroll
        for (int f = 0; f < NF; ++f) wind[f] = LDG(S.v + kb, roff + (unsigned)(f * RB));
      } else {
#pragma unroll
        for (int f = 0; f < NF; ++f) wind[f] = LDG(S.u + kb, roff + (unsigned)(f * sj8));
      }
    }
    if (run_outer) {
      {
        double Q[NF + 5], out[NF];
        if (xrole) {
#pragma unroll
          for (int u = 0; u < NF + 5; ++u) Q[u] = L.u.s.sqi[T.xr * P + C * T.xg + u];
          fvt_run<MORD, EX>(Q, c, lane_lo, lane_hi, spacing(), out);
        } else {
#pragma unroll
          for (int u = 0; u < NF + 5; ++u) Q[u] = L.u.s.sqj[(C * T.yg + u) * PJ + T.ycol - 3];
          fvt_run<MORD, EY>(Q, c, lane_lo, lane_hi, spacing(), out);
        }
        if (xrole) {
#pragma unroll
          for (int f = 0; f < C; ++f) v[f] = 0.5 * (out[f] + slot[f * XS]);
        } else {
#pragma unroll
          for (int f = 0; f < C; ++f) v[f] = 0.5 * (out[f] + slot[f * YS]);
        }
        v[C] = 0.5 * (out[C] + si_last);
      }
      FVT_FENCE();
      if (is_vort) {
#pragma unroll
        for (int f = 0; f < NF; ++f) v[f] = v[f] * af[f];  // (the unit fluxes of the vorticity are the area fluxes)
      } else if (is_delp) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          mf[f] = v[f] * af[f] + priv[f * NT + tid];
          v[f] = mf[f];
        }
        // flux_capacitor (d_sw.py:33-60): mfx += fx, mfy += fy, each face by the run that opens it.  (As load - add - store the
        // accumulators' loads sat between the sweep and the stores with a memory latency to wait for: 7.6 k cycles for delp's
        // last stage against 4.5 k for pt's.)
        if (xrole) {
#pragma unroll
          for (int f = 0; f < NF; ++f)
            if (f < C || last_face) fvt_accumulate((real*)((char*)(S.mfx + kb) + roff + (unsigned)(f * RB)), mf[f]);
        } else {
#pragma unroll
          for (int f = 0; f < NF; ++f)
            if (f < C || last_face) fvt_accumulate((real*)((char*)(S.mfy + kb) + roff + (unsigned)(f * sj8)), mf[f]);
        }
      } else {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          v[f] = v[f] * mf[f];
          if (mass_weighted) {  // the cells on either side of the face
            const double* ms = xrole ? LS.mass + (T.xr + 1) * MP + C * T.xg + f : LS.mass + (C * T.yg + f) * MP + T.ycol - 2;
            const double m1 = xrole ? ms[1] : ms[MP];
            v[f] = v[f] + 0.5 * damp * (ms[0] + m1) * priv[f * NT + tid];
          }
        }
      }
      // a face's flux is put down by the run that opens it; the last face of the row / column by its last run
      if (is_vort) {
        // (the winds take the fluxes from the registers below)
      } else if (xrole) {
#pragma unroll
        for (int f = 0; f < C; ++f) slot[f * XS] = v[f];
        if (T.xg == GXN - 1) slot[C * XS] = v[C];
      } else {
#pragma unroll
        for (int f = 0; f < C; ++f) slot[f * YS] = v[f];
        if (T.yg == GYN - 1) slot[C * YS] = v[C];
      }
    }
    __syncthreads();
    if (is_vort) {
      // ---- the winds.  On a face of the tile: u_and_v_from_ke (d_sw.py:406-477) with the vorticity flux through it, the
      // vorticity-damping increment (vort_differencing :353-380, the damping flux from priv), the final wind (update_u_and_v
      // :582-608) and the face's terms of heat_source_from_vorticity_damping (:493-577: ubt, fy / vbt, fx); then, per cell, the
      // damping term from its four faces and the heating.  The face terms travel through the LDS arrays the sweeps are done
      // with: vbt -> ax, fx -> sqj (rows x TI + 1 faces), ubt -> ay, fy -> sqi (TJ + 1 faces x columns).
      const bool upd = S.damp_vt[k] > 1e-5;
      const double dck = S.d_con_k[k];
      const bool don = dck > 1e-5;
      double* const avbt = L.u.s.ax;
      double* const afx = L.u.s.sqj;
      double* const aubt = L.u.s.ay;
      double* const afy = L.u.s.sqi;
      if (run_outer) {
        if (xrole) {  // x-faces: v-points (i0 + C * xg + f, j0 + xr)
          double dyv[NF], rdyv[NF];
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            dyv[f] = LDG(m.dy, roff + (unsigned)(f * RB));
            rdyv[f] = LDG(S.rdy, roff + (unsigned)(f * RB));
          }
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            const int b = T.xr * BW + C * T.xg + f;
            const double vmid = wind[f] * dyv[f] + tke[b] - tke[b + BW] - v[f];  // v_from_ke (d_sw.py:423-436)
            const double ut2 = priv[f * NT + tid];
            const double vyd = don ? tvb[b] - tvb[b + BW] : 0.0;
            const double vbt = (vyd - ut2) * rdyv[f];
            const double fxh = vmid * rdyv[f];
            if (f < C || last_face) STG(S.v_out + kb, roff + (unsigned)(f * RB)) = upd ? vmid - ut2 : vmid;
            if (f < C || T.xg == GXN - 1) avbt[T.xr * PJ + C * T.xg + f] = vbt, afx[T.xr * PJ + C * T.xg + f] = fxh;
          }
        } else {  // y-faces: u-points (i0 + ycol - 3, j0 + C * yg + f)
          double dxv[NF], rdxv[NF];
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            dxv[f] = LDG(m.dx, roff + (unsigned)(f * sj8));
            rdxv[f] = LDG(S.rdx, roff + (unsigned)(f * sj8));
          }
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            const int b = (C * T.yg + f) * BW + T.ycol - 3;
            const double umid = wind[f] * dxv[f] + tke[b] - tke[b + 1] + v[f];  // u_from_ke (d_sw.py:406-420)
            const double vt2 = priv[f * NT + tid];
            const double vxd = don ? tvb[b] - tvb[b + 1] : 0.0;
            const double ubt = (vxd + vt2) * rdxv[f];
            const double fyh = umid * rdxv[f];
            if (f < C || last_face) STG(S.u_out + kb, roff + (unsigned)(f * sj8)) = upd ? umid + vt2 : umid;
            if (f < C || T.yg == GYN - 1) aubt[(C * T.yg + f) * TI + T.ycol - 3] = ubt, afy[(C * T.yg + f) * TI + T.ycol - 3] = fyh;
          }
        }
      }
      __syncthreads();
      {
        int jj[NCU], ii[NCU];
        unsigned c2[NCU];
        T.cell_places(jj, ii, c2);
        const bool any = S.d_con > 1e-5 || S.do_skeb;
#pragma unroll
        for (int t = 0; t < NCU; ++t) {
          if (NCU * NT > TI * TJ && tid + NT * t >= TI * TJ) continue;  // (no cell of its own: the spare lanes would add twice)
          const double heat_s = LS.heat[jj[t] * TI + ii[t]];
          if (don || S.do_skeb) {
            const int ey = jj[t] * TI + ii[t], ex = jj[t] * PJ + ii[t];
            const double ubt0 = aubt[ey], ubtj = aubt[ey + TI], fy0 = afy[ey], fyj = afy[ey + TI];
            const double vbt0 = avbt[ex], vbti = avbt[ex + 1], fx0 = afx[ex], fxi = afx[ex + 1];
            const double gy0 = fy0 * ubt0, gyj = fyj * ubtj, gx0 = fx0 * vbt0, gxi = fxi * vbti;
            const double u2 = fy0 + fyj, du2 = ubt0 + ubtj, v2 = fx0 + fxi, dv2 = vbt0 + vbti;
            const double dampterm = LDG(S.rsin2, c2[t]) * 0.25 *
                                    ((ubt0 * ubt0 + ubtj * ubtj + vbt0 * vbt0 + vbti * vbti) + 2.0 * (gy0 + gyj + gx0 + gxi) -
                                     LDG(S.cosa_s, c2[t]) * (u2 * dv2 + v2 * du2 + du2 * dv2));
            const double hs = LS.newmass[jj[t] * TI + ii[t]] * (heat_s - dck * dampterm);
            if (any) {
              fvt_accumulate((real*)((char*)(S.heat_source + kb) + c2[t]), hs);
              if (S.do_skeb) STG(S.diss_est + kb, c2[t]) = LDG(S.diss_est + kb, c2[t]) - dampterm;
            }
          } else if (any) {
            fvt_accumulate((real*)((char*)(S.heat_source + kb) + c2[t]), heat_s);
          }
        }
      }
      if ((EX || EY) && S.copy_wind_halo) {
        // the output buffers of the winds get the halo the inputs have (the caller swaps the buffers): this tile's share of the
        // storage outside the faces the kernel writes -- u: [is, ie] x [js, je + 1], v: [is, ie + 1] x [js, je]
        const int xa = T.west ? 0 : T.i0, xb = T.east ? g.ni : T.i0 + TI, ya = T.south ? 0 : T.j0, yb = T.north ? g.nj : T.j0 + TJ;
        const int bw = xb - xa, nbox = bw * (yb - ya);
        for (int e = tid; e < nbox; e += NT) {
          const int r = e / bw, i = xa + (e - r * bw), j = ya + r;
          const unsigned o = (unsigned)(j * sj8 + i * RB);
          const bool in_i = i >= g.is && i <= g.ie, in_j = j >= g.js && j <= g.je;
          if (!(in_i && (in_j || j == g.je + 1))) STG(S.u_out + kb, o) = LDG(S.u + kb, o);
          if (!((in_i || i == g.ie + 1) && in_j)) STG(S.v_out + kb, o) = LDG(S.v + kb, o);
        }
      }
      return;
    }
    {
      int jj[NCU], ii[NCU];
      unsigned c2[NCU];
      double ra[NCU], dwv[NCU];
      T.cell_places(jj, ii, c2);
#pragma unroll
      for (int t = 0; t < NCU; ++t) ra[t] = LDG(m.rarea, c2[t]);
      if (is_w && w_on) {
#pragma unroll
        for (int t = 0; t < NCU; ++t) dwv[t] = priv[jj[t] * TI + ii[t]];
      }
#pragma unroll
      for (int t = 0; t < NCU; ++t) {
        const double am = LS.mass[(jj[t] + 1) * MP + ii[t] + 1];
        double val;
        if (is_delp) {
          // the new delp (apply_pt_delp_fluxes, d_sw.py:148-201)
          val = am + T.flux_increment(jj[t], ii[t], ra[t]);
          LS.newmass[jj[t] * TI + ii[t]] = val;  // (read back by this same thread)
        } else {
          val = T.flux_form(jj[t], ii[t], am, ra[t]) / LS.newmass[jj[t] * TI + ii[t]];
          if (is_w && w_on) val = val + dwv[t];  // adjust_w_and_qcon (d_sw.py:331-350)
        }
        STG(qout, c2[t]) = val;
      }
    }
    FVT_ARRIVE(4 * s + 3);
    if (s < 3 || S.winds) __syncthreads();  // (the cell update read sq / ax / ay)
  };
  pass(std::integral_constant<int, 0>{});
  pass(std::integral_constant<int, 1>{});
  pass(std::integral_constant<int, 2>{});
  pass(std::integral_constant<int, 3>{});
  if (S.winds) pass(std::integral_constant<int, 4>{});
  FVT_STAMP(S.winds ? 20 : 16);
}


// ---- round 6: the 512-thread form on the resident layout (FvtLdsRes) --------------------------------------------------------------
// What changes against fvt_scalars_tile_split (same stages, same expressions, same bits):
//  * del6_v / del6_u are staged once per tile; the iterate has a plane of its own; no store of face values (see FvtLdsRes).
//  * The barrier intervals of a pass -- each now holds a piece of the damping AND a piece of the transport, which read different
//    arrays:   [footprint] | [damping pass 1: q -> iterate; inner sweep: q -> q_i / q_j] | [outer sweep -> registers; damping pass 2
//    read] | [pass 2 written] | [face values from the iterate -> fluxes] | [cell update]      -- six barriers where there were eight,
//    and none of them closes an interval of a dozen instructions.
//  * The new mass and w's heating term of a thread's cells stay in its registers (the thread that forms them is the thread that
//    uses them; the footprint pieces prefetched at kernel start have been released by then): 12 KB of LDS less.  73.6 KB per workgroup.
struct FvtLdsScalarsRes {
  FvtLdsRes t;
  double mass[(TJ + 2) * MP];  // the mass on the tile and one cell around it (the winds: the damped vorticity at the B-grid points)
};
static_assert(2 * sizeof(FvtLdsScalarsRes) <= 160 * 1024, "two workgroups per CU");

template <int MORD, bool EX, bool EY>
__device__ __forceinline__ void fvt_scalars_tile_res(FvtLdsScalarsRes& LS, const Geo& g, const FvMet& m, const FvtScalars& S, int bx, int by, int k) {
  constexpr int NT = 512;
  using Tile = FvtTile<MORD, EX, EY, NT, FvtLdsRes>;
  constexpr int NCU = Tile::NCU, DRC = Tile::DRC;
  FvtLdsRes& L = LS.t;
  const int tid = (int)threadIdx.x;
  const bool xrole = tid < 256;  // wave-uniform
  const long kb = (long)k * g.sk;
  const int sj8 = g.sj * RB;
  const bool w_on = S.damp_w[k] > 1e-5;
  double c[NF], af[NF], mf[NF];    // Courant numbers, area fluxes, mass fluxes (unit fluxes of w, q_con, pt) of the run's faces
  double nm[NCU], heat_r[NCU];     // the new mass and w's heating term of the thread's cells
  D2 fp[5][Tile::Pieces::NP];      // the thread's pieces of the footprints: every load of the tile's inputs is in flight at once
#ifndef FVT_RA_REGS
#define FVT_RA_REGS 0  // 1: rarea of the thread's damping run loaded once per tile and kept in registers (0: once per pass)
#endif
  double ra_keep[DRC];
#pragma unroll
  for (int t = 0; t < NCU; ++t) nm[t] = 1.0, heat_r[t] = 0.0;
  {
    Tile T(L, g, m, bx, by, k, tid);
    if (FVT_RA_REGS) {
      T.load_damping_rarea();
#pragma unroll
      for (int t = 0; t < DRC; ++t) ra_keep[t] = T.dra[t];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) T.fetch_footprint(S.q[s] + kb, fp[s]);
    if (S.winds) T.fetch_footprint(S.q[4] + kb, fp[4]);
    T.stage_damping_planes();  // once per tile
    T.stage_spacings();
    T.publish_corner_places();
    const unsigned roff = xrole ? T.xoff : T.yoff;
    if (xrole) {
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        c[f] = LDG(S.crx + kb, roff + (unsigned)(f * RB));
        af[f] = LDG(S.xfx + kb, roff + (unsigned)(f * RB));
      }
    } else {
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        c[f] = LDG(S.cry + kb, roff + (unsigned)(f * sj8));
        af[f] = LDG(S.yfx + kb, roff + (unsigned)(f * sj8));
      }
    }
  }

  // one pass of the tile: s = 0 .. 4 = delp, w, q_con, pt, (the winds:) relative vorticity -- a compile-time constant
  auto pass = [&](auto s_) {
    constexpr int s = decltype(s_)::value;
    constexpr bool is_delp = s == 0, is_w = s == 1, is_vort = s == 4;
    FVT_STAMP(4 * s);
    const real* const q = S.q[s] + kb;
    real* const qout = is_vort ? nullptr : S.qout[is_vort ? 0 : s] + kb;
    // DelnFlux with mass (q_con, pt); delp: plain DelnFlux; w: DelnFluxNoSG -> heat_diss; vorticity: DelnFluxNoSG -> the winds
    const bool mass_weighted = s == 2 || s == 3;
    const double damp = S.fac[s][k];
    const double d0 = mass_weighted ? 1.0 : damp;
    const int iters = (S.nord[s][k] > 0.0) ? S.nmax[s] : 0;  // passes of the damping before its fluxes: 0, 1 or 2 (block-uniform)
    // the thread's places, derived again for every scalar: only the operands above live through the whole kernel
    Tile T(L, g, m, bx, by, k, FVT_LAUNDER(tid), s > 0);  // (pass 0's barrier has published the corner cells' places)
    const unsigned roff = xrole ? T.xoff : T.yoff;
    const bool run_outer = xrole ? T.x_outer : T.y_outer;
    const bool lane_lo = xrole ? (T.west && T.xg == 0) : (T.south && T.yg == 0);
    const bool lane_hi = xrole ? (T.east && T.xg == GXN - 1) : (T.north && T.yg == GYN - 1);
    const bool last_face = lane_hi;  // the run that stores the face past the end of the tile (ie + 1 / je + 1)
    // the one-sided forms' spacings, from the tile's LDS table (the same values fvt_spacing loads)
    auto spacing = [&]() {
      FvtSpacing sp;
#pragma unroll
      for (int t = 0; t < 4; ++t) sp.d[t] = 0.0;
      if (EX && xrole && (lane_lo || lane_hi)) {
#pragma unroll
        for (int t = 0; t < 4; ++t) sp.d[t] = L.spx[T.xrow * 4 + t];
      }
      if (EY && !xrole && (lane_lo || lane_hi)) {
#pragma unroll
        for (int t = 0; t < 4; ++t) sp.d[t] = L.spy[T.ycol * 4 + t];
      }
      return sp;
    };
    // the damping flux through face f of the thread's run, from the last iterate (q itself if no pass ran)
    const typename Tile::Damped D{iters == 0 ? L.sq : T.plane, d0, iters == 0};
    auto dface = [&](int f) -> double {
      return xrole ? Tile::face(D, T.sdv[T.xbase + f + 3], D.last[T.xbase + f + 2], D.last[T.xbase + f + 3])
                   : Tile::face(D, T.sdu[T.ybase + (f + 3) * P], D.last[T.ybase + (f + 2) * P], D.last[T.ybase + (f + 3) * P]);
    };

    // ---- interval 0: the footprint
    if (FVT_RA_REGS) {
#pragma unroll
      for (int t = 0; t < DRC; ++t) T.dra[t] = ra_keep[t];
    } else {
      T.load_damping_rarea();
    }
    T.template place_footprint<true>(q, fp[s], qout);
    constexpr int BW = TI + 1, NBP = (BW * (TJ + 1) + NT - 1) / NT;
    double bke[NBP], bvb[NBP];  // the winds: kinetic energy and damped vorticity at the tile's B-grid points, on their way
    if (is_vort) {
#pragma unroll
      for (int t = 0; t < NBP; ++t) {
        int e = tid + NT * t;
        if (e >= BW * (TJ + 1)) e = BW * (TJ + 1) - 1;
        const int r = e / BW, cc = e - r * BW;
        const unsigned o = (unsigned)((T.j0 + r) * sj8 + (T.i0 + cc) * RB);
        bke[t] = LDG(S.ke + kb, o);
        bvb[t] = LDG(S.vort_b + kb, o);
      }
    }
    if (is_vort && (EX || EY) && S.copy_wind_halo) {
      // the output buffers of the winds get the halo the inputs have (the caller swaps the buffers): this tile's share of the
      // storage outside the faces the kernel writes -- u: [is, ie] x [js, je + 1], v: [is, ie + 1] x [js, je].  Here, at the top of
      // the pass, the copy travels under the pass (at its end it was 4.5 k cycles of an edge tile's 110 k with nothing beside it).
      const int xa = T.west ? 0 : T.i0, xb = T.east ? g.ni : T.i0 + TI, ya = T.south ? 0 : T.j0, yb = T.north ? g.nj : T.j0 + TJ;
      const int bw = xb - xa, nbox = bw * (yb - ya);
      for (int e = tid; e < nbox; e += NT) {
        const int r = e / bw, i = xa + (e - r * bw), j = ya + r;
        const unsigned o = (unsigned)(j * sj8 + i * RB);
        const bool in_i = i >= g.is && i <= g.ie, in_j = j >= g.js && j <= g.je;
        if (!(in_i && (in_j || j == g.je + 1))) STG(S.u_out + kb, o) = LDG(S.u + kb, o);
        if (!((in_i || i == g.ie + 1) && in_j)) STG(S.v_out + kb, o) = LDG(S.v + kb, o);
      }
    }
    FVT_ARRIVE(4 * s);
    __syncthreads();
    if (Tile::RC) {  // a corner tile: its corner block from the footprint (copy_corners_y; copy_corners_x to sqc and to the output)
      T.corners_from_lds(qout);
      __syncthreads();
    }
    FVT_STAMP(4 * s + 1);

    // ---- interval 1: damping pass 1 (q -> iterate) and the inner sweep (q -> q_i / q_j, its fluxes to the faces' places)
    if (is_delp) {  // the mass on the tile and one cell around it, from the footprint while it is there
      for (int e = tid; e < (TJ + 2) * (TI + 2); e += NT) {
        const int r = e / (TI + 2), cc = e - r * (TI + 2);
        LS.mass[r * MP + cc] = L.sq[(r + 2) * P + cc + 2];
      }
    }
    double fv0[NF];  // (the winds without a damping pass: the face values of the RELATIVE vorticity, before f is added to q)
    if (is_vort && iters == 0 && run_outer) {
#pragma unroll
      for (int f = 0; f < NF; ++f) fv0[f] = dface(f);
    }
    if (iters >= 1) {
      double res[DRC], fix;
      T.template deln_compute<true>(d0, res, fix);
      T.deln_store(res, fix);
    }
    if (is_vort) {
      __syncthreads();      // (the pass above has read the relative vorticity)
      T.add_2d_lds(S.fC);   // ... the transported scalar is the absolute one (d_sw.py:389-402); ends with a barrier
    }
    double si_last = 0.0;
    double* const slot = xrole ? L.u.s.ax + T.xr * PJ + C * T.xg : L.u.s.ay + (C * T.yg) * TI + T.ycol - 3;  // of the run's first face
    constexpr int XS = 1, YS = TI;  // from face to face in ax / ay
    if (xrole) {
      if (T.x_on) {
        double Q[NF + 5], ar[C];
#pragma unroll
        for (int u = 0; u < NF + 5; ++u) Q[u] = L.sq[T.xbase + u];
        if (Tile::RC) {  // halo rows of a corner tile: the three corner columns hold the x-direction copies
          const bool halo_row = T.south ? T.xrow < 3 : T.xrow >= TJ + 3;
          const int b = T.south ? T.xrow : T.xrow - (TJ + 3);
          if (halo_row && T.west && T.xg == 0) {
#pragma unroll
            for (int a = 0; a < 3; ++a) Q[a] = L.sqc[b * 3 + a];
          }
          if (halo_row && T.east && T.xg == GXN - 1) {
#pragma unroll
            for (int a = 0; a < 3; ++a) Q[NF + 2 + a] = L.sqc[b * 3 + a];
          }
        }
        double si[NF];
        fvt_run<MORD, EX>(Q, c, lane_lo, lane_hi, spacing(), si);
        FVT_FENCE();
#pragma unroll
        for (int t = 0; t < C; ++t) ar[t] = LDG(m.area, roff + (unsigned)(t * RB));
        if (T.x_outer) {
#pragma unroll
          for (int f = 0; f < C; ++f) slot[f * XS] = si[f];
          si_last = si[C];
        }
#pragma unroll
        for (int t = 0; t < C; ++t)
          L.u.s.sqj[T.xrow * PJ + C * T.xg + t] = (Q[t + 3] * ar[t] + af[t] * si[t] - af[t + 1] * si[t + 1]) / (ar[t] + af[t] - af[t + 1]);
      }
    } else {
      if (T.y_on) {
        double Q[NF + 5], ar[C];
#pragma unroll
        for (int u = 0; u < NF + 5; ++u) Q[u] = L.sq[T.ybase + u * P];
        double si[NF];
        fvt_run<MORD, EY>(Q, c, lane_lo, lane_hi, spacing(), si);
        FVT_FENCE();
#pragma unroll
        for (int t = 0; t < C; ++t) ar[t] = LDG(m.area, roff + (unsigned)(t * sj8));
        if (T.y_outer) {
#pragma unroll
          for (int f = 0; f < C; ++f) slot[f * YS] = si[f];
          si_last = si[C];
        }
#pragma unroll
        for (int t = 0; t < C; ++t)
          L.u.s.sqi[T.ybase + t * P] = (Q[t + 3] * ar[t] + af[t] * si[t] - af[t + 1] * si[t + 1]) / (ar[t] + af[t] - af[t + 1]);
      }
    }
    FVT_ARRIVE(4 * s + 1);
    __syncthreads();
    FVT_STAMP(4 * s + 2);

    // ---- interval 2: the outer sweep on the field advected across the run (fvtp2d.py:80-119) -> the mean advected values through
    // the run's faces, in registers; then damping pass 2's read of the iterate
    double* const tke = L.sq;     // the winds: kinetic energy / damped vorticity at the tile's B-grid points, pitch BW
    double* const tvb = LS.mass;
    if (is_vort) {  // (q's footprint and the mass tile are dead)
#pragma unroll
      for (int t = 0; t < NBP; ++t) {
        const int e = tid + NT * t;
        if (e < BW * (TJ + 1)) tke[e] = S.ke_plus_vort ? bke[t] + bvb[t] : bke[t], tvb[e] = bvb[t];
      }
    }
    double v[NF];
    // the winds: the old wind on the run's faces and the faces' grid spacings (dy, 1 / dy at the x-faces; dx, 1 / dx at the y-faces),
    // on their way while the sweep runs (at the top of the face stage they were a round trip to memory with nothing beside it)
    double wind[NF], dsp[NF], rdsp[NF];
#ifndef FVT_HOIST_WIND
#define FVT_HOIST_WIND 2  // the faces' spacings of the winds pass: 0 loaded where they are used, 1 before the outer sweep, 2 behind it
#endif
    auto load_spacings = [&]() {
      if (xrole) {
#pragma unroll
        for (int f = 0; f < NF; ++f) dsp[f] = LDG(m.dy, roff + (unsigned)(f * RB)), rdsp[f] = LDG(S.rdy, roff + (unsigned)(f * RB));
      } else {
#pragma unroll
        for (int f = 0; f < NF; ++f) dsp[f] = LDG(m.dx, roff + (unsigned)(f * sj8)), rdsp[f] = LDG(S.rdx, roff + (unsigned)(f * sj8));
      }
    };
    if (is_vort && run_outer) {
      if (xrole) {
#pragma unroll
        for (int f = 0; f < NF; ++f) wind[f] = LDG(S.v + kb, roff + (unsigned)(f * RB));
      } else {
#pragma unroll
        for (int f = 0; f < NF; ++f) wind[f] = LDG(S.u + kb, roff + (unsigned)(f * sj8));
      }
      if (FVT_HOIST_WIND == 1) load_spacings();
    }
    if (run_outer) {
      double Q[NF + 5], out[NF];
      if (xrole) {
#pragma unroll
        for (int u = 0; u < NF + 5; ++u) Q[u] = L.u.s.sqi[T.xr * P + C * T.xg + u];
        fvt_run<MORD, EX>(Q, c, lane_lo, lane_hi, spacing(), out);
#pragma unroll
        for (int f = 0; f < C; ++f) v[f] = 0.5 * (out[f] + slot[f * XS]);
      } else {
#pragma unroll
        for (int u = 0; u < NF + 5; ++u) Q[u] = L.u.s.sqj[(C * T.yg + u) * PJ + T.ycol - 3];
        fvt_run<MORD, EY>(Q, c, lane_lo, lane_hi, spacing(), out);
#pragma unroll
        for (int f = 0; f < C; ++f) v[f] = 0.5 * (out[f] + slot[f * YS]);
      }
      v[C] = 0.5 * (out[C] + si_last);
      FVT_FENCE();
      if (is_w) {  // (no damping fluxes in w's transport: its fluxes are final here)
#pragma unroll
        for (int f = 0; f < NF; ++f) v[f] = v[f] * mf[f];
#pragma unroll
        for (int f = 0; f < C; ++f) slot[f * (xrole ? XS : YS)] = v[f];
        if (xrole ? T.xg == GXN - 1 : T.yg == GYN - 1) slot[C * (xrole ? XS : YS)] = v[C];
      }
    }
    FVT_FENCE();
    if (is_vort && run_outer && FVT_HOIST_WIND == 2) load_spacings();  // (behind the sweep: they travel under the damping's pass 2)
    double res2[DRC], fix2 = 0.0;
    if (iters >= 2) T.template deln_compute<false>(d0, res2, fix2);
    if (iters >= 2) {
      FVT_ARRIVE(4 * s + 2);
      __syncthreads();  // (everyone has read the iterate)
      // ---- interval 3: damping pass 2 put down
      T.deln_store(res2, fix2);
    }
    __syncthreads();  // (the iterate is final; the sweeps' inputs have been read)
    FVT_STAMP(4 * s + 3);

    // ---- interval 4: the damping's face fluxes from the iterate, the fluxes through the run's faces
    if (is_w) {
      // heat_diss: dw and the heating term on the tile, where the sweeps' inputs were (w has no face values)
      T.heat_diss(D, S.dw + kb, S.heat_s + kb, S.diss_est + kb, w_on, S.ke_bg[k] * fabs(S.dt), L.u.s.sqi, S.winds ? L.u.s.sqj : nullptr);
    } else if (run_outer && !is_vort) {
      if (is_delp) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          mf[f] = v[f] * af[f] + dface(f);
          v[f] = mf[f];
        }
        // flux_capacitor (d_sw.py:33-60): mfx += fx, mfy += fy, each face by the run that opens it
        if (xrole) {
#pragma unroll
          for (int f = 0; f < NF; ++f)
            if (f < C || last_face) fvt_accumulate((real*)((char*)(S.mfx + kb) + roff + (unsigned)(f * RB)), mf[f]);
        } else {
#pragma unroll
          for (int f = 0; f < NF; ++f)
            if (f < C || last_face) fvt_accumulate((real*)((char*)(S.mfy + kb) + roff + (unsigned)(f * sj8)), mf[f]);
        }
      } else {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          v[f] = v[f] * mf[f];
          // the cells on either side of the face
          const double* ms = xrole ? LS.mass + (T.xr + 1) * MP + C * T.xg + f : LS.mass + (C * T.yg + f) * MP + T.ycol - 2;
          const double m1 = xrole ? ms[1] : ms[MP];
          v[f] = v[f] + 0.5 * damp * (ms[0] + m1) * dface(f);
        }
      }
      // a face's flux is put down by the run that opens it; the last face of the row / column by its last run
#pragma unroll
      for (int f = 0; f < C; ++f) slot[f * (xrole ? XS : YS)] = v[f];
      if (xrole ? T.xg == GXN - 1 : T.yg == GYN - 1) slot[C * (xrole ? XS : YS)] = v[C];
    }
    if (is_vort) {
      // ---- the winds.  On a face of the tile: u_and_v_from_ke (d_sw.py:406-477) with the vorticity flux through it, the
      // vorticity-damping increment (vort_differencing :353-380: the damping flux from the iterate), the final wind (update_u_and_v
      // :582-608) and the face's terms of heat_source_from_vorticity_damping (:493-577: ubt, fy / vbt, fx); then, per cell, the
      // damping term from its four faces and the heating.  The face terms travel through the LDS arrays the sweeps are done
      // with: vbt -> ax, fx -> sqj (rows x TI + 1 faces), ubt -> ay, fy -> sqi (TJ + 1 faces x columns).
      const bool upd = S.damp_vt[k] > 1e-5;
      const double dck = S.d_con_k[k];
      const bool don = dck > 1e-5;
      double* const avbt = L.u.s.ax;
      double* const afx = L.u.s.sqj;
      double* const aubt = L.u.s.ay;
      double* const afy = L.u.s.sqi;
      if (run_outer) {
        if (FVT_HOIST_WIND == 0) load_spacings();
        if (xrole) {  // x-faces: v-points (i0 + C * xg + f, j0 + xr)
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            const int b = T.xr * BW + C * T.xg + f;
            const double vf = v[f] * af[f];  // (the unit fluxes of the vorticity are the area fluxes)
            const double vmid = wind[f] * dsp[f] + tke[b] - tke[b + BW] - vf;  // v_from_ke (d_sw.py:423-436)
            const double ut2 = iters == 0 ? fv0[f] : dface(f);
            const double vyd = don ? tvb[b] - tvb[b + BW] : 0.0;
            const double vbt = (vyd - ut2) * rdsp[f];
            const double fxh = vmid * rdsp[f];
            if (f < C || last_face) STG(S.v_out + kb, roff + (unsigned)(f * RB)) = upd ? vmid - ut2 : vmid;
            if (f < C || T.xg == GXN - 1) avbt[T.xr * PJ + C * T.xg + f] = vbt, afx[T.xr * PJ + C * T.xg + f] = fxh;
          }
        } else {  // y-faces: u-points (i0 + ycol - 3, j0 + C * yg + f)
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            const int b = (C * T.yg + f) * BW + T.ycol - 3;
            const double vf = v[f] * af[f];
            const double umid = wind[f] * dsp[f] + tke[b] - tke[b + 1] + vf;  // u_from_ke (d_sw.py:406-420)
            const double vt2 = iters == 0 ? fv0[f] : dface(f);
            const double vxd = don ? tvb[b] - tvb[b + 1] : 0.0;
            const double ubt = (vxd + vt2) * rdsp[f];
            const double fyh = umid * rdsp[f];
            if (f < C || last_face) STG(S.u_out + kb, roff + (unsigned)(f * sj8)) = upd ? umid + vt2 : umid;
            if (f < C || T.yg == GYN - 1) aubt[(C * T.yg + f) * TI + T.ycol - 3] = ubt, afy[(C * T.yg + f) * TI + T.ycol - 3] = fyh;
          }
        }
      }
      int jj[NCU], ii[NCU];
      unsigned c2[NCU];
      double rs2[NCU], csa[NCU];  // (loaded before the barrier: they travel while the last waves finish their faces)
      T.cell_places(jj, ii, c2);
#pragma unroll
      for (int t = 0; t < NCU; ++t) rs2[t] = LDG(S.rsin2, c2[t]), csa[t] = LDG(S.cosa_s, c2[t]);
      __syncthreads();
      {
        const bool any = S.d_con > 1e-5 || S.do_skeb;
#pragma unroll
        for (int t = 0; t < NCU; ++t) {
          if (NCU * NT > TI * TJ && tid + NT * t >= TI * TJ) continue;  // (no cell of its own: the spare lanes would add twice)
          const double heat_s = heat_r[t];
          if (don || S.do_skeb) {
            const int ey = jj[t] * TI + ii[t], ex = jj[t] * PJ + ii[t];
            const double ubt0 = aubt[ey], ubtj = aubt[ey + TI], fy0 = afy[ey], fyj = afy[ey + TI];
            const double vbt0 = avbt[ex], vbti = avbt[ex + 1], fx0 = afx[ex], fxi = afx[ex + 1];
            const double gy0 = fy0 * ubt0, gyj = fyj * ubtj, gx0 = fx0 * vbt0, gxi = fxi * vbti;
            const double u2 = fy0 + fyj, du2 = ubt0 + ubtj, v2 = fx0 + fxi, dv2 = vbt0 + vbti;
            const double dampterm = rs2[t] * 0.25 *
                                    ((ubt0 * ubt0 + ubtj * ubtj + vbt0 * vbt0 + vbti * vbti) + 2.0 * (gy0 + gyj + gx0 + gxi) -
                                     csa[t] * (u2 * dv2 + v2 * du2 + du2 * dv2));
            const double hs = nm[t] * (heat_s - dck * dampterm);
            if (any) {
              fvt_accumulate((real*)((char*)(S.heat_source + kb) + c2[t]), hs);
              if (S.do_skeb) STG(S.diss_est + kb, c2[t]) = LDG(S.diss_est + kb, c2[t]) - dampterm;
            }
          } else if (any) {
            fvt_accumulate((real*)((char*)(S.heat_source + kb) + c2[t]), heat_s);
          }
        }
      }
      return;
    }
    int jj[NCU], ii[NCU];
    unsigned c2[NCU];
    double ra[NCU];  // (loaded before the barrier, like the winds' cell stage)
    T.cell_places(jj, ii, c2);
#pragma unroll
    for (int t = 0; t < NCU; ++t) ra[t] = LDG(m.rarea, c2[t]);
    __syncthreads();

    // ---- interval 5: the cell update (apply_fluxes / apply_pt_delp_fluxes / adjust_w_and_qcon, d_sw.py:122-201,331-350)
    {
#pragma unroll
      for (int t = 0; t < NCU; ++t) {
        const double am = LS.mass[(jj[t] + 1) * MP + ii[t] + 1];
        double val;
        if (is_delp) {
          val = am + T.flux_increment(jj[t], ii[t], ra[t]);  // the new delp (apply_pt_delp_fluxes, d_sw.py:148-201)
          nm[t] = val;
        } else {
          val = T.flux_form(jj[t], ii[t], am, ra[t]) / nm[t];
          if (is_w) {
            if (w_on) val = val + L.u.s.sqi[jj[t] * TI + ii[t]];  // adjust_w_and_qcon (d_sw.py:331-350): + dw
            if (S.winds) heat_r[t] = L.u.s.sqj[jj[t] * TI + ii[t]];
          }
        }
        STG(qout, c2[t]) = val;
      }
    }
    FVT_ARRIVE(4 * s + 3);
    if (s < 3 || S.winds) __syncthreads();  // (the cell update read sq / ax / ay)
  };
  pass(std::integral_constant<int, 0>{});
  pass(std::integral_constant<int, 1>{});
  pass(std::integral_constant<int, 2>{});
  pass(std::integral_constant<int, 3>{});
  if (S.winds) pass(std::integral_constant<int, 4>{});
  FVT_STAMP(S.winds ? 20 : 16);
}

}  // namespace FVT_NS
#endif  // FVT_AVAILABLE
