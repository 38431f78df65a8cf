// NonhydrostaticVerticalSolver / NonhydrostaticVerticalSolverCGrid (Fortran Riem_Solver3 / Riem_Solver_c with sim1_solver) as
// ONE kernel, k-cooperative: SIXTEEN LANES PER COLUMN.
// Reference: fv3core/pace/fv3core/stencils/riem_solver3.py:26-321, riem_solver_c.py:21-250, sim1_solver.py:20-141.
//
// Why (round-1 profile, C192 x 79): the five-kernel thread-per-column version moved 1.35 GB for 0.30 GB of algorithmic traffic
// (28 field passes through six workspace fields in the tridiagonal kernel alone) at 0.56 waves per SIMD -- 36 864 columns are
// not enough threads, and neither LDS (a [80][64] array of doubles is 40 KB) nor registers (six sweeps x 80 unrolled levels) can
// hold a whole column per lane.  Here a column is spread over a 16-lane row of a wave, each lane owning L consecutive levels
// (L = 5 for 79 levels); every intermediate of the column lives in the registers of those 16 lanes, nothing but the
// operator's own inputs and outputs touches memory (15 field passes), and 16 x as many threads exist (590 k at C192).
//
// The vertical recurrences become lane-local sweeps over L levels plus a 4-step scan across the 16 lanes of the row:
//   * prefix sums (interface pressures, perturbation pressure, the height rebuild): scan of additions;
//   * Thomas elimination of the two tridiagonal systems: the pivots bet_k = D_k - S_k / bet_{k-1} are a Moebius recurrence
//     (scan of 2 x 2 matrices, rescaled by a power of two at every product), the eliminated right-hand side
//     y_k = (r_k - s_k y_{k-1}) / bet_k and both back-substitutions are affine recurrences (scan of (a, b) pairs);
//   * p1_k = a_k - g_k p1_{k+1} (sim1_solver.py:118-132): affine, backwards.
// A scan only delivers the value ENTERING a lane's block of levels; inside its block every lane then evaluates the
// reference's own expressions level by level (with the pivot's reciprocal formed once per level and used as a factor where
// the reference divides three times), so results differ from the sequential solver only through last-place roundings
// (the systems are diagonally dominant: such perturbations decay).
// The reference's bound for this operator is 5e-6 on every backend (overrides/standard.yaml:49-61); measured on MI355X
// against the numpy oracle: see tests/opchain.py.
//
// Memory access: a wave holds 4 adjacent columns x 16 level blocks, a workgroup 16 adjacent columns = one 128 B line per level
// row, so every line fetched is used completely by the workgroup (through the CU's vector L1 / the L2).
#include "common.h"
#include "kernels.h"

#define RDGAS 287.05
#define GRAV 9.80665
#define RGRAV (1.0 / GRAV)
#define CP_AIR 1004.6
#define KAPPA (RDGAS / CP_AIR)
#define ROW 16  // lanes per column

namespace {

// Value of the lane d places before (up) / after (dn) this one within its 16-lane row; a lane without such a neighbour keeps
// its own value.  A row of the wave is exactly a DPP row, so these are two `v_mov_b32 ... row_shr:d / row_shl:d` each
// (register-to-register, no LDS round trip as in ds_bpermute) -- the scans below are chains of 4-5 such steps.
#ifdef PACE_EMU
__device__ __forceinline__ double up(double v, int d) { return __shfl_up(v, (unsigned)d, ROW); }
__device__ __forceinline__ double dn(double v, int d) { return __shfl_down(v, (unsigned)d, ROW); }
#else
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
  const int nlo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  const int nhi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __longlong_as_double(((long long)nhi << 32) | (long long)(unsigned)nlo);
}
__device__ __forceinline__ double up(double v, int d) {  // d is a compile-time constant after unrolling
  switch (d) {
    case 1: return dpp_move<0x111>(v);  // row_shr:1
    case 2: return dpp_move<0x112>(v);
    case 4: return dpp_move<0x114>(v);
    default: return dpp_move<0x118>(v);
  }
}
__device__ __forceinline__ double dn(double v, int d) {
  switch (d) {
    case 1: return dpp_move<0x101>(v);  // row_shl:1
    case 2: return dpp_move<0x102>(v);
    case 4: return dpp_move<0x104>(v);
    default: return dpp_move<0x108>(v);
  }
}
#endif

// ---- exp / log of the column solvers ---------------------------------------------------------------------------------------
// riem_solver3 takes seven exp / log per level (riem_solver3.py:63-141, sim1_solver.py:118-141), riem_solver_c four; the
// library's versions are ~65 instructions each with their special-case handling.  The arguments here are pressures and
// pressure ratios -- positive, finite, far from the ends of the exponent range -- so: straight range reduction + polynomial,
// no special cases.  Against the numpy oracle (round 5's versions, ~1 ulp): riem_solver3 <= 1.1e-7 (bound 5e-6,
// overrides/standard.yaml:49-61), riem_solver_c 3.3e-15 (bound 5e-14, translate_riem_solver_c.py:33; the library functions
// gave 1.5e-15); tools/riem_check.py prints the errors per variable.
// Round 6: both are evaluated so that the result is rounded ONCE (the leading terms are summed as a pair of doubles, every smaller
// term goes into the low part before the final addition): <= ~0.55 ulp, against ~1 ulp before.  Why it matters: pe = exp(gm * log(..))
// - pm and w's damping heating amplify an ulp of these functions by 1e6 .. 1e9 (the reference's own DynCore bound is 2e-6 on every
// variable, translate_dyncore.py:120); numpy's exp / log are within 0.52 ulp, so a function that is off by one ulp in a quarter of
// its results disagrees with the oracle everywhere, one that is nearly correctly rounded in a few per cent of them.  Measured with
// the oracle itself (exp / log in extended precision, rounded once): diss_estd 1.0e-7, w 8e-9 -- the conditioning of the loop; the
// previous versions: 2.9e-6 / 4.0e-7 (profiles/r06_transcendental_accuracy.txt).
__device__ __forceinline__ double lean_rcp(double d) {  // 1 / d to ~1e-16 (the quotients below carry their own residuals)
#ifdef PACE_EMU
  return 1.0 / d;
#else
  const double r0 = __builtin_amdgcn_rcp(d);
  return fma(fma(-d, r0, 1.0), r0, r0);
#endif
}
__device__ __forceinline__ double lean_log(double x) {
  int e;
  double m = frexp(x, &e);  // [0.5, 1)
  if (m < 0.70710678118654752440) {
    m = m + m;
    e = e - 1;
  }
  // log m = 2 atanh(s), s = (m - 1) / (m + 1), |s| <= 0.1716
  const double f = m - 1.0;           // exact
  const double d = 2.0 + f;           // rounded: m + 1 needs up to two more bits than m has ...
  const double d_lo = f - (d - 2.0);  // ... which are here (Fast2Sum, |2| >= |f|)
  const double r = lean_rcp(d);
  const double s = f * r;
  // s_lo: f / (d + d_lo) - s, from the exact residual of s against d
  const double s_lo = (fma(-s, d, f) - s * d_lo) * r;
  const double z = s * s;
  double p = 2.0 / 21.0;
  p = fma(p, z, 2.0 / 19.0);
  p = fma(p, z, 2.0 / 17.0);
  p = fma(p, z, 2.0 / 15.0);
  p = fma(p, z, 2.0 / 13.0);
  p = fma(p, z, 2.0 / 11.0);
  p = fma(p, z, 2.0 / 9.0);
  p = fma(p, z, 2.0 / 7.0);
  p = fma(p, z, 2.0 / 5.0);
  p = fma(p, z, 2.0 / 3.0);
  // log x = e ln2_hi + 2 s  (a pair: e ln2_hi is exact -- ln2_hi has 32 trailing zero bits -- and the larger of the two unless e = 0)
  //       + [2 s_lo + s z p + e ln2_lo]
  const double de = (double)e;
  const double a = de * 6.93147180369123816490e-01, b = 2.0 * s;
  const double h = a + b;
  const double l = b - (h - a);
  const double low = fma(s * z, p, fma(de, 1.90821492927058770002e-10, 2.0 * s_lo)) + l;
  const double v = h + low;
  // a column that has gone bad (a negative pressure, dz >= 0 upstream) must not come back as a plausible finite number: what the
  // library's log answers there -- NaN below zero, -inf at zero -- for one compare and one select (NaN and +inf pass through)
  return x > 0.0 ? v : (x == 0.0 ? -__builtin_huge_val() : __builtin_nan(""));
}
__device__ __forceinline__ double lean_exp(double x) {
  const double k = rint(x * 1.44269504088896338700e+00);
  const double r = fma(-k, 6.93147180369123816490e-01, x);  // exact (k ln2_hi is, and it lies within a factor of two of x); |r| <= 0.3466
  const double c0 = -k * 1.90821492927058770002e-10;         // the rest of the reduction: exp(x) = 2^k exp(r) exp(c0),
  const double c = fma(0.5 * c0, c0, c0);                    // exp(c0) - 1 = c0 + c0^2 / 2 (|c0| < 4e-7 for |k| < 2000: c0^3 / 6 < 1e-20)
  // exp(r) = 1 + r + r^2 / 2 + r^3 q(r): the first three terms as pairs of doubles, the rest (<= 0.008) and every low part in one sum
  double q = 1.0 / 6227020800.0;  // 1 / 13!
  q = fma(q, r, 1.0 / 479001600.0);
  q = fma(q, r, 1.0 / 39916800.0);
  q = fma(q, r, 1.0 / 3628800.0);
  q = fma(q, r, 1.0 / 362880.0);
  q = fma(q, r, 1.0 / 40320.0);
  q = fma(q, r, 1.0 / 5040.0);
  q = fma(q, r, 1.0 / 720.0);
  q = fma(q, r, 1.0 / 120.0);
  q = fma(q, r, 1.0 / 24.0);
  q = fma(q, r, 1.0 / 6.0);
  const double rr = r * r, rr_lo = fma(r, r, -rr);    // r^2 as a pair
  const double t = (r * rr) * q;                      // r^3 q
  const double h1 = 1.0 + r, l1 = r - (h1 - 1.0);     // 1 + r as a pair (Fast2Sum)
  const double hr = 0.5 * rr;
  const double h = h1 + hr, l2 = hr - (h - h1);       // + r^2 / 2
  const double e1 = h + t;                            // ~exp(r), for the reduction's remainder: exp(x) = 2^k exp(r) (1 + c)
  const double p = h + (fma(c, e1, fma(0.5, rr_lo, t)) + (l1 + l2));  // ONE rounding of a sum whose low part is <= 0.008
  // (k clamped: ldexp with an int from a huge or NaN k would be undefined; +-2000 saturates to inf / 0 like the library's exp)
  const double kc = fmin(fmax(k, -2000.0), 2000.0);
  return x != x ? x : ldexp(p, (int)kc);
}
template <int CG>
__device__ __forceinline__ double col_log(double x) { return lean_log(x); }  // (one place to switch a solver back to the library)
template <int CG>
__device__ __forceinline__ double col_exp(double x) { return lean_exp(x); }

// ---- scan of additions over the row: returns the sum of the values of the lanes BEFORE (fwd) / AFTER (bwd) this one ----
__device__ __forceinline__ double excl_add_fwd(double v, int r) {
#pragma unroll
  for (int d = 1; d < ROW; d <<= 1) {
    const double t = up(v, d);
    if (r >= d) v = t + v;
  }
  const double e = up(v, 1);
  return r == 0 ? 0.0 : e;
}
__device__ __forceinline__ double excl_add_bwd(double v, int r) {
#pragma unroll
  for (int d = 1; d < ROW; d <<= 1) {
    const double t = dn(v, d);
    if (r + d < ROW) v = t + v;
  }
  const double e = dn(v, 1);
  return r == ROW - 1 ? 0.0 : e;
}

// ---- affine maps x -> a x + b ----
struct Aff {
  double a, b;
};
// then(first, second): apply `first`, then `second`
__device__ __forceinline__ Aff then(const Aff& f, const Aff& s) { return Aff{s.a * f.a, s.a * f.b + s.b}; }
// composition of the maps of all lanes before this one, in level order (identity for the first lane)
__device__ __forceinline__ Aff excl_aff_fwd(Aff m, int r) {
#pragma unroll
  for (int d = 1; d < ROW; d <<= 1) {
    const Aff t{up(m.a, d), up(m.b, d)};
    if (r >= d) m = then(t, m);
  }
  const Aff e{up(m.a, 1), up(m.b, 1)};
  return r == 0 ? Aff{1.0, 0.0} : e;
}
// composition of the maps of all lanes after this one, applied from the bottom upwards (identity for the last lane)
__device__ __forceinline__ Aff excl_aff_bwd(Aff m, int r) {
#pragma unroll
  for (int d = 1; d < ROW; d <<= 1) {
    const Aff t{dn(m.a, d), dn(m.b, d)};
    if (r + d < ROW) m = then(t, m);
  }
  const Aff e{dn(m.a, 1), dn(m.b, 1)};
  return r == ROW - 1 ? Aff{1.0, 0.0} : e;
}

// ---- Moebius maps x -> (a x + b) / (c x + d), as 2 x 2 matrices up to a factor ----
struct Mob {
  double a, b, c, d;
};
__device__ __forceinline__ Mob mob_then(const Mob& f, const Mob& s) {
  Mob o{s.a * f.a + s.b * f.c, s.a * f.b + s.b * f.d, s.c * f.a + s.d * f.c, s.c * f.b + s.d * f.d};
  // rescale by a power of two (exact): the pivots of 80 ... 128 levels multiply up to far beyond the double range otherwise
  const double mx = fmax(fmax(fabs(o.a), fabs(o.b)), fmax(fabs(o.c), fabs(o.d)));
  const int e = (mx > 0.0 && mx < 1.0e300) ? ilogb(mx) : 0;
  o.a = ldexp(o.a, -e);
  o.b = ldexp(o.b, -e);
  o.c = ldexp(o.c, -e);
  o.d = ldexp(o.d, -e);
  return o;
}
__device__ __forceinline__ Mob excl_mob_fwd(Mob m, int r) {
#pragma unroll
  for (int d = 1; d < ROW; d <<= 1) {
    const Mob t{up(m.a, d), up(m.b, d), up(m.c, d), up(m.d, d)};
    if (r >= d) m = mob_then(t, m);
  }
  const Mob e{up(m.a, 1), up(m.b, 1), up(m.c, 1), up(m.d, 1)};
  return r == 0 ? Mob{1.0, 0.0, 0.0, 1.0} : e;
}

}  // namespace

// ---- the lanes of a workgroup as movers of whole lines (see k_riem_column) ----
#ifdef PACE_EMU
#define RIEM_OPAQUE(x)
#else
#define RIEM_OPAQUE(x) asm volatile("" : "+v"(x))
#endif
#define RIEM_LP 17
#define RIEM_LDG(p, off) (*(const real*)((const char*)(p) + (off)))
#define RIEM_STG(p, off) (*(real*)((char*)(p) + (off)))
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
// The columns lo .. hi of a row in windows of sixteen that are 128-byte lines of every level.  The partial windows at the two
// ends (13 + 3 columns at C192, whose compute domain starts at i = 3) share workgroup 0 when they fit side by side -- each
// column stays at its place in its line, x = column mod 16 -- so that 192 columns are twelve workgroups, not thirteen: 2 304
// workgroups are exactly three rounds of the 768 the chip holds, 2 496 were a fourth round one quarter full.
struct ColumnWindows {
  int a0, nfull, head, tail;  // first line boundary >= lo; whole windows from there; columns before / after them
  bool merged;
  __host__ __device__ ColumnWindows(int lo, int hi) {
    a0 = (lo + 15) & ~15;
    head = a0 - lo;
    if (hi + 1 < a0) {  // (the whole row inside one window)
      nfull = 0, tail = 0, merged = false;
    } else {
      nfull = (hi + 1 - a0) / 16;
      tail = (hi + 1 - a0) - 16 * nfull;
      merged = head + tail <= 16;
    }
  }
  __host__ __device__ int workgroups() const { return merged ? nfull + (head + tail > 0 ? 1 : 0) : nfull + (head > 0) + (tail > 0); }
  // column of place x of workgroup b (any column of the row, nearest valid one, for places that hold none)
  __host__ __device__ int column(int b, int x, int lo, int hi, bool& valid) const {
    int c;
    if (merged && head + tail > 0) {
      if (b == 0) c = x < tail ? a0 + 16 * nfull + x : a0 - 16 + x;
      else c = a0 + 16 * (b - 1) + x;
      valid = b == 0 ? (x < tail || x >= 16 - head) : true;
    } else {
      c = a0 - (head > 0 ? 16 : 0) + 16 * b + x;
      valid = c >= lo && c <= hi;
    }
    return c < lo ? lo : (c > hi ? hi : c);
  }
};
__device__ __forceinline__ int own_column(const Geo& g, int cg, const ColumnWindows& cw, int bx, int x) {
  bool valid;
  return cw.column(bx, x, g.is - cg, g.ie + cg, valid);
}
template <int L>
struct Mover {
  int xl, xc;        // level (of the first piece) and column within the window
  unsigned row, sk;  // element offset of (column, j) at level 0; level stride
  bool in;           // the column is in the domain
  __device__ __forceinline__ Mover(const Geo& g, int cg, const ColumnWindows& cw, int bx, int j) {
    int tid = threadIdx.x;
    RIEM_OPAQUE(tid);
    xl = tid >> 4;
    xc = tid & 15;
    row = (unsigned)IDX2(g, cw.column(bx, xc, g.is - cg, g.ie + cg, in), j);
    sk = (unsigned)g.sk;
  }
  // byte offset of piece n in a field whose last level is `last` (pieces beyond it repeat it); 32 bits: riem_column_supported
  __device__ __forceinline__ unsigned at(int n, int last) const {
    const int lv = xl + 16 * n;
    return (row + (unsigned)(lv < last ? lv : last) * sk) * (unsigned)sizeof(real);
  }
  __device__ __forceinline__ int slot(int n) const { return (xl + 16 * n) * RIEM_LP + xc; }
  __device__ __forceinline__ int slot_upto(int n, int last) const {
    const int lv = xl + 16 * n;
    return (lv < last ? lv : last) * RIEM_LP + xc;
  }
  __device__ __forceinline__ bool has(int n, int nlev) const { return in && xl + 16 * n < nlev; }
};

// The same with SIXTEEN bytes per lane (round 6, experiment x31; -DRIEM_PAIR=1): a lane takes two neighbouring columns of one level, a
// wave eight whole lines.  This chip's stream rates are 5.2 TB/s for 8-byte and 6.6 TB/s for 16-byte accesses
// (profiles/r05_ubench_streams.txt), and the solver's load and store phases run at the former -- but the kernel is NOT faster with
// pairs (104.5 against 100.1 us in the step, alternating builds on one box): its load phase waits for the first line to arrive, not
// for the lines to stream.  Pairs that are not two neighbouring columns at an even place (the merged head / tail window when the
// tail is odd, clamped places of a partial window) move as two single elements.  Off.
#ifndef RIEM_PAIR
#define RIEM_PAIR 0
#endif
struct alignas(2 * sizeof(real)) RiemPair {
  real x, y;
};
template <int L>
struct PairMover {
  static constexpr int NP = (16 * L + 1 + 31) / 32;  // pieces per lane: levels xl + 32 n
  int xl, x0;               // level (of the first piece), first of the lane's two places in the window
  unsigned row0, row1, sk;  // element offsets of the two columns at level 0; level stride
  bool in0, in1, contig;
  __device__ __forceinline__ PairMover(const Geo& g, int cg, const ColumnWindows& cw, int bx, int j) {
    int tid = threadIdx.x;
    RIEM_OPAQUE(tid);
    xl = tid >> 3;
    x0 = 2 * (tid & 7);
    const int c0 = cw.column(bx, x0, g.is - cg, g.ie + cg, in0), c1 = cw.column(bx, x0 + 1, g.is - cg, g.ie + cg, in1);
    row0 = (unsigned)IDX2(g, c0, j), row1 = (unsigned)IDX2(g, c1, j);
    sk = (unsigned)g.sk;
    contig = c1 == c0 + 1 && (row0 & 1u) == 0 && (sk & 1u) == 0;
  }
  __device__ __forceinline__ int level(int n, int last) const {
    const int lv = xl + 32 * n;
    return lv < last ? lv : last;
  }
  __device__ __forceinline__ void load(const real* f, int n, int last, double& a, double& b) const {
    const unsigned lo = (unsigned)level(n, last) * sk;
    if (contig && ((uintptr_t)f & (2 * sizeof(real) - 1)) == 0) {
      const RiemPair r = *(const RiemPair*)((const char*)f + (size_t)(row0 + lo) * sizeof(real));
      a = r.x, b = r.y;
    } else {
      a = RIEM_LDG(f, (row0 + lo) * (unsigned)sizeof(real));
      b = RIEM_LDG(f, (row1 + lo) * (unsigned)sizeof(real));
    }
  }
  __device__ __forceinline__ void store(real* f, int n, int last, double a, double b) const {
    const unsigned lo = (unsigned)level(n, last) * sk;
    if (contig && in0 && in1 && ((uintptr_t)f & (2 * sizeof(real) - 1)) == 0) {
      RiemPair r;
      r.x = (real)a, r.y = (real)b;
      *(RiemPair*)((char*)f + (size_t)(row0 + lo) * sizeof(real)) = r;
    } else {
      if (in0) RIEM_STG(f, (row0 + lo) * (unsigned)sizeof(real)) = (real)a;
      if (in1) RIEM_STG(f, (row1 + lo) * (unsigned)sizeof(real)) = (real)b;
    }
  }
};

#ifndef RIEM_STAMP
#define RIEM_STAMP(n)  // (tools/census/riem_prof.hip compiles shader-clock stamps in here)
#endif
#ifndef RIEM_WAVES
#define RIEM_WAVES 3  // waves per SIMD the register budget is set for (168 VGPRs: the five-levels-per-lane instance needs 161)
#endif
// CG = 0: riem_solver3 on the compute domain.  CG = 1: riem_solver_c on compute +- 1 (w3 is not modified; outputs gz, pef).
// L = levels per lane; 16 L >= km.
template <int CG, int L>
__global__ void __launch_bounds__(256, RIEM_WAVES)
k_riem_column(Geo g, int last_call, double dt, double ptop, double p_fac, double peln1, double ptk,
              const real* __restrict__ cappa, const real* __restrict__ zs, const real* __restrict__ ws,
              const real* __restrict__ q_con, const real* __restrict__ delp, const real* __restrict__ pt,
              real* __restrict__ delz, real* __restrict__ zh, real* __restrict__ pe, real* __restrict__ ppe,
              real* __restrict__ pk3, real* __restrict__ pk, real* __restrict__ peln, real* __restrict__ w) {
  const int r = threadIdx.x & (ROW - 1);   // level block of this lane
  const int col = threadIdx.x >> 4;        // column within the workgroup
  // A workgroup's sixteen columns are one 128-byte line of every level row it touches (ColumnWindows; windows anchored at the
  // start of the compute domain, i = 3, straddled two lines each, and the line shared with the neighbouring workgroup -- which
  // runs on another XCD -- was fetched by both: measured 1.76 x the algorithmic reads).  Places of a window that hold no column
  // of the domain solve a copy of the nearest one and store nothing.
  const ColumnWindows cw(g.is - CG, g.ie + CG);
  const int j = g.js - CG + blockIdx.y;
  const int km = g.nk;
  // (this lane's own column, for the two surface fields)
#define OWN_COLUMN() IDX2(g, own_column(g, CG, cw, blockIdx.x, col), j)
  // Fields move between memory and the lanes that solve through LDS.  In the solver's own arrangement (a lane = five levels of
  // one column) a wave's load touches sixteen lines for 32 bytes each, and the four waves of the workgroup ask for the same
  // lines at different times: the load and store stages were 60 % of a wave's 74 000 cycles (tools/riem_stage_times.py), and
  // the same kernel with line-shaped addresses (wrong data) ran in 92 us instead of 142.  So: as MOVERS the 256 lanes are
  // 16 levels x 16 columns -- a wave's access is four whole lines -- and level xl + 16 n of column xc is lane (xl, xc)'s n-th
  // piece; as SOLVERS they read their own levels from the LDS copy ([level][17]: the pad keeps the two views off each other's
  // banks).
  constexpr int LP = RIEM_LP, NLV = 16 * L + 1;
  __shared__ double xs_[3][NLV * LP];
  // (the movers' addresses are rebuilt from the lane number at each of the three places they are used -- RIEM_OPAQUE keeps the
  // compiler from holding them in registers across the solver, which sits at the 168-register limit of three waves per SIMD)
#if RIEM_PAIR
#define MOVER() const PairMover<L> M(g, CG, cw, blockIdx.x, j)
  constexpr int NPP = PairMover<L>::NP;
  // fetch: the lane's pieces of a field with nlev levels (pieces beyond the last level repeat it)
#define FETCH(v, f, nlev)                                                                          \
  double v[NPP][2];                                                                                \
  _Pragma("unroll") for (int n = 0; n < NPP; ++n) M.load(f, n, (nlev) - 1, v[n][0], v[n][1])
#define PUT(b, v, nlev)                                                                            \
  do {                                                                                             \
    _Pragma("unroll") for (int n = 0; n < NPP; ++n) {                                              \
      const int lv_ = M.xl + 32 * n;                                                               \
      if (lv_ < NLV) xs_[b][lv_ * LP + M.x0] = v[n][0], xs_[b][lv_ * LP + M.x0 + 1] = v[n][1];     \
    }                                                                                              \
  } while (0)
  // store: the lane's pieces of the LDS copy b go to field f (levels 0 .. nlev-1, the columns of the domain); a piece beyond the
  // field's last level stores that level's value to that level's place once more, the same bits its owner stores
#define STORE(b, f, nlev)                                                                          \
  do {                                                                                             \
    _Pragma("unroll") for (int n = 0; n < NPP; ++n) {                                              \
      const int lv_ = M.level(n, (nlev) - 1);                                                      \
      M.store(f, n, (nlev) - 1, xs_[b][lv_ * LP + M.x0], xs_[b][lv_ * LP + M.x0 + 1]);             \
    }                                                                                              \
  } while (0)
#else
#define MOVER() const Mover<L> M(g, CG, cw, blockIdx.x, j)
  // fetch: the pieces of a field with nlev levels (16 L + 1 at most: the last one is the extra piece of the movers of level 0)
#define FETCH(v, f, nlev)                                                                          \
  double v[L + 1];                                                                                 \
  _Pragma("unroll") for (int n = 0; n < L; ++n) v[n] = RIEM_LDG(f, M.at(n, (nlev) - 1));           \
  v[L] = (16 * L < (nlev)) ? (double)RIEM_LDG(f, M.at(L, (nlev) - 1)) : 0.0
#define PUT(b, v, nlev)                                                                            \
  do {                                                                                             \
    _Pragma("unroll") for (int n = 0; n < L; ++n) xs_[b][M.slot(n)] = v[n];                        \
    if (16 * L < (nlev) && M.xl == 0) xs_[b][M.slot(L)] = v[L];                                    \
  } while (0)
  // store: the mover's pieces of the LDS copy b go to field f (levels 0 .. nlev-1, the columns of the domain)
  // (one test -- of the column -- for all pieces: a piece beyond the field's last level stores that level's value to that level's
  // place once more, the same bits its owner stores; a test per piece cost four scalar instructions and a branch each)
#define STORE(b, f, nlev)                                                                          \
  do {                                                                                             \
    if (M.in) {                                                                                    \
      _Pragma("unroll") for (int n = 0; n < L; ++n) RIEM_STG(f, M.at(n, (nlev) - 1)) = xs_[b][M.slot_upto(n, (nlev) - 1)]; \
      if (16 * L < (nlev) && M.xl == 0) RIEM_STG(f, M.at(L, (nlev) - 1)) = xs_[b][M.slot(L)];      \
    }                                                                                              \
  } while (0)
#endif  // RIEM_PAIR
#define MINE(b, k) xs_[b][(k) * LP + col]
  const int k0 = r * L;
  const double t1g = 2.0 * dt * dt, rdt = 1.0 / dt;
#define LEV(t) (k0 + (t))
#define ON(t) (LEV(t) < km)
#define CL(k) ((k) < km ? (k) : km - 1)
#define DM(x) (CG ? (x) / GRAV : (x) * RGRAV)

  RIEM_STAMP(0);
  // ---------------- loads ----------------
  double d_[L], qc_[L], ca_[L], pt_[L], w1_[L], zh_[L + 1];
  {
    MOVER();
    FETCH(m0, delp, km);
    FETCH(m1, q_con, km);
    FETCH(m2, cappa, km);
    FETCH(m3, pt, km);
    FETCH(m4, w, km);
    FETCH(m5, zh, km + 1);
    PUT(0, m0, km);
    PUT(1, m1, km);
    PUT(2, m2, km);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < L; ++t) {
      d_[t] = MINE(0, CL(LEV(t)));
      qc_[t] = MINE(1, CL(LEV(t)));
      ca_[t] = MINE(2, CL(LEV(t)));
    }
    __syncthreads();
    PUT(0, m3, km);
    PUT(1, m4, km);
    PUT(2, m5, km + 1);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < L; ++t) {
      pt_[t] = MINE(0, CL(LEV(t)));
      w1_[t] = MINE(1, CL(LEV(t)));
    }
#pragma unroll
    for (int t = 0; t <= L; ++t) zh_[t] = MINE(2, LEV(t) <= km ? LEV(t) : km);
  }

  // ---------------- interface pressures (riem_solver3.py:63-81 / riem_solver_c.py:50-66) ----------------
  double sp = 0.0, sg = 0.0;
#pragma unroll
  for (int t = 0; t < L; ++t)
    if (ON(t)) {
      sp = sp + d_[t];
      sg = sg + d_[t] * (1.0 - qc_[t]);
    }
  double pem_[L + 1], pg_[L + 1];
  pem_[0] = ptop + excl_add_fwd(sp, r);
  pg_[0] = ptop + excl_add_fwd(sg, r);
#pragma unroll
  for (int t = 0; t < L; ++t) {
    pem_[t + 1] = pem_[t] + d_[t];
    pg_[t + 1] = pg_[t] + d_[t] * (1.0 - qc_[t]);
  }

  RIEM_STAMP(1);
  // ---------------- per-level quantities of precompute + the first statement of sim1_solver ----------------
  double dm_[L], gm_[L], dz_[L], pm_[L], pe0_[L];
  {
    double lg_prev = CG ? 0.0 : ((LEV(0) == 0) ? peln1 : col_log<CG>(pg_[0]));  // (riem_solver_c takes the log of the ratio instead)
#pragma unroll
    for (int t = 0; t < L; ++t) {
      dm_[t] = DM(d_[t]);
      gm_[t] = 1.0 / (1.0 - ca_[t]);
      dz_[t] = zh_[t + 1] - zh_[t];
      if (CG) {
        pm_[t] = (pg_[t + 1] - pg_[t]) / col_log<CG>(pg_[t + 1] / pg_[t]);
      } else {
        const double lg = col_log<CG>(pg_[t + 1]);
        pm_[t] = (pg_[t + 1] - pg_[t]) / (lg - lg_prev);
        lg_prev = lg;
      }
      pe0_[t] = col_exp<CG>(gm_[t] * col_log<CG>(-dm_[t] / dz_[t] * RDGAS * pt_[t])) - pm_[t];
    }
  }
  RIEM_STAMP(2);
  if (!CG) {
    // pk3 = p_interface ** kappa, and on the last call peln = log p_interface, pk, pe (riem_solver3.py:66-76,136-141).
    // A lane writes the interfaces above its own levels; the lane that owns level km - 1 also writes interface km.
    __syncthreads();  // (everyone has read the inputs)
#pragma unroll
    for (int t = 0; t <= L; ++t) {
      const int k = LEV(t);
      const bool mine = (t < L) ? (k < km) : false;
      const bool bottom = (t >= 1) && (LEV(t - 1) == km - 1);
      if (mine || bottom) {
        double logp, pk3v;
        if (k == 0) {
          logp = peln1;
          pk3v = ptk;
        } else {
          logp = col_log<CG>(pem_[t]);
          pk3v = col_exp<CG>(KAPPA * logp);
        }
        MINE(0, k) = pk3v;
        if (last_call) {
          MINE(1, k) = logp;
          MINE(2, k) = pem_[t];
        }
      }
    }
    __syncthreads();
    MOVER();
    STORE(0, pk3, km + 1);
    if (last_call) {
      STORE(1, peln, km + 1);
      STORE(0, pk, km + 1);
      STORE(2, pe, km + 1);
    }
  }

  RIEM_STAMP(3);
  // neighbours' edge values
  const double dm_next = dn(dm_[0], 1);    // dm of the level after this block
  const double pe0_next = dn(pe0_[0], 1);
  const double gm_prev = up(gm_[L - 1], 1);
  const double dz_prev = up(dz_[L - 1], 1);

  // ---------------- system 1: pp on interfaces 1 .. km (sim1_solver.py:76-98) ----------------
  // row k: pp_k + bb_k pp_{k+1} + g_k pp_{k+2} = dd_k, g_k = dm_k / dm_{k+1} (0 in the last row), bb_k = 2 (1 + g_k)
  double g_[L], bb_[L], dd_[L];
#pragma unroll
  for (int t = 0; t < L; ++t) {
    const double dmn = (t + 1 < L) ? dm_[t + 1] : dm_next;
    const double pen = (t + 1 < L) ? pe0_[t + 1] : pe0_next;
    if (LEV(t) < km - 1) {
      g_[t] = dm_[t] / dmn;
      bb_[t] = 2.0 * (1.0 + g_[t]);
      dd_[t] = 3.0 * (pe0_[t] + g_[t] * pen);
    } else {
      g_[t] = 0.0;
      bb_[t] = 2.0;
      dd_[t] = 3.0 * pe0_[t];
    }
  }
  const double g_before = up(g_[L - 1], 1);  // g of the level before this block
  double gam_[L], x_[L];                      // gam_k (k >= 1) and, at the end, pp_{k+1}
  {
    // pivots: bet_0 = bb_0, bet_k = bb_k - g_{k-1} / bet_{k-1}
    Mob P{1.0, 0.0, 0.0, 1.0};
#pragma unroll
    for (int t = 0; t < L; ++t) {
      if (!ON(t)) continue;
      const double gp = (t == 0) ? g_before : g_[t - 1];
      const Mob Mk = (LEV(t) == 0) ? Mob{0.0, bb_[t], 0.0, 1.0} : Mob{bb_[t], -gp, 1.0, 0.0};
      P = mob_then(P, Mk);
    }
    const Mob E = excl_mob_fwd(P, r);
    // (one division per level: the reciprocal pivot; products with it replace the reference's divisions by the pivot, a
    // last-place difference)
    double rb = (E.c + E.d) / (E.a + E.b);  // 1 / bet of the level before this block: the map applied to 1 (its argument
                                            // is irrelevant: level 0 is a constant map)
    double rb_[L];
#pragma unroll
    for (int t = 0; t < L; ++t) {
      if (LEV(t) == 0) {
        gam_[t] = 0.0;
        rb = 1.0 / bb_[t];
      } else {
        const double gp = (t == 0) ? g_before : g_[t - 1];
        gam_[t] = gp * rb;
        rb = 1.0 / (bb_[t] - gam_[t]);
      }
      rb_[t] = rb;
    }
    // y_k = (dd_k - y_{k-1}) / bet_k, y_0 = dd_0 / bet_0
    Aff A{1.0, 0.0};
#pragma unroll
    for (int t = 0; t < L; ++t) {
      if (!ON(t)) continue;
      A = then(A, Aff{LEV(t) == 0 ? 0.0 : -rb_[t], dd_[t] * rb_[t]});
    }
    const Aff EA = excl_aff_fwd(A, r);
    double y = EA.b;  // applied to y_{-1} = 0
#pragma unroll
    for (int t = 0; t < L; ++t) {
      y = (LEV(t) == 0) ? dd_[t] * rb_[t] : (dd_[t] - y) * rb_[t];
      x_[t] = y;
    }
    // back substitution: pp_{k+1} = y_k - gam_{k+1} pp_{k+2}; the last row keeps y
    const double gam_after = dn(gam_[0], 1);
    Aff B{1.0, 0.0};
#pragma unroll
    for (int t = L - 1; t >= 0; --t) {
      if (!ON(t)) continue;
      const double gn = (t + 1 < L) ? gam_[t + 1] : gam_after;
      B = then(B, Aff{LEV(t) == km - 1 ? 0.0 : -gn, x_[t]});
    }
    const Aff EB = excl_aff_bwd(B, r);
    double xn = EB.b;  // applied to 0
#pragma unroll
    for (int t = L - 1; t >= 0; --t) {
      if (!ON(t)) continue;
      const double gn = (t + 1 < L) ? gam_[t + 1] : gam_after;
      xn = (LEV(t) == km - 1) ? x_[t] : x_[t] - gn * xn;
      x_[t] = xn;
    }
  }
  RIEM_STAMP(4);
  // pp on the interfaces of this block: ppi_[t] = pp_{k0+t}
  double ppi_[L + 1];
  {
    const double before = up(x_[L - 1], 1);
    ppi_[0] = (r == 0) ? 0.0 : before;
#pragma unroll
    for (int t = 0; t < L; ++t) ppi_[t + 1] = x_[t];
  }

  // ---------------- aa on interfaces 1 .. km-1 (sim1_solver.py:92-98) ----------------
  double aa_[L + 1];
#pragma unroll
  for (int t = 0; t < L; ++t) {
    const double gmp = (t == 0) ? gm_prev : gm_[t - 1];
    const double dzp = (t == 0) ? dz_prev : dz_[t - 1];
    aa_[t] = (LEV(t) >= 1 && LEV(t) <= km - 1) ? t1g * 0.5 * (gmp + gm_[t]) / (dzp + dz_[t]) * (pem_[t] + ppi_[t]) : 0.0;
  }
  aa_[L] = dn(aa_[0], 1);

  RIEM_STAMP(5);
  // ---------------- system 2: w (sim1_solver.py:99-117) ----------------
  double wn_[L];
  {
    const double wsv = ws[OWN_COLUMN()];
    // the surface term replaces aa_{k+1} in the last row
    double p1s_[L];
#pragma unroll
    for (int t = 0; t < L; ++t)
      p1s_[t] = (LEV(t) == km - 1) ? t1g * gm_[t] / dz_[t] * (pem_[t + 1] + ppi_[t + 1]) : aa_[t + 1];
    Mob P{1.0, 0.0, 0.0, 1.0};
#pragma unroll
    for (int t = 0; t < L; ++t) {
      if (!ON(t)) continue;
      const double diag = dm_[t] - (aa_[t] + p1s_[t]);
      const Mob Mk = (LEV(t) == 0) ? Mob{0.0, diag, 0.0, 1.0} : Mob{diag, -(aa_[t] * aa_[t]), 1.0, 0.0};
      P = mob_then(P, Mk);
    }
    const Mob E = excl_mob_fwd(P, r);
    double rb = (E.c + E.d) / (E.a + E.b);
    double rb_[L], gam2_[L], rhs_[L];
#pragma unroll
    for (int t = 0; t < L; ++t) {
      if (LEV(t) == 0) {
        gam2_[t] = 0.0;
        rb = 1.0 / (dm_[t] - aa_[t + 1]);
        rhs_[t] = dm_[t] * w1_[t] + dt * ppi_[t + 1];
      } else {
        gam2_[t] = aa_[t] * rb;
        rb = 1.0 / (dm_[t] - (aa_[t] + p1s_[t] + aa_[t] * gam2_[t]));
        rhs_[t] = (LEV(t) == km - 1) ? dm_[t] * w1_[t] + dt * (ppi_[t + 1] - ppi_[t]) - p1s_[t] * wsv
                                     : dm_[t] * w1_[t] + dt * (ppi_[t + 1] - ppi_[t]);
      }
      rb_[t] = rb;
    }
    Aff A{1.0, 0.0};
#pragma unroll
    for (int t = 0; t < L; ++t) {
      if (!ON(t)) continue;
      A = then(A, Aff{LEV(t) == 0 ? 0.0 : -aa_[t] * rb_[t], rhs_[t] * rb_[t]});
    }
    const Aff EA = excl_aff_fwd(A, r);
    double y = EA.b;
#pragma unroll
    for (int t = 0; t < L; ++t) {
      y = (LEV(t) == 0) ? rhs_[t] * rb_[t] : (rhs_[t] - aa_[t] * y) * rb_[t];
      wn_[t] = y;
    }
    const double gam_after = dn(gam2_[0], 1);
    Aff B{1.0, 0.0};
#pragma unroll
    for (int t = L - 1; t >= 0; --t) {
      if (!ON(t)) continue;
      const double gn = (t + 1 < L) ? gam2_[t + 1] : gam_after;
      B = then(B, Aff{LEV(t) == km - 1 ? 0.0 : -gn, wn_[t]});
    }
    const Aff EB = excl_aff_bwd(B, r);
    double xn = EB.b;
#pragma unroll
    for (int t = L - 1; t >= 0; --t) {
      if (!ON(t)) continue;
      const double gn = (t + 1 < L) ? gam2_[t + 1] : gam_after;
      xn = (LEV(t) == km - 1) ? wn_[t] : wn_[t] - gn * xn;
      wn_[t] = xn;
    }
  }

  RIEM_STAMP(6);
  // ---------------- perturbation pressure on interfaces (sim1_solver.py:112-117) ----------------
  double pei_[L + 2];
  {
    double s = 0.0;
#pragma unroll
    for (int t = 0; t < L; ++t)
      if (ON(t)) s = s + dm_[t] * (wn_[t] - w1_[t]) * rdt;
    pei_[0] = excl_add_fwd(s, r);
#pragma unroll
    for (int t = 0; t < L; ++t) pei_[t + 1] = ON(t) ? pei_[t] + dm_[t] * (wn_[t] - w1_[t]) * rdt : pei_[t];
    pei_[L + 1] = dn(pei_[1], 1);
  }

  RIEM_STAMP(7);
  // ---------------- p1 (sim1_solver.py:118-132), backwards ----------------
  double p1_[L];
  {
    double a_[L];
    Aff B{1.0, 0.0};
#pragma unroll
    for (int t = L - 1; t >= 0; --t) {
      if (!ON(t)) continue;
      if (LEV(t) == km - 1) {
        a_[t] = (pei_[t] + 2.0 * pei_[t + 1]) * 1.0 / 3.0;
        B = then(B, Aff{0.0, a_[t]});
      } else {
        a_[t] = (pei_[t] + bb_[t] * pei_[t + 1] + g_[t] * pei_[t + 2]) * 1.0 / 3.0;
        B = then(B, Aff{-g_[t], a_[t]});
      }
    }
    const Aff EB = excl_aff_bwd(B, r);
    double pn = EB.b;
#pragma unroll
    for (int t = L - 1; t >= 0; --t) {
      if (!ON(t)) continue;
      pn = (LEV(t) == km - 1) ? a_[t] : a_[t] - g_[t] * pn;
      p1_[t] = pn;
    }
  }

  RIEM_STAMP(8);
  // ---------------- new layer thickness (sim1_solver.py:133-141) and the height rebuild ----------------
  double dzn_[L];
  double sdz = 0.0;
#pragma unroll
  for (int t = 0; t < L; ++t) {
    // NB: the reference tests p_fac * delta_mass (sim1_solver.py:134), kept as is
    const double maxp = (p_fac * dm_[t] > p1_[t] + pm_[t]) ? p_fac * pm_[t] : p1_[t] + pm_[t];
    dzn_[t] = -dm_[t] * RDGAS * pt_[t] * col_exp<CG>((ca_[t] - 1.0) * col_log<CG>(maxp));
    if (ON(t)) sdz = sdz + (CG ? dzn_[t] * GRAV : dzn_[t]);
  }
  RIEM_STAMP(9);
  {
    // zh_km = zs, zh_k = zh_{k+1} - delz_k (riem_solver3.py:142-145); gz_km = hs, gz_k = gz_{k+1} - dz_k g (riem_solver_c.py:117-123)
    double z = zs[OWN_COLUMN()] - excl_add_bwd(sdz, r);  // height of the interface below this block
    // (interface km is written by the lane whose block ends with level km - 1 or, if km is a multiple of L, by nobody else)
    __syncthreads();  // (the movers have taken pk3 ... / everyone has read the inputs)
#pragma unroll
    for (int t = L - 1; t >= 0; --t) {
      if (!ON(t)) continue;
      if (LEV(t) == km - 1) MINE(0, km) = z;
      z = z - (CG ? dzn_[t] * GRAV : dzn_[t]);
      MINE(0, LEV(t)) = z;
    }
  }
#pragma unroll
  for (int t = 0; t < L; ++t) {
    if (!ON(t)) continue;
    const int k = LEV(t);
    if (CG) {
      // finalize (riem_solver_c.py:91-116): pef = pe + pem below the top
      MINE(1, k) = (k == 0) ? ptop : pei_[t] + pem_[t];
      if (k == km - 1) MINE(1, km) = pei_[t + 1] + pem_[t + 1];
    } else {
      MINE(1, k) = pei_[t];
      if (k == km - 1) MINE(1, km) = pei_[t + 1];
      MINE(2, k) = dzn_[t];
    }
  }
  __syncthreads();
  MOVER();
  STORE(0, zh, km + 1);
  STORE(1, ppe, km + 1);
  if (!CG) {
    STORE(2, delz, km);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < L; ++t)
      if (ON(t)) MINE(2, LEV(t)) = wn_[t];
    __syncthreads();
    STORE(2, w, km);
  }
  RIEM_STAMP(10);
#undef MOVER
#undef OWN_COLUMN
#undef FETCH
#undef PUT
#undef STORE
#undef MINE
#undef LEV
#undef ON
#undef CL
#undef DM
}

template <int CG>
static int launch_column(const Geo& g, int last_call, double dt, double ptop, double p_fac, const real* cappa, const real* zs,
                         const real* ws, const real* q_con, const real* delp, const real* pt, real* delz, real* zh,
                         real* pe, real* ppe, real* pk3, real* pk, real* peln, real* w, hipStream_t st) {
  const double peln1 = log(ptop);
  const double ptk = exp(KAPPA * peln1);
  const int ncol = g.n + 2 * CG;
  const dim3 grid(ColumnWindows(g.is - CG, g.ie + CG).workgroups(), ncol), block(256);
#define GO(L)                                                                                                                  \
  hipLaunchKernelGGL((k_riem_column<CG, L>), grid, block, 0, st, g, last_call, dt, ptop, p_fac, peln1, ptk, cappa, zs, ws, q_con, \
                     delp, pt, delz, zh, pe, ppe, pk3, pk, peln, w)
  const int km = g.nk;
  if (km < 2) return PACE_ERR_UNSUPPORTED;
  if (km <= 16 * 2) GO(2);
  else if (km <= 16 * 4) GO(4);
  else if (km <= 16 * 5) GO(5);
  else if (km <= 16 * 6) GO(6);
  else if (km <= 16 * 8) GO(8);
  else return PACE_ERR_UNSUPPORTED;
#undef GO
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}

// the column solvers accept up to 128 layers in this form (the thread-per-column kernels of k_riem3.hip serve beyond that)
bool riem_column_supported(const Geo& g) {
  // (the movers of k_riem_column address a field with 32-bit byte offsets)
  return g.nk >= 2 && g.nk <= 128 && (double)g.sk * (g.nk + 1) * sizeof(real) < 4.0e9;
}

int launch_riem_solver3_column(const Geo& g, int last_call, double dt, const real* cappa, double ptop, const real* zs,
                               const real* wsd, real* delz, const real* q_con, const real* delp, const real* pt,
                               real* zh, real* pe, real* ppe, real* pk3, real* pk, real* peln, real* w,
                               double p_fac, hipStream_t st) {
  return launch_column<0>(g, last_call, dt, ptop, p_fac, cappa, zs, wsd, q_con, delp, pt, delz, zh, pe, ppe, pk3, pk, peln, w, st);
}

int launch_riem_solver_c_column(const Geo& g, double dt2, const real* cappa, double ptop, const real* hs, const real* ws3,
                                const real* ptc, const real* q_con, const real* delpc, real* gz, real* pef,
                                const real* w3, double p_fac, hipStream_t st) {
  // the C-grid solver does not return w: the kernel reads w3 and writes nothing back (CG = 1 stores only gz and pef)
  return launch_column<1>(g, 0, dt2, ptop, p_fac, cappa, hs, ws3, q_con, delpc, ptc, nullptr, gz, nullptr, pef, nullptr, nullptr,
                          nullptr, const_cast<real*>(w3), st);
}
