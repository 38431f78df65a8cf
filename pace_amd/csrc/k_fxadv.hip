// FiniteVolumeFluxPrep (Fortran fxadv) -- contravariant C-grid winds, Courant numbers and area
// fluxes.  Reference: fv3core/pace/fv3core/stencils/fxadv.py:10-661 (8 stencils, each a full
// pass over two to four 3-D fields).  Here: one thin launch for the frame of the plane (the interior formula there, then the
// edge and corner rules, which only touch O(N) points), and one streaming pass that forms the interior's contravariant winds
// in registers and writes the fluxes.  HBM-bound: 2 reads + 4 writes (+ cx, cy read-modify-write in d_sw) are algorithmic.
#include "common.h"
#include "kernels.h"

// Interior / frame split (launch_fxadv `part`): the points of the box [is+2, ie-2] x [js+2, je-2] read no halo value of uc / vc
// and are touched by none of the edge / corner stages, so their share of stage A and of the fluxes can run while the uc / vc
// halo exchange is in flight (dyn_core.py:817-820); the frame (everything else, incl. the whole halo region) runs after it.
typedef SplitBox FxBox;

__device__ __forceinline__ double contra(double v1, double v2, double cosa, double rsin2) {
  return (v1 - v2 * cosa) * rsin2;  // d2a2c_vect.py:225-281
}

// stage A: main_uc_vc_contra (fxadv.py:10-48) + uc_contra_y_edge (:51-77)
__device__ __forceinline__ void fx_main_point(const Geo& g, const Met& m, const real* __restrict__ uc, const real* __restrict__ vc,
                                              real* __restrict__ ut, real* __restrict__ vt, int i, int j, int k) {
  if (i > g.ni - 2 || j > g.nj - 2) return;  // domain_full = N+6 points
  const long c = IDX3(g, i, j, k);
  const long c2 = IDX2(g, i, j);
  if (i == g.is || i == g.ie + 1) {
    const double u = uc[c];
    ut[c] = (u > 0.0) ? (u / m.sin_sg3[c2 - 1]) : (u / m.sin_sg1[c2]);
  } else if (i >= g.is - 1 && i <= g.ie + 2 && !(j == g.js - 1 || j == g.js || j == g.je || j == g.je + 1)) {
    const double vb = 0.25 * (vc[c - 1] + vc[c] + vc[c - 1 + g.sj] + vc[c + g.sj]);
    ut[c] = contra(uc[c], vb, m.cosa_u[c2], m.rsin_u[c2]);
  }
  if (j >= g.js - 1 && j <= g.je + 2) {
    const double ub = 0.25 * (uc[c - g.sj] + uc[c + 1 - g.sj] + uc[c] + uc[c + 1]);
    vt[c] = contra(vc[c], ub, m.cosa_v[c2], m.rsin_v[c2]);
  }
}

// stage B: vc_contra_y_edge (:80-125) then vc_contra_x_edge (:128-145); touches edge strips only
// (stages B, C, D touch O(N) points per level: they are launched on strips that contain those points -- `R` -- instead of on
// whole planes, where 99 % of the threads only found out that they had nothing to do: 9 + 9 + 6 us -> see profiles)
__device__ __forceinline__ void fx_vt_edges_point(const Geo& g, const Met& m, const real* vc, const real* ut, real* vt, int i, int j,
                                                  int k) {
  if (i > g.ni - 2 || j > g.nj - 2) return;
  const long c = IDX3(g, i, j, k);
  const long c2 = IDX2(g, i, j);
  if (j == g.js || j == g.je + 1) {
    const double v = vc[c];
    vt[c] = (v > 0.0) ? (v / m.sin_sg4[c2 - g.sj]) : (v / m.sin_sg2[c2]);
    return;
  }
  const bool strip = (i == g.is - 1 || i == g.is || i == g.ie || i == g.ie + 1);
  if (strip && j >= g.js + 2 && j <= g.je - 1) {
    const double ub = 0.25 * (ut[c - g.sj] + ut[c + 1 - g.sj] + ut[c] + ut[c + 1]);
    vt[c] = contra(vc[c], ub, m.cosa_v[c2], 1.0);
  }
}

// stage C: uc_contra_x_edge (:148-180), uc_contra_corners (:183-300), vc_contra_corners (:303-404)
__device__ __forceinline__ void fx_ut_edges_corners_point(const Geo& g, const Met& m, const real* uc, const real* vc, real* ut,
                                                          const real* vt, int i, int j, int k) {
  if (i < 1 || j < 1 || i > g.ni - 2 || j > g.nj - 2) return;
  const long c = IDX3(g, i, j, k);
  const long c2 = IDX2(g, i, j);
  const int sj = g.sj;
  const bool rows = (j == g.js - 1 || j == g.js || j == g.je || j == g.je + 1);
  if (rows && i >= g.is + 2 && i <= g.ie - 1) {
    const double vb = 0.25 * (vt[c - 1] + vt[c] + vt[c - 1 + sj] + vt[c + sj]);
    ut[c] = contra(uc[c], vb, m.cosa_u[c2], 1.0);
    return;
  }
  // The corner formulas read ut/vt only at points no corner formula writes, so
  // the in-place update equals the reference's aliased uc_contra/uc_contra_copy call.
  if (rows && (i == g.is + 1 || i == g.ie)) {
    const double cu = m.cosa_u[c2];
    double damp, val;
    if (i == g.is + 1 && (j == g.js - 1 || j == g.je)) {
      const double cv = m.cosa_v[c2 - 1];
      damp = 1.0 / (1.0 - 0.0625 * cu * cv);
      val = (uc[c] - 0.25 * cu * (vt[c - 1 + sj] + vt[c + sj] + vt[c] + vc[c - 1] -
                                  0.25 * cv * (ut[c - 1] + ut[c - 1 - sj] + ut[c - sj]))) * damp;
    } else if (i == g.is + 1) {
      const double cv = m.cosa_v[c2 - 1 + sj];
      damp = 1.0 / (1.0 - 0.0625 * cu * cv);
      val = (uc[c] - 0.25 * cu * (vt[c - 1] + vt[c] + vt[c + sj] + vc[c - 1 + sj] -
                                  0.25 * cv * (ut[c - 1] + ut[c - 1 + sj] + ut[c + sj]))) * damp;
    } else if (j == g.js - 1 || j == g.je) {
      const double cv = m.cosa_v[c2];
      damp = 1.0 / (1.0 - 0.0625 * cu * cv);
      val = (uc[c] - 0.25 * cu * (vt[c + sj] + vt[c - 1 + sj] + vt[c - 1] + vc[c] -
                                  0.25 * cv * (ut[c + 1] + ut[c + 1 - sj] + ut[c - sj]))) * damp;
    } else {
      const double cv = m.cosa_v[c2 + sj];
      damp = 1.0 / (1.0 - 0.0625 * cu * cv);
      val = (uc[c] - 0.25 * cu * (vt[c] + vt[c - 1] + vt[c - 1 + sj] + vc[c + sj] -
                                  0.25 * cv * (ut[c + 1] + ut[c + 1 + sj] + ut[c + sj]))) * damp;
    }
    ut[c] = val;
    return;
  }
}

__device__ __forceinline__ void fx_vt_corners_point(const Geo& g, const Met& m, const real* uc, const real* vc, const real* ut,
                                                    real* vt, int i, int j, int k) {
  if (i < 1 || j < 1 || i > g.ni - 2 || j > g.nj - 2) return;
  const bool cols = (i == g.is - 1 || i == g.is || i == g.ie || i == g.ie + 1);
  if (!cols || !(j == g.js + 1 || j == g.je)) return;
  const long c = IDX3(g, i, j, k);
  const long c2 = IDX2(g, i, j);
  const int sj = g.sj;
  const double cv = m.cosa_v[c2];
  double damp, val;
  if (j == g.js + 1 && (i == g.is - 1 || i == g.ie)) {
    const double cu = m.cosa_u[c2 - sj];
    damp = 1.0 / (1.0 - 0.0625 * cu * cv);
    val = (vc[c] - 0.25 * cv * (ut[c + 1 - sj] + ut[c + 1] + ut[c] + uc[c - sj] -
                                0.25 * cu * (vt[c - sj] + vt[c - 1 - sj] + vt[c - 1]))) * damp;
  } else if (j == g.js + 1) {
    const double cu = m.cosa_u[c2 + 1 - sj];
    damp = 1.0 / (1.0 - 0.0625 * cu * cv);
    val = (vc[c] - 0.25 * cv * (ut[c - sj] + ut[c] + ut[c + 1] + uc[c + 1 - sj] -
                                0.25 * cu * (vt[c - sj] + vt[c + 1 - sj] + vt[c + 1]))) * damp;
  } else if (i == g.ie + 1 || i == g.is) {
    const double cu = m.cosa_u[c2 + 1];
    damp = 1.0 / (1.0 - 0.0625 * cu * cv);
    val = (vc[c] - 0.25 * cv * (ut[c] + ut[c - sj] + ut[c + 1 - sj] + uc[c + 1] -
                                0.25 * cu * (vt[c + sj] + vt[c + 1 + sj] + vt[c + 1]))) * damp;
  } else {
    const double cu = m.cosa_u[c2];
    damp = 1.0 / (1.0 - 0.0625 * cu * cv);
    val = (vc[c] - 0.25 * cv * (ut[c + 1] + ut[c + 1 - sj] + ut[c - sj] + uc[c] -
                                0.25 * cu * (vt[c + sj] + vt[c - 1 + sj] + vt[c - 1]))) * damp;
  }
  vt[c] = val;
}

// fxadv_fluxes_stencil (:436-486).  A thread takes FX_CH consecutive levels of its point: the metric values a point needs
// (rdxa, rdya, dx, dy, the sin_sg of either side, cosa / rsin of its two faces) are the same on every level and are loaded once.
// Round 5: a point of the interior box forms its contravariant winds here, from uc and vc (stage A's interior formula, the only
// one that applies there), instead of reading them back from ut / vt: the two 3-D fields ut and vt are no longer written and
// read on 90 % of the plane (4 of 12 field passes of fxadv).  Frame points read the ut / vt the edge kernel below left there.
// contra_out: also store the winds formed here (the stand-alone entry pace_fxadv returns ut and vt whole).
// FX_CH levels per thread where that still leaves the chip several workgroups per compute unit (C192: 149 blocks x 10 chunks), FX_CH_SMALL
// on small tiles (C48 at eight levels per thread: 120 workgroups on 256 compute units, each walking its levels one dependent load round
// after the other -- 21.6 us for 15 MB).
#ifndef FX_CH
#define FX_CH 8
#endif
#ifndef FX_CH_SMALL
#define FX_CH_SMALL 2
#endif
template <int FXC>
__global__ void __launch_bounds__(256) k_fxadv_fluxes(Geo g, Met m, const real* __restrict__ uc, const real* __restrict__ vc,
                                                      real* __restrict__ ut, real* __restrict__ vt, real* __restrict__ crx,
                                                      real* __restrict__ cry, real* __restrict__ xfx,
                                                      real* __restrict__ yfx, double dt,
                                                      real* __restrict__ cx_acc, real* __restrict__ cy_acc, FxBox box, int contra_out) {
  // cx_acc / cy_acc (optional): the Courant-number half of d_sw's flux_capacitor (d_sw.py:33-60), cx += crx, cy += cry,
  // done where crx / cry are produced
  // (flattened rows; 64 x 4 patches, which halve the re-reads of the rows above and below, measured 64 us against 61)
#ifdef PACE_EMU
  const int bxp = (int)blockIdx.x, chunk = (int)blockIdx.y;
#else
  // Workgroups are dealt to the eight XCDs round-robin in launch order, and a point reads the rows above and below its own: with
  // the plain order those rows belong to workgroups on OTHER XCDs and every XCD's L2 fetches its own copy (counted 324 MB for 233
  // algorithmic).  Here XCD x takes the x-th eighth of the plane's blocks (contiguous rows) of every chunk of levels: workgroup
  // 8 q + x of a chunk is block x S + q, S = ceil(blocks / 8); only the seams between the eighths are fetched twice.
  const int nbx = (int)gridDim.x, seg = (nbx + 7) / 8;
  const int lin = (int)blockIdx.x + nbx * (int)blockIdx.y;
  const int per_chunk = 8 * seg;  // (the grid is launched with 8 * seg blocks per chunk: see launch_fxadv)
  const int chunk = lin / per_chunk, r = lin - chunk * per_chunk;
  const int bxp = (r & 7) * seg + (r >> 3);
  if (bxp >= (int)((((long)g.sj * g.nj) + 255) / 256)) return;
#endif
  const long p = (long)bxp * 256 + threadIdx.x;
  const int j = (int)(p / g.sj);
  const int i = (int)(p - (long)j * g.sj);
  const int k0 = chunk * FXC;
  if (j >= g.nj || i >= g.ni) return;
  if (i > g.ni - 2 || j > g.nj - 2 || box.skip(i, j)) return;
  const long c2 = IDX2(g, i, j);
  const int sj = g.sj;
  const bool inner = i >= box.i0 && i <= box.i1 && j >= box.j0 && j <= box.j1;  // then do_x and do_y both hold
  const bool do_x = i >= g.is && i <= g.ie + 1, do_y = j >= g.js && j <= g.je + 1;
  double rdxa_m = 0.0, rdxa_0 = 0.0, dy = 0.0, sg3_m = 0.0, sg1_0 = 0.0;
  double rdya_m = 0.0, rdya_0 = 0.0, dx = 0.0, sg4_m = 0.0, sg2_0 = 0.0;
  double cosa_u = 0.0, rsin_u = 0.0, cosa_v = 0.0, rsin_v = 0.0;
  if (do_x) rdxa_m = m.rdxa[c2 - 1], rdxa_0 = m.rdxa[c2], dy = m.dy[c2], sg3_m = m.sin_sg3[c2 - 1], sg1_0 = m.sin_sg1[c2];
  if (do_y) rdya_m = m.rdya[c2 - sj], rdya_0 = m.rdya[c2], dx = m.dx[c2], sg4_m = m.sin_sg4[c2 - sj], sg2_0 = m.sin_sg2[c2];
  if (inner) cosa_u = m.cosa_u[c2], rsin_u = m.rsin_u[c2], cosa_v = m.cosa_v[c2], rsin_v = m.rsin_v[c2];
#pragma unroll
  for (int t = 0; t < FXC; ++t) {
    const int k = k0 + t;
    if (k >= g.nk) break;
    const long c = c2 + (long)k * g.sk;
    double u = 0.0, v = 0.0;
    if (inner) {
      const double uc0 = uc[c], vc0 = vc[c];
      const double vb = 0.25 * (vc[c - 1] + vc0 + vc[c - 1 + sj] + vc[c + sj]);
      const double ub = 0.25 * (uc[c - sj] + uc[c + 1 - sj] + uc0 + uc[c + 1]);
      u = contra(uc0, vb, cosa_u, rsin_u);
      v = contra(vc0, ub, cosa_v, rsin_v);
      if (contra_out) ut[c] = u, vt[c] = v;
    } else {
      if (do_x) u = ut[c];
      if (do_y) v = vt[c];
    }
    if (do_x) {
      double cr;
      if (u > 0.0) {
        cr = dt * u * rdxa_m;
        xfx[c] = dy * dt * u * sg3_m;
      } else {
        cr = dt * u * rdxa_0;
        xfx[c] = dy * dt * u * sg1_0;
      }
      crx[c] = cr;
      if (cx_acc) cx_acc[c] = cx_acc[c] + cr;
    }
    if (do_y) {
      double cr;
      if (v > 0.0) {
        cr = dt * v * rdya_m;
        yfx[c] = dx * dt * v * sg4_m;
      } else {
        cr = dt * v * rdya_0;
        yfx[c] = dx * dt * v * sg2_0;
      }
      cry[c] = cr;
      if (cy_acc) cy_acc[c] = cy_acc[c] + cr;
    }
  }
}

// Stage A on the frame: a launch over its four rectangles.  (As a first stage of the one-workgroup-per-level kernel below it
// cost 15 us: four rounds of dependent loads on 79 compute units.)
__global__ void __launch_bounds__(256) k_fxadv_frame(Geo g, Met m, const real* __restrict__ uc, const real* __restrict__ vc,
                                                    real* __restrict__ ut, real* __restrict__ vt, Regions R) {
  REGION_POINT(R);
  (void)interior;
  fx_main_point(g, m, uc, vc, ut, vt, i, j, k);
}

// Stages B, C, D in ONE launch: a workgroup of 1024 threads per level walks the strips of each stage (one or two points per
// thread); a stage reads what the previous one wrote at other points of the same level, so the stages are separated by a workgroup
// barrier (writes to global memory made visible to the workgroup by the fence).  The frame is everything outside the interior
// box [is+2, ie-2] x [js+2, je-2] (11 % of the plane at C192): every point a stage B, C or D formula reads or writes lies in it,
// and so does every ut / vt d_sw's kinetic energy reads at the tile edges (rows js-2 .. js+1 and je-1 .. je+2 of ut, the same
// columns of vt: k_dsw.hip kinetic_energy_point).
struct FxStrips {
  Regions b, c, d;
};
__device__ __forceinline__ bool strip_point(const Regions& R, int p, int& i, int& j) {
  for (int q = 0; q < R.n; ++q) {
    const int w = R.ie[q] - R.ib[q] + 1, h = R.je[q] - R.jb[q] + 1;
    if (p < w * h) {
      j = R.jb[q] + p / w;
      i = R.ib[q] + p % w;
      return true;
    }
    p -= w * h;
  }
  return false;
}
__device__ __forceinline__ int strip_count(const Regions& R) {
  int n = 0;
  for (int q = 0; q < R.n; ++q) n += (R.ie[q] - R.ib[q] + 1) * (R.je[q] - R.jb[q] + 1);
  return n;
}
__global__ void __launch_bounds__(1024) k_fxadv_edges(Geo g, Met m, const real* __restrict__ uc, const real* __restrict__ vc,
                                                     real* ut, real* vt, FxStrips S) {
  const int k = (int)blockIdx.x;
  const int tid = (int)threadIdx.x;
  int i, j;
  for (int p = tid, n = strip_count(S.b); p < n; p += 1024)
    if (strip_point(S.b, p, i, j)) fx_vt_edges_point(g, m, vc, ut, vt, i, j, k);
  __threadfence_block();
  __syncthreads();
  for (int p = tid, n = strip_count(S.c); p < n; p += 1024)
    if (strip_point(S.c, p, i, j)) fx_ut_edges_corners_point(g, m, uc, vc, ut, vt, i, j, k);
  __threadfence_block();
  __syncthreads();
  for (int p = tid, n = strip_count(S.d); p < n; p += 1024)
    if (strip_point(S.d, p, i, j)) fx_vt_corners_point(g, m, uc, vc, ut, vt, i, j, k);
}

// Stage A and the fluxes of the frame inside ONE workgroup per level (k_fxadv_fused): a thread has several points of the frame, and a
// point's formula is a round trip to memory.  Load everything a batch of points needs first, then compute and store: one round trip per
// batch instead of one per point (the workgroup's stages are the launch's critical path at C192: 85 us one point after the other).
struct FxFramePointA {
  long c;
  bool xedge, umain, vmain;
  double u0, u_s, u1_s, u1, v0, v_1, v_1s, vs, s3, s1, cu, ru, cv, rv;
  __device__ __forceinline__ void load(const Geo& g, const Met& m, const real* __restrict__ uc, const real* __restrict__ vc, bool on, int i,
                                       int j, int k) {
    // (fx_main_point's cases, loads only)
    const long c2 = IDX2(g, i, j);
    c = IDX3(g, i, j, k);
    const int sj = g.sj;
    xedge = on && (i == g.is || i == g.ie + 1);
    umain = on && !xedge && i >= g.is - 1 && i <= g.ie + 2 && !(j == g.js - 1 || j == g.js || j == g.je || j == g.je + 1);
    vmain = on && j >= g.js - 1 && j <= g.je + 2;
    u0 = u_s = u1_s = u1 = v0 = v_1 = v_1s = vs = s3 = s1 = cu = ru = cv = rv = 0.0;
    if (xedge || umain || vmain) u0 = uc[c];
    if (umain || vmain) v0 = vc[c];
    if (xedge) s3 = m.sin_sg3[c2 - 1], s1 = m.sin_sg1[c2];
    if (umain) v_1 = vc[c - 1], v_1s = vc[c - 1 + sj], vs = vc[c + sj], cu = m.cosa_u[c2], ru = m.rsin_u[c2];
    if (vmain) u_s = uc[c - sj], u1_s = uc[c + 1 - sj], u1 = uc[c + 1], cv = m.cosa_v[c2], rv = m.rsin_v[c2];
  }
  __device__ __forceinline__ void finish(real* ut, real* vt) const {
    if (xedge) {
      ut[c] = (u0 > 0.0) ? (u0 / s3) : (u0 / s1);
    } else if (umain) {
      const double vb = 0.25 * (v_1 + v0 + v_1s + vs);
      ut[c] = contra(u0, vb, cu, ru);
    }
    if (vmain) {
      const double ub = 0.25 * (u_s + u1_s + u0 + u1);
      vt[c] = contra(v0, ub, cv, rv);
    }
  }
};
struct FxFramePointF {
  long c;
  bool do_x, do_y;
  double u, v, rdxa_m, rdxa_0, dy, sg3_m, sg1_0, rdya_m, rdya_0, dx, sg4_m, sg2_0, cxa, cya;
  __device__ __forceinline__ void load(const Geo& g, const Met& m, const real* ut, const real* vt, const real* cx_acc, const real* cy_acc,
                                       bool on, int i, int j, int k) {
    const long c2 = IDX2(g, i, j);
    c = c2 + (long)k * g.sk;
    const int sj = g.sj;
    do_x = on && i >= g.is && i <= g.ie + 1, do_y = on && j >= g.js && j <= g.je + 1;
    u = v = rdxa_m = rdxa_0 = dy = sg3_m = sg1_0 = rdya_m = rdya_0 = dx = sg4_m = sg2_0 = cxa = cya = 0.0;
    if (do_x) {
      u = ut[c], rdxa_m = m.rdxa[c2 - 1], rdxa_0 = m.rdxa[c2], dy = m.dy[c2], sg3_m = m.sin_sg3[c2 - 1], sg1_0 = m.sin_sg1[c2];
      if (cx_acc) cxa = cx_acc[c];
    }
    if (do_y) {
      v = vt[c], rdya_m = m.rdya[c2 - sj], rdya_0 = m.rdya[c2], dx = m.dx[c2], sg4_m = m.sin_sg4[c2 - sj], sg2_0 = m.sin_sg2[c2];
      if (cy_acc) cya = cy_acc[c];
    }
  }
  __device__ __forceinline__ void finish(double dt, real* crx, real* cry, real* xfx, real* yfx, real* cx_acc, real* cy_acc) const {
    if (do_x) {
      double cr;
      if (u > 0.0) {
        cr = dt * u * rdxa_m;
        xfx[c] = dy * dt * u * sg3_m;
      } else {
        cr = dt * u * rdxa_0;
        xfx[c] = dy * dt * u * sg1_0;
      }
      crx[c] = cr;
      if (cx_acc) cx_acc[c] = cxa + cr;
    }
    if (do_y) {
      double cr;
      if (v > 0.0) {
        cr = dt * v * rdya_m;
        yfx[c] = dx * dt * v * sg4_m;
      } else {
        cr = dt * v * rdya_0;
        yfx[c] = dx * dt * v * sg2_0;
      }
      cry[c] = cr;
      if (cy_acc) cy_acc[c] = cya + cr;
    }
  }
};
#ifndef FX_NB
#define FX_NB 1  // points of the frame a thread takes through a stage together.  (2 - 4, loads first: no faster -- 128 registers at 4 leave an
                 // interior workgroup no room beside a frame one; 2 under a 64-register cap spills; the frame's chain waits for round trips
                 // that are slow BECAUSE the interior's stream runs beside it: profiles/r06_experiments/x34)
#endif

// ---- ONE launch for all of it (round 6) -------------------------------------------------------------------------------------
// Two kinds of workgroups, in this order: (1) one per level for the FRAME of the plane -- stage A on the frame, the edge / corner stages
// B, C, D, then the fluxes of the frame's points, one after the other behind workgroup barriers (the three launches above and the frame's
// share of the fourth: ~25 us of dependent round trips on a handful of points); (2) the interior box's blocks, which depend on nothing
// the frame kind writes (a point of the box forms its winds from uc / vc).  The few long frame workgroups are first in launch order and
// run beside the interior's stream instead of in front of it.
#ifndef FX_NT
#define FX_NT 1024
#endif
#ifndef FX_MIN_WGS
#define FX_MIN_WGS (512 * 1024 / FX_NT)  // interior blocks the launch should have at least (two workgroups of 1024 threads per compute unit)
#endif
#ifndef FX_ATTR
#define FX_ATTR  // (55 registers as it stands: two workgroups of 1024 threads per compute unit, a frame one and an interior one)
#endif
template <int FXC>
__global__ void __launch_bounds__(FX_NT) FX_ATTR k_fxadv_fused(Geo g, Met m, const real* __restrict__ uc, const real* __restrict__ vc, real* ut, real* vt,
                                                      real* __restrict__ crx, real* __restrict__ cry, real* __restrict__ xfx,
                                                      real* __restrict__ yfx, double dt, real* __restrict__ cx_acc,
                                                      real* __restrict__ cy_acc, FxBox box, int contra_out, Regions A, FxStrips S,
                                                      int nframe, int nbx, FxWindHalo wh) {
  const int tid = (int)threadIdx.x;
  int b = (int)blockIdx.x;
  if (b < nframe) {
    const int k = b;
    int i, j;
    const int na = strip_count(A);
    // (d_sw's separate wind outputs: the halo of this level, copied here -- nothing below depends on it)
    if (wh.u_out != nullptr)
      for (int p = tid, n = wind_halo_points(g); p < n; p += FX_NT) wind_halo_copy_point(g, p, k, wh.u, wh.v, wh.u_out, wh.v_out);
    for (int p0 = tid; p0 < na; p0 += FX_NT * FX_NB) {
      FxFramePointA P[FX_NB];
#pragma unroll
      for (int q = 0; q < FX_NB; ++q) {
        int pi = 0, pj = 0;
        const bool on = strip_point(A, p0 + q * FX_NT, pi, pj) && pi <= g.ni - 2 && pj <= g.nj - 2;
        P[q].load(g, m, uc, vc, on, on ? pi : 1, on ? pj : 1, k);
      }
#pragma unroll
      for (int q = 0; q < FX_NB; ++q) P[q].finish(ut, vt);
    }
    __threadfence_block();
    __syncthreads();
    for (int p = tid, n = strip_count(S.b); p < n; p += FX_NT)
      if (strip_point(S.b, p, i, j)) fx_vt_edges_point(g, m, vc, ut, vt, i, j, k);
    __threadfence_block();
    __syncthreads();
    for (int p = tid, n = strip_count(S.c); p < n; p += FX_NT)
      if (strip_point(S.c, p, i, j)) fx_ut_edges_corners_point(g, m, uc, vc, ut, vt, i, j, k);
    __threadfence_block();
    __syncthreads();
    for (int p = tid, n = strip_count(S.d); p < n; p += FX_NT)
      if (strip_point(S.d, p, i, j)) fx_vt_corners_point(g, m, uc, vc, ut, vt, i, j, k);
    __threadfence_block();
    __syncthreads();
    // the frame's fluxes (fxadv_fluxes_stencil :436-486 at the points outside the interior box)
    for (int p0 = tid; p0 < na; p0 += FX_NT * FX_NB) {
      FxFramePointF P[FX_NB];
#pragma unroll
      for (int q = 0; q < FX_NB; ++q) {
        int pi = 0, pj = 0;
        const bool on = strip_point(A, p0 + q * FX_NT, pi, pj);
        P[q].load(g, m, ut, vt, cx_acc, cy_acc, on, on ? pi : 1, on ? pj : 1, k);
      }
#pragma unroll
      for (int q = 0; q < FX_NB; ++q) P[q].finish(dt, crx, cry, xfx, yfx, cx_acc, cy_acc);
    }
    return;
  }
  b -= nframe;
  // ---- the interior box: FXC levels of a point per thread (the metric values of the point loaded once) ----
#ifdef PACE_EMU
  const int chunk = b / nbx, bxp = b - chunk * nbx;
#else
  // (XCD x takes the x-th eighth of the plane's blocks of every chunk of levels: see k_fxadv_fluxes)
  const int seg = nbx / 8;
  const int chunk = b / nbx, r = b - chunk * nbx;
  const int bxp = (r & 7) * seg + (r >> 3);
#endif
  const long p = (long)bxp * FX_NT + tid;
  const int j = (int)(p / g.sj);
  const int i = (int)(p - (long)j * g.sj);
  if (i < box.i0 || i > box.i1 || j < box.j0 || j > box.j1) return;
  const int k0 = chunk * FXC;
  const long c2 = IDX2(g, i, j);
  const int sj = g.sj;
  const double rdxa_m = m.rdxa[c2 - 1], rdxa_0 = m.rdxa[c2], dy = m.dy[c2], sg3_m = m.sin_sg3[c2 - 1], sg1_0 = m.sin_sg1[c2];
  const double rdya_m = m.rdya[c2 - sj], rdya_0 = m.rdya[c2], dx = m.dx[c2], sg4_m = m.sin_sg4[c2 - sj], sg2_0 = m.sin_sg2[c2];
  const double cosa_u = m.cosa_u[c2], rsin_u = m.rsin_u[c2], cosa_v = m.cosa_v[c2], rsin_v = m.rsin_v[c2];
#pragma unroll
  for (int t = 0; t < FXC; ++t) {
    const int k = k0 + t;
    if (k >= g.nk) break;
    const long c = c2 + (long)k * g.sk;
    const double uc0 = uc[c], vc0 = vc[c];
    const double vb = 0.25 * (vc[c - 1] + vc0 + vc[c - 1 + sj] + vc[c + sj]);
    const double ub = 0.25 * (uc[c - sj] + uc[c + 1 - sj] + uc0 + uc[c + 1]);
    const double u = contra(uc0, vb, cosa_u, rsin_u);
    const double v = contra(vc0, ub, cosa_v, rsin_v);
    if (contra_out) ut[c] = u, vt[c] = v;
    double cr;
    if (u > 0.0) {
      cr = dt * u * rdxa_m;
      xfx[c] = dy * dt * u * sg3_m;
    } else {
      cr = dt * u * rdxa_0;
      xfx[c] = dy * dt * u * sg1_0;
    }
    crx[c] = cr;
    if (cx_acc) cx_acc[c] = cx_acc[c] + cr;
    if (v > 0.0) {
      cr = dt * v * rdya_m;
      yfx[c] = dx * dt * v * sg4_m;
    } else {
      cr = dt * v * rdya_0;
      yfx[c] = dx * dt * v * sg2_0;
    }
    cry[c] = cr;
    if (cy_acc) cy_acc[c] = cy_acc[c] + cr;
  }
}

// the frame of the plane (stage A's regions) and the strips of stages B, C, D
static void fxadv_frame_regions(const Geo& g, bool has_box, Regions& A, FxStrips& S) {
  // stage A: the frame -- whole rows below js+2 and above je-2, the columns left of is+2 and right of ie-2 between them
  if (has_box) {
    add_region(A, 0, g.ni - 2, 0, g.js + 1);
    add_region(A, 0, g.ni - 2, g.je - 1, g.nj - 2);
    add_region(A, 0, g.is + 1, g.js + 2, g.je - 2);
    add_region(A, g.ie - 1, g.ni - 2, g.js + 2, g.je - 2);
  } else {
    add_region(A, 0, g.ni - 2, 0, g.nj - 2);
  }
  // stage B: rows js, je+1 over the whole width; columns is-1, is, ie, ie+1 between them
  add_region(S.b, 0, g.ni - 2, g.js, g.js);
  add_region(S.b, 0, g.ni - 2, g.je + 1, g.je + 1);
  add_region(S.b, g.is - 1, g.is, g.js + 2, g.je - 1);
  add_region(S.b, g.ie, g.ie + 1, g.js + 2, g.je - 1);
  // stage C: rows js-1, js, je, je+1, columns is+1 .. ie
  add_region(S.c, g.is + 1, g.ie, g.js - 1, g.js);
  add_region(S.c, g.is + 1, g.ie, g.je, g.je + 1);
  // stage D: the eight points (is-1, is, ie, ie+1) x (js+1, je)
  add_region(S.d, g.is - 1, g.ie + 1, g.js + 1, g.js + 1);
  add_region(S.d, g.is - 1, g.ie + 1, g.je, g.je);
}

static bool fxadv_split_launches() {
  const char* e = getenv("PACE_FXADV_SPLIT");  // (read per call: an A/B switch)
  return e != nullptr && e[0] == '1';
}

int launch_fxadv(const Geo& g, const Met& m, const real* uc, const real* vc, real* crx, real* cry,
                 real* xfx, real* yfx, real* ut, real* vt, double dt, real* cx_acc, real* cy_acc,
                 hipStream_t st, int part, int contra_out, FxWindHalo* wind_halo) {
  // part 0: everything; 1: the interior box only (needs no halo of uc / vc, and neither ut nor vt); 2: the rest, after part 1
  FxBox box{g.is + 2, g.ie - 2, g.js + 2, g.je - 2, part};
  const bool has_box = box.i1 >= box.i0 && box.j1 >= box.j0;
  if (!has_box) box.i1 = box.i0 - 1, box.j1 = box.j0 - 1;  // nothing is inside
  if (part == 1 && !has_box) return PACE_OK;
  if (!fxadv_split_launches()) {
    Regions A{};
    FxStrips S{};
    if (part != 1) fxadv_frame_regions(g, has_box, A, S);
    const int nframe = (part != 1) ? g.nk : 0;
    // the interior's blocks: the whole plane's flattened rows (a block whose points all lie outside the box returns at once)
    unsigned nbx = (unsigned)(((long)g.sj * g.nj + FX_NT - 1) / FX_NT);
#ifndef PACE_EMU
    nbx = (nbx + 7) / 8 * 8;
#endif
    const bool interior = has_box && part != 2;
    // levels per thread of the interior's blocks: as many as still leave two workgroups (of 1024 threads) per compute unit
    int ch = 8;
    while (ch > 1 && nbx * (unsigned)((g.nk + ch - 1) / ch) < (unsigned)FX_MIN_WGS) ch >>= 1;
    const unsigned nint = interior ? nbx * (unsigned)((g.nk + ch - 1) / ch) : 0u;
    if (nframe + nint == 0) return PACE_OK;
    FxWindHalo wcopy{nullptr, nullptr, nullptr, nullptr, false};
    if (wind_halo != nullptr && nframe > 0) wcopy = *wind_halo, wind_halo->done = true;
#define FX_GO(CH)                                                                                                                            \
  hipLaunchKernelGGL(k_fxadv_fused<CH>, dim3((unsigned)nframe + nint), dim3(FX_NT), 0, st, g, m, uc, vc, ut, vt, crx, cry, xfx, yfx, dt, cx_acc, \
                     cy_acc, box, contra_out, A, S, nframe, (int)nbx, wcopy)
    if (ch == 8) FX_GO(8);
    else if (ch == 4) FX_GO(4);
    else if (ch == 2) FX_GO(2);
    else FX_GO(1);
#undef FX_GO
    PACE_CHECK_LAUNCH();
    return PACE_OK;
  }
  if (part != 1) {
    Regions A{};
    FxStrips S{};
    fxadv_frame_regions(g, has_box, A, S);
    hipLaunchKernelGGL(k_fxadv_frame, regions_grid(A, g.nk), dim3(64, 4), 0, st, g, m, uc, vc, ut, vt, A);
    hipLaunchKernelGGL(k_fxadv_edges, dim3((unsigned)g.nk), dim3(1024), 0, st, g, m, uc, vc, ut, vt, S);
  }
#ifdef PACE_EMU
  const unsigned nbx = plane_grid(g, 1).x;
#else
  const unsigned nbx = (plane_grid(g, 1).x + 7) / 8 * 8;  // (eight equal segments: k_fxadv_fluxes)
#endif
  const bool small = nbx * (unsigned)((g.nk + FX_CH - 1) / FX_CH) < 1024u;  // fewer than four workgroups per compute unit
  if (small) {
    hipLaunchKernelGGL(k_fxadv_fluxes<FX_CH_SMALL>, dim3(nbx, (unsigned)((g.nk + FX_CH_SMALL - 1) / FX_CH_SMALL), 1), dim3(256), 0, st, g, m, uc,
                       vc, ut, vt, crx, cry, xfx, yfx, dt, cx_acc, cy_acc, box, contra_out);
  } else {
    hipLaunchKernelGGL(k_fxadv_fluxes<FX_CH>, dim3(nbx, (unsigned)((g.nk + FX_CH - 1) / FX_CH), 1), dim3(256), 0, st, g, m, uc, vc, ut, vt, crx,
                       cry, xfx, yfx, dt, cx_acc, cy_acc, box, contra_out);
  }
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}
