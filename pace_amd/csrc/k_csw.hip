// C-grid half step (Fortran c_sw) and the D->A->C wind interpolation (d2a2c_vect).
// Reference: fv3core/pace/fv3core/stencils/c_sw.py:31-766 (12 launches + 5 temporaries) and
// d2a2c_vect.py:27-655 (12 launches), stencils/pace/stencils/corners.py:129-305 (multiplier fills).
//
// Four streaming passes instead of 24 launches; every pass is one thread per (i, j, k) with i fastest:
//   A  utmp, vtmp (4-point Lagrange / 2-point average of the D-grid winds), ua, va; the corner
//      fills of all four are produced by the writer itself (the value a fill would copy is recomputed
//      from its source point), so nothing ever rewrites a corner afterwards
//   B  uc, vc (+ contravariant ut, vt with the geoadjust factor folded in) and divgd
//   C  upwind transport of delp / pt / w (corner fills of the *inputs* become read-side index maps,
//      the x/y fluxes are recomputed at both faces instead of stored), kinetic energy, C-grid
//      absolute vorticity
//   D  uc, vc update
// HBM-bound; algorithmic traffic 5 reads (delp, pt, u, v, w) + 11 writes = 128 B/cell.
//
// Round 5: the four passes only run on the BAND of the plane near the tile edges; the tiles of the interior (every point at
// least six cells from an edge: every formula is its plain form there) go through ONE kernel, k_csw_tile, that stages the
// footprints of u, v, delp, pt, w in LDS and keeps every intermediate (utmp, vtmp, the pre-update uc / vc, ke, the vorticity)
// there: 5 reads + 10 writes per cell instead of 41 field passes.
#include "common.h"
#include "kernels.h"

#define A1 (9.0 / 16.0)
#define A2 (-1.0 / 16.0)

__device__ __forceinline__ double contra2(double v1, double v2, double cosa, double rsin2) {
  return (v1 - v2 * cosa) * rsin2;  // d2a2c_vect.py:225-281
}

struct D2A {
  const Geo& g;
  const real* u;  // level base applied
  const real* v;
  // d2a2c_vect.py:283-360 (lagrange_interpolation_{y,x}_p1 inside the tile, avg_box within 3 of an edge)
  __device__ __forceinline__ bool boxed(int i, int j) const {
    return (j < g.js + 3) || (j >= g.je - 2) || (i < g.is + 3) || (i >= g.ie - 2);
  }
  __device__ __forceinline__ double utmp(int i, int j) const {
    const long c = IDX2(g, i, j);
    if (boxed(i, j)) return 0.5 * (u[c] + u[c + g.sj]);
    return A2 * (u[c - g.sj] + u[c + 2 * g.sj]) + A1 * (u[c] + u[c + g.sj]);
  }
  __device__ __forceinline__ double vtmp(int i, int j) const {
    const long c = IDX2(g, i, j);
    if (boxed(i, j)) return 0.5 * (v[c] + v[c + 1]);
    return A2 * (v[c - 1] + v[c + 2]) + A1 * (v[c] + v[c + 1]);
  }
};

// fill_corners_{2,3}cells_mult_x(q, qc, sw=-1, se=1, ne=-1, nw=1): destination (i, j) of the x fill ->
// source point in qc and multiplier; returns false if (i, j) is not a destination.  (corners.py:129-217)
__device__ __forceinline__ bool fill_x_src(const Geo& g, int ncells, int i, int j, int& si, int& sj_, double& mult) {
  if (j == g.js - 1) {
    if (i < g.is && i >= g.is - ncells) { const int a = g.is - i; si = g.is - 1; sj_ = g.js + a - 1; mult = -1.0; return true; }
    if (i > g.ie && i <= g.ie + ncells) { const int a = i - g.ie; si = g.ie + 1; sj_ = g.js + a - 1; mult = 1.0; return true; }
  } else if (j == g.je + 1) {
    if (i < g.is && i >= g.is - ncells) { const int a = g.is - i; si = g.is - 1; sj_ = g.je + 1 - a; mult = 1.0; return true; }
    if (i > g.ie && i <= g.ie + ncells) { const int a = i - g.ie; si = g.ie + 1; sj_ = g.je + 1 - a; mult = -1.0; return true; }
  }
  return false;
}
// ... and of the y fill (corners.py:219-305)
__device__ __forceinline__ bool fill_y_src(const Geo& g, int ncells, int i, int j, int& si, int& sj_, double& mult) {
  if (i == g.is - 1) {
    if (j < g.js && j >= g.js - ncells) { const int a = g.js - j; si = g.is + a - 1; sj_ = g.js - 1; mult = -1.0; return true; }
    if (j > g.je && j <= g.je + ncells) { const int a = j - g.je; si = g.is + a - 1; sj_ = g.je + 1; mult = 1.0; return true; }
  } else if (i == g.ie + 1) {
    if (j < g.js && j >= g.js - ncells) { const int a = g.js - j; si = g.ie + 1 - a; sj_ = g.js - 1; mult = 1.0; return true; }
    if (j > g.je && j <= g.je + ncells) { const int a = j - g.je; si = g.ie + 1 - a; sj_ = g.je + 1; mult = -1.0; return true; }
  }
  return false;
}

// pass A
__device__ __forceinline__ void d2a2c_a_point(const Geo& g, const Met& m, const real* __restrict__ u, const real* __restrict__ v,
                                              real* __restrict__ utmp, real* __restrict__ vtmp, real* __restrict__ ua,
                                              real* __restrict__ va, int i, int j, int k) {
  const long c = IDX3(g, i, j, k);
  const long kb = (long)k * g.sk;
  D2A d{g, u + kb, v + kb};
  int si, sj_;
  double mult;
  const double ut_raw = d.utmp(i, j), vt_raw = d.vtmp(i, j);
  double ut_ = ut_raw, vt_ = vt_raw;
  if (fill_x_src(g, 3, i, j, si, sj_, mult)) ut_ = mult * d.vtmp(si, sj_);
  if (fill_y_src(g, 3, i, j, si, sj_, mult)) vt_ = mult * d.utmp(si, sj_);
  utmp[c] = ut_;
  vtmp[c] = vt_;
  if (i >= g.is - 2 && i <= g.ie + 2 && j >= g.js - 2 && j <= g.je + 2) {
    const long c2 = IDX2(g, i, j);
    double ua_ = contra2(ut_raw, vt_raw, m.cosa_s[c2], m.rsin2[c2]);
    double va_ = contra2(vt_raw, ut_raw, m.cosa_s[c2], m.rsin2[c2]);
    if (fill_x_src(g, 2, i, j, si, sj_, mult)) {
      const long s2 = IDX2(g, si, sj_);
      ua_ = mult * contra2(d.vtmp(si, sj_), d.utmp(si, sj_), m.cosa_s[s2], m.rsin2[s2]);
    }
    if (fill_y_src(g, 2, i, j, si, sj_, mult)) {
      const long s2 = IDX2(g, si, sj_);
      va_ = mult * contra2(d.utmp(si, sj_), d.vtmp(si, sj_), m.cosa_s[s2], m.rsin2[s2]);
    }
    ua[c] = ua_;
    va[c] = va_;
  }
}

__global__ void __launch_bounds__(256)
k_d2a2c_a(Geo g, Met m, const real* __restrict__ u, const real* __restrict__ v, real* __restrict__ utmp,
          real* __restrict__ vtmp, real* __restrict__ ua, real* __restrict__ va, SplitBox box) {
  PLANE_IJK(g);
  if (i > g.ni - 2 || j > g.nj - 2 || box.skip(i, j)) return;
  d2a2c_a_point(g, m, u, v, utmp, vtmp, ua, va, i, j, k);
}
// ... on the rectangles of the band only (launching the plane and returning from nine threads in ten cost 28 us of pass A's 46)
__global__ void __launch_bounds__(256)
k_d2a2c_a_band(Geo g, Met m, const real* __restrict__ u, const real* __restrict__ v, real* __restrict__ utmp,
               real* __restrict__ vtmp, real* __restrict__ ua, real* __restrict__ va, SplitBox box, Regions R) {
  REGION_POINT(R);
  (void)interior;
  if (box.skip(i, j)) return;
  d2a2c_a_point(g, m, u, v, utmp, vtmp, ua, va, i, j, k);
}

// pass B: d2a2c_vect.py:362-527 (ut_main / east_west_edges / north_south_edges / vt_main), geoadjust_ut/vt
// (c_sw.py:159-203) and divergence_corner (c_sw.py:31-156)
// (the points of the frame strips: the edge forms)
__device__ __forceinline__ void d2a2c_b_frame_point(const Geo& g, const Met& m, const real* __restrict__ u, const real* __restrict__ v,
                                                    const real* __restrict__ utmp, const real* __restrict__ vtmp,
                                                    const real* __restrict__ ua, const real* __restrict__ va, real* __restrict__ uc,
                                                    real* __restrict__ vc, real* __restrict__ ut, real* __restrict__ vt,
                                                    real* __restrict__ divgd, double dt2, int do_divg, int geoadjust, int i, int j,
                                                    int k) {
  const long c = IDX3(g, i, j, k);
  const long c2 = IDX2(g, i, j);
  const int sj = g.sj;
  if (j <= g.je + 1) {  // uc, ut on i = is-1 .. ie+2, j = js-1 .. je+1
    double ucv, utv;
    if (i == g.is || i == g.ie + 1) {
      const real* dxa = m.dxa;
      const double t1 = dxa[c2 - 2] + dxa[c2 - 1];
      const double t2 = dxa[c2] + dxa[c2 + 1];
      const double n1 = (t1 + dxa[c2 - 1]) * ua[c - 1] - dxa[c2 - 1] * ua[c - 2];
      const double n2 = (t1 + dxa[c2]) * ua[c] - dxa[c2] * ua[c + 1];
      utv = 0.5 * (n1 / t1 + n2 / t2);
      ucv = (utv > 0.0) ? utv * m.sin_sg3[c2 - 1] : utv * m.sin_sg1[c2];
    } else {
      if (i == g.is - 1 || i == g.ie) {
        ucv = PPM_C1 * utmp[c - 2] + PPM_C2 * utmp[c - 1] + PPM_C3 * utmp[c];
      } else if (i == g.is + 1 || i == g.ie + 2) {
        ucv = PPM_C1 * utmp[c + 1] + PPM_C2 * utmp[c] + PPM_C3 * utmp[c - 1];
      } else {
        ucv = A2 * (utmp[c - 2] + utmp[c + 1]) + A1 * (utmp[c - 1] + utmp[c]);
      }
      utv = contra2(ucv, v[c], m.cosa_u[c2], m.rsin_u[c2]);
    }
    uc[c] = ucv;
    if (geoadjust) utv = (utv > 0.0) ? dt2 * utv * m.dy[c2] * m.sin_sg3[c2 - 1] : dt2 * utv * m.dy[c2] * m.sin_sg1[c2];
    ut[c] = utv;
  }
  if (i <= g.ie + 1) {  // vc, vt on i = is-1 .. ie+1, j = js-1 .. je+2
    double vcv, vtv;
    if (j == g.js || j == g.je + 1) {
      const real* dya = m.dya;
      const double t1 = dya[c2 - 2 * sj] + dya[c2 - sj];
      const double t2 = dya[c2] + dya[c2 + sj];
      const double n1 = (t1 + dya[c2 - sj]) * va[c - sj] - dya[c2 - sj] * va[c - 2 * sj];
      const double n2 = (t1 + dya[c2]) * va[c] - dya[c2] * va[c + sj];
      vtv = 0.5 * (n1 / t1 + n2 / t2);
      vcv = (vtv > 0.0) ? vtv * m.sin_sg4[c2 - sj] : vtv * m.sin_sg2[c2];
    } else {
      if (j == g.js - 1 || j == g.je) {
        vcv = PPM_C1 * vtmp[c - 2 * sj] + PPM_C2 * vtmp[c - sj] + PPM_C3 * vtmp[c];
      } else if (j == g.js + 1 || j == g.je + 2) {
        vcv = PPM_C1 * vtmp[c + sj] + PPM_C2 * vtmp[c] + PPM_C3 * vtmp[c - sj];
      } else {
        vcv = A2 * (vtmp[c - 2 * sj] + vtmp[c + sj]) + A1 * (vtmp[c - sj] + vtmp[c]);
      }
      vtv = contra2(vcv, u[c], m.cosa_v[c2], m.rsin_v[c2]);
    }
    vc[c] = vcv;
    if (geoadjust) vtv = (vtv > 0.0) ? dt2 * vtv * m.dx[c2] * m.sin_sg4[c2 - sj] : dt2 * vtv * m.dx[c2] * m.sin_sg2[c2];
    vt[c] = vtv;
  }
  if (do_divg && i >= g.is && i <= g.ie + 1 && j >= g.js && j <= g.je + 1) {
    const bool iedge = (i == g.is || i == g.ie + 1), jedge = (j == g.js || j == g.je + 1);
    const real* sg1 = m.sin_sg1; const real* sg2 = m.sin_sg2; const real* sg3 = m.sin_sg3; const real* sg4 = m.sin_sg4;
    const real* cg1 = m.cos_sg1; const real* cg2 = m.cos_sg2; const real* cg3 = m.cos_sg3; const real* cg4 = m.cos_sg4;
    // uf at (i, j) and (i-1, j); vf at (i, j) and (i, j-1), interior or edge form (c_sw.py:60-156)
    const double uf = jedge ? u[c] * m.dyc[c2] * 0.5 * (sg4[c2 - sj] + sg2[c2])
                            : (u[c] - 0.25 * (va[c - sj] + va[c]) * (cg4[c2 - sj] + cg2[c2])) * m.dyc[c2] * 0.5 * (sg4[c2 - sj] + sg2[c2]);
    const double uf1 = jedge ? u[c - 1] * m.dyc[c2 - 1] * 0.5 * (sg4[c2 - 1 - sj] + sg2[c2 - 1])
                             : (u[c - 1] - 0.25 * (va[c - 1 - sj] + va[c - 1]) * (cg4[c2 - 1 - sj] + cg2[c2 - 1])) * m.dyc[c2 - 1] * 0.5 * (sg4[c2 - 1 - sj] + sg2[c2 - 1]);
    const double vf = iedge ? v[c] * m.dxc[c2] * 0.5 * (sg3[c2 - 1] + sg1[c2])
                            : (v[c] - 0.25 * (ua[c - 1] + ua[c]) * (cg3[c2 - 1] + cg1[c2])) * m.dxc[c2] * 0.5 * (sg3[c2 - 1] + sg1[c2]);
    const double vf1 = iedge ? v[c - sj] * m.dxc[c2 - sj] * 0.5 * (sg3[c2 - 1 - sj] + sg1[c2 - sj])
                             : (v[c - sj] - 0.25 * (ua[c - 1 - sj] + ua[c - sj]) * (cg3[c2 - 1 - sj] + cg1[c2 - sj])) * m.dxc[c2 - sj] * 0.5 * (sg3[c2 - 1 - sj] + sg1[c2 - sj]);
    double d;
    if (iedge && j == g.js) d = (-vf + uf1 - uf) * m.rarea_c[c2];
    else if (iedge && j == g.je + 1) d = (vf1 + uf1 - uf) * m.rarea_c[c2];
    else d = (vf1 - vf + uf1 - uf) * m.rarea_c[c2];
    divgd[c] = d;
  }
}

// A thread takes D2B_CH consecutive levels of its point.  An interior point reads ~30 metric values (cosa / rsin of both winds,
// dx, dy, the sin_sg and cos_sg of four neighbours, dxc, dyc, rarea_c) against 14 values of the 3-D fields: they -- and the sums
// of them the divergence uses -- are formed once per point instead of once per level.
#define D2B_CH 8
// (CH = 2 when only the band of the plane is left to this pass: few points, so levels side by side instead of one after the other)
template <int CH>
__global__ void __launch_bounds__(256)
k_d2a2c_b(Geo g, Met m, const real* __restrict__ u, const real* __restrict__ v,
          const real* __restrict__ utmp, const real* __restrict__ vtmp, const real* __restrict__ ua,
          const real* __restrict__ va, real* __restrict__ uc, real* __restrict__ vc, real* __restrict__ ut,
          real* __restrict__ vt, real* __restrict__ divgd, double dt2, int do_divg, int geoadjust, Regions R) {
  REGION_POINT_CHUNKED(R, CH, g.nk);  // (CH levels per thread in the interior, one in the frame strips)
  const int sj = g.sj;
  if (!interior) {
    d2a2c_b_frame_point(g, m, u, v, utmp, vtmp, ua, va, uc, vc, ut, vt, divgd, dt2, do_divg, geoadjust, i, j, k0);
    return;
  }
  // is+2 <= i <= ie-1, js+2 <= j <= je-1: 4-point Lagrange everywhere, no edge wind, interior divergence
  const long c2 = IDX2(g, i, j);
  const double cosa_u = m.cosa_u[c2], rsin_u = m.rsin_u[c2], cosa_v = m.cosa_v[c2], rsin_v = m.rsin_v[c2];
  const double dy = m.dy[c2], dx = m.dx[c2];
  const double sg3_m = m.sin_sg3[c2 - 1], sg1_0 = m.sin_sg1[c2], sg4_m = m.sin_sg4[c2 - sj], sg2_0 = m.sin_sg2[c2];
  // divergence_corner: the sums of cos_sg / sin_sg on the four faces around the corner, dyc / dxc there, rarea_c
  double cu = 0.0, su = 0.0, dyc0 = 0.0, cu1 = 0.0, su1 = 0.0, dyc1 = 0.0, cv = 0.0, sv = 0.0, dxc0 = 0.0, cv1 = 0.0, sv1 = 0.0,
         dxc1 = 0.0, rarea_c = 0.0;
  if (do_divg) {
    cu = m.cos_sg4[c2 - sj] + m.cos_sg2[c2], su = sg4_m + sg2_0, dyc0 = m.dyc[c2];
    cu1 = m.cos_sg4[c2 - 1 - sj] + m.cos_sg2[c2 - 1], su1 = m.sin_sg4[c2 - 1 - sj] + m.sin_sg2[c2 - 1], dyc1 = m.dyc[c2 - 1];
    cv = m.cos_sg3[c2 - 1] + m.cos_sg1[c2], sv = sg3_m + sg1_0, dxc0 = m.dxc[c2];
    cv1 = m.cos_sg3[c2 - 1 - sj] + m.cos_sg1[c2 - sj], sv1 = m.sin_sg3[c2 - 1 - sj] + m.sin_sg1[c2 - sj], dxc1 = m.dxc[c2 - sj];
    rarea_c = m.rarea_c[c2];
  }
#pragma unroll
  for (int t = 0; t < CH; ++t) {
    if (t >= nk_here) break;
    const long c = c2 + (long)(k0 + t) * g.sk;
    const double ucv = A2 * (utmp[c - 2] + utmp[c + 1]) + A1 * (utmp[c - 1] + utmp[c]);
    double utv = contra2(ucv, v[c], cosa_u, rsin_u);
    uc[c] = ucv;
    if (geoadjust) utv = (utv > 0.0) ? dt2 * utv * dy * sg3_m : dt2 * utv * dy * sg1_0;
    ut[c] = utv;
    const double vcv = A2 * (vtmp[c - 2 * sj] + vtmp[c + sj]) + A1 * (vtmp[c - sj] + vtmp[c]);
    double vtv = contra2(vcv, u[c], cosa_v, rsin_v);
    vc[c] = vcv;
    if (geoadjust) vtv = (vtv > 0.0) ? dt2 * vtv * dx * sg4_m : dt2 * vtv * dx * sg2_0;
    vt[c] = vtv;
    if (do_divg) {
      const double uf = (u[c] - 0.25 * (va[c - sj] + va[c]) * cu) * dyc0 * 0.5 * su;
      const double uf1 = (u[c - 1] - 0.25 * (va[c - 1 - sj] + va[c - 1]) * cu1) * dyc1 * 0.5 * su1;
      const double vf = (v[c] - 0.25 * (ua[c - 1] + ua[c]) * cv) * dxc0 * 0.5 * sv;
      const double vf1 = (v[c - sj] - 0.25 * (ua[c - 1 - sj] + ua[c - sj]) * cv1) * dxc1 * 0.5 * sv1;
      divgd[c] = (vf1 - vf + uf1 - uf) * rarea_c;
    }
  }
}

static void launch_d2a2c_b(const Geo& g, const Met& m, const real* u, const real* v, const real* utmp, const real* vtmp,
                           const real* ua, const real* va, real* uc, real* vc, real* ut, real* vt, real* divgd, double dt2,
                           int do_divg, int geoadjust, const Regions& rb, hipStream_t st) {
  if (rb.nplain > 1)  // (a band: the plain region has a hole)
    hipLaunchKernelGGL(k_d2a2c_b<2>, regions_grid_chunked(rb, g.nk, 2), dim3(64, 4), 0, st, g, m, u, v, utmp, vtmp, ua, va, uc, vc, ut,
                       vt, divgd, dt2, do_divg, geoadjust, rb);
  else
    hipLaunchKernelGGL(k_d2a2c_b<D2B_CH>, regions_grid_chunked(rb, g.nk, D2B_CH), dim3(64, 4), 0, st, g, m, u, v, utmp, vtmp, ua, va, uc, vc, ut,
                       vt, divgd, dt2, do_divg, geoadjust, rb);
}

// read-side forms of fill_corners_2cells_x / _y with unit multipliers on the transported scalars
// (c_sw.py:206-228,699,724; corners.py:129-305)
__device__ __forceinline__ long cell_xfill(const Geo& g, int i, int j) {
  int si, sj_;
  double mult;
  if (fill_x_src(g, 2, i, j, si, sj_, mult)) return IDX2(g, si, sj_);
  return IDX2(g, i, j);
}
__device__ __forceinline__ long cell_yfill(const Geo& g, int i, int j) {
  int si, sj_;
  double mult;
  if (fill_y_src(g, 2, i, j, si, sj_, mult)) return IDX2(g, si, sj_);
  return IDX2(g, i, j);
}

// What the reference LEAVES next to the corners of delp, pt, w: its C-grid transport fills two cells per corner in x, forms the x
// fluxes, fills two in y, forms the y fluxes -- in place (c_sw.py:483-600; corners.py:129-305 fill_corners_2cells_x / _y with
// multipliers 1) -- and TranslateC_SW compares the three fields over the whole storage (translate_c_sw.py:82-113).  The kernels here
// apply the fills as index maps on reads; this writes what stays: per corner the x fill's second cell and the y fill's two (the
// y fill overwrites the x fill's first).  Every source lies in an edge halo, which c_sw never writes.  36 cells per level.
__global__ void __launch_bounds__(64)
k_csw_corner_cells(Geo g, real* __restrict__ delp, real* __restrict__ pt, real* __restrict__ w) {
  const int t = (int)threadIdx.x;
  if (t >= 36) return;
  const int f = t / 12, e = t - f * 12, q = e / 3, c = e - q * 3;  // field, corner (bit 0: east, bit 1: north), cell
  real* const p = (f == 0 ? delp : f == 1 ? pt : w) + (long)blockIdx.x * g.sk;
  const bool east = q & 1, north = q & 2;
  const int ic = east ? g.ie + 1 : g.is - 1, jc = north ? g.je + 1 : g.js - 1;  // the corner cell next to the tile
  const int di = east ? 1 : -1, dj = north ? 1 : -1;                             // away from the tile
  int i, j, si, sj_;
  if (c == 0) i = ic + di, j = jc, si = ic, sj_ = jc - 2 * dj;        // x fill, a = 2: (is-2, js-1) <- (is-1, js+1)
  else if (c == 1) i = ic, j = jc, si = ic - di, sj_ = jc;            // y fill, a = 1: (is-1, js-1) <- (is, js-1)
  else i = ic, j = jc + dj, si = ic - 2 * di, sj_ = jc;               // y fill, a = 2: (is-1, js-2) <- (is+1, js-1)
  p[IDX2(g, i, j)] = p[IDX2(g, si, sj_)];
}

// pass C: compute_nonhydrostatic_fluxes_x (c_sw.py:231-259), transportdelp_update_vorticity_and_kineticenergy
// (:262-364), circulation_cgrid (:367-397), absolute_vorticity (:400-408)
__global__ void __launch_bounds__(256)
k_csw_transport(Geo g, Met m, const real* __restrict__ delp, const real* __restrict__ pt,
                const real* __restrict__ w, const real* __restrict__ u, const real* __restrict__ v,
                const real* __restrict__ ua, const real* __restrict__ va, const real* __restrict__ uc,
                const real* __restrict__ vc, const real* __restrict__ ut, const real* __restrict__ vt,
                real* __restrict__ delpc, real* __restrict__ ptc, real* __restrict__ omga,
                real* __restrict__ ke, real* __restrict__ vort, double dt2, Regions R) {
  REGION_POINT(R);
  const long kb = (long)k * g.sk;
  const long c2 = IDX2(g, i, j);
  const long c = c2 + kb;
  const int sj = g.sj;
  if (interior) {
    // is+1 <= i <= ie-1, js+1 <= j <= je-1: no corner-filled operand, no edge form of ke / vorticity
    double fx1[2], fx[2], fx2[2], fy1[2], fy[2], fy2[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const double utv = ut[c + t];
      const long s = (utv > 0.0) ? c + t - 1 : c + t;
      fx1[t] = utv * delp[s];
      fx[t] = fx1[t] * pt[s];
      fx2[t] = fx1[t] * w[s];
      const double vtv = vt[c + (long)t * sj];
      const long s2 = (vtv > 0.0) ? c + (long)(t - 1) * sj : c + (long)t * sj;
      fy1[t] = vtv * delp[s2];
      fy[t] = fy1[t] * pt[s2];
      fy2[t] = fy1[t] * w[s2];
    }
    const double ra = m.rarea[c2];
    const double dp = delp[c];
    const double dpc = dp + (fx1[0] - fx1[1] + fy1[0] - fy1[1]) * ra;
    delpc[c] = dpc;
    ptc[c] = (pt[c] * dp + (fx[0] - fx[1] + fy[0] - fy[1]) * ra) / dpc;
    omga[c] = (w[c] * dp + (fx2[0] - fx2[1] + fy2[0] - fy2[1]) * ra) / dpc;
    const double uav = ua[c], vav = va[c];
    const double kev = (uav > 0.0) ? uc[c] : uc[c + 1];
    const double vov = (vav > 0.0) ? vc[c] : vc[c + sj];
    ke[c] = 0.5 * dt2 * (uav * kev + vav * vov);
    const double fxc = m.dxc[c2] * uc[c];
    const double fyc = m.dyc[c2] * vc[c];
    const double fx1c = m.dxc[c2 - sj] * uc[c - sj];
    const double fy1c = m.dyc[c2 - 1] * vc[c - 1];
    vort[c] = m.fC[c2] + m.rarea_c[c2] * (fx1c - fxc - fy1c + fyc);
    return;
  }
  {
    // x faces i and i+1 (corner-filled in x), y faces j and j+1 (corner-filled in y)
    double fx1[2], fx[2], fx2[2], fy1[2], fy[2], fy2[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const double utv = ut[c + t];
      const long s = kb + ((utv > 0.0) ? cell_xfill(g, i + t - 1, j) : cell_xfill(g, i + t, j));
      fx1[t] = utv * delp[s];
      fx[t] = fx1[t] * pt[s];
      fx2[t] = fx1[t] * w[s];
      const double vtv = vt[c + (long)t * sj];
      const long s2 = kb + ((vtv > 0.0) ? cell_yfill(g, i, j + t - 1) : cell_yfill(g, i, j + t));
      fy1[t] = vtv * delp[s2];
      fy[t] = fy1[t] * pt[s2];
      fy2[t] = fy1[t] * w[s2];
    }
    // the cell's own delp/pt/w are read after both fills: the y fill is the last writer of a corner cell
    const long s0 = kb + cell_yfill(g, i, j);
    const double ra = m.rarea[c2];
    const double dp = delp[s0];
    const double dpc = dp + (fx1[0] - fx1[1] + fy1[0] - fy1[1]) * ra;
    delpc[c] = dpc;
    ptc[c] = (pt[s0] * dp + (fx[0] - fx[1] + fy[0] - fy[1]) * ra) / dpc;
    omga[c] = (w[s0] * dp + (fx2[0] - fx2[1] + fy2[0] - fy2[1]) * ra) / dpc;
  }
  {
    const double uav = ua[c], vav = va[c];
    double kev = (uav > 0.0) ? uc[c] : uc[c + 1];
    double vov = (vav > 0.0) ? vc[c] : vc[c + sj];
    if ((j == g.js - 1 || j == g.je) && !(vav > 0.0)) vov = vov * m.sin_sg4[c2] + u[c + sj] * m.cos_sg4[c2];
    if ((j == g.js || j == g.je + 1) && (vav > 0.0)) vov = vov * m.sin_sg2[c2] + u[c] * m.cos_sg2[c2];
    if ((i == g.ie || i == g.is - 1) && !(uav > 0.0)) kev = kev * m.sin_sg3[c2] + v[c + 1] * m.cos_sg3[c2];
    if ((i == g.ie + 1 || i == g.is) && (uav > 0.0)) kev = kev * m.sin_sg1[c2] + v[c] * m.cos_sg1[c2];
    ke[c] = 0.5 * dt2 * (uav * kev + vav * vov);
  }
  if (i >= g.is && j >= g.js) {
    const double fxc = m.dxc[c2] * uc[c];
    const double fyc = m.dyc[c2] * vc[c];
    const double fx1c = m.dxc[c2 - sj] * uc[c - sj];
    const double fy1c = m.dyc[c2 - 1] * vc[c - 1];
    double vcirc = fx1c - fxc - fy1c + fyc;
    const bool jc = (j == g.js || j == g.je + 1);
    if (i == g.is && jc) vcirc = fx1c - fxc + fyc;
    if (i == g.ie + 1 && jc) vcirc = fx1c - fxc - fy1c;
    vort[c] = m.fC[c2] + m.rarea_c[c2] * vcirc;
  }
}

// pass D: update_y_velocity (c_sw.py:445-480), update_x_velocity (:411-442)
#define UV_CH 8
template <int CH>
__device__ __forceinline__ void csw_update_point(const Geo& g, const Met& m, const real* __restrict__ u, const real* __restrict__ v,
                                                 const real* __restrict__ ke, const real* __restrict__ vort,
                                                 const real* __restrict__ ucw, const real* __restrict__ vcw, real* __restrict__ uc,
                                                 real* __restrict__ vc, double dt2, int i, int j, int k0) {
  // ucw / vcw: the winds pass B left in the workspace; uc / vc: the caller's, written on the band only (the tiles write theirs)
  // (UV_CH levels per thread: the six metric values of a point are loaded once)
  if (i < g.is - 1 || i > g.ie + 2 || j < g.js - 1 || j > g.je + 2) return;
  const long c2 = IDX2(g, i, j);
  const int sj = g.sj;
  const bool inside = i >= g.is && i <= g.ie + 1 && j >= g.js && j <= g.je + 1;
  const bool do_v = inside && i <= g.ie, do_u = inside && j <= g.je;
  // (around the updated windows the caller's uc / vc end as pass B left them, on pass B's domains: TranslateC_SW compares them)
  const bool copy_v = !do_v && i <= g.ie + 1, copy_u = !do_u && j <= g.je + 1;
  const bool v_edge = j == g.js || j == g.je + 1, u_edge = i == g.is || i == g.ie + 1;
  double cosa_v = 0.0, sina_v = 1.0, rdyc = 0.0, cosa_u = 0.0, sina_u = 1.0, rdxc = 0.0;
  if (do_v) cosa_v = m.cosa_v[c2], sina_v = m.sina_v[c2], rdyc = m.rdyc[c2];
  if (do_u) cosa_u = m.cosa_u[c2], sina_u = m.sina_u[c2], rdxc = m.rdxc[c2];
#pragma unroll
  for (int t = 0; t < CH; ++t) {
    if (k0 + t >= g.nk) break;
    const long c = c2 + (long)(k0 + t) * g.sk;
    if (do_v) {
      const double vc0 = vcw[c];
      const double tmp = v_edge ? dt2 * u[c] : dt2 * (u[c] - vc0 * cosa_v) / sina_v;
      const double flux = (tmp > 0.0) ? vort[c] : vort[c + 1];
      vc[c] = vc0 - tmp * flux + rdyc * (ke[c - sj] - ke[c]);
    }
    if (do_u) {
      const double uc0 = ucw[c];
      const double tmp = u_edge ? dt2 * v[c] : dt2 * (v[c] - uc0 * cosa_u) / sina_u;
      const double flux = (tmp > 0.0) ? vort[c] : vort[c + sj];
      uc[c] = uc0 + tmp * flux + rdxc * (ke[c - 1] - ke[c]);
    }
    if (copy_v) vc[c] = vcw[c];
    if (copy_u) uc[c] = ucw[c];
  }
}
__global__ void __launch_bounds__(256)
k_csw_update_uc_vc(Geo g, Met m, const real* __restrict__ u, const real* __restrict__ v,
                   const real* __restrict__ ke, const real* __restrict__ vort, const real* __restrict__ ucw,
                   const real* __restrict__ vcw, real* __restrict__ uc, real* __restrict__ vc, double dt2) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  const int j = (int)(p / g.sj);
  const int i = (int)(p - (long)j * g.sj);
  if (j >= g.nj || i >= g.ni) return;
  csw_update_point<UV_CH>(g, m, u, v, ke, vort, ucw, vcw, uc, vc, dt2, i, j, (int)blockIdx.y * UV_CH);
}
// ... on the rectangles of the band (blockIdx.z: the chunk of UV_CH_BAND levels: few points, so levels side by side)
#define UV_CH_BAND 2
__global__ void __launch_bounds__(256)
k_csw_update_uc_vc_band(Geo g, Met m, const real* __restrict__ u, const real* __restrict__ v,
                        const real* __restrict__ ke, const real* __restrict__ vort, const real* __restrict__ ucw,
                        const real* __restrict__ vcw, real* __restrict__ uc, real* __restrict__ vc, double dt2, Regions R) {
  REGION_POINT(R);
  (void)interior;
  csw_update_point<UV_CH_BAND>(g, m, u, v, ke, vort, ucw, vcw, uc, vc, dt2, i, j, k * UV_CH_BAND);
}

// ------------------------------------------------------------------------------------------------------------------------------
// The interior tiles: all of c_sw for CSW_TI x CSW_TJ cells of one level in one workgroup.  Valid where every formula is its
// plain form: own cells within [is+6, ie-5] x [js+6, je-5] (utmp / vtmp by 4-point Lagrange three cells beyond the winds that
// use them, no edge wind, no corner fill, interior divergence, no edge term of ke / vorticity).  Expressions as in passes A-D
// above, term for term: the results are the same bits.
// LDS planes on the frame [i0-3, i1+3] x [j0-3, j1+3]: u, v (later delp, pt), utmp (later ke), vtmp (later the vorticity), ua
// (later w), va, uc, vc (before the update), ut, vt.
// ------------------------------------------------------------------------------------------------------------------------------
#ifndef CSW_TI
#define CSW_TI 30  // 181 = 192 - 11 plain cells per row at C192: six tiles of 30
#define CSW_TJ 14  //                                           twelve of 14 (the ring of cells a phase works on is then 32 x 16 = 512: one per thread)
#endif
#define CSW_NT 512
#ifndef CSW_STAMP
#define CSW_STAMP(n)  // (tools/census/csw_prof.hip: wall-clock stamps of every workgroup of k_csw_tile)
#endif
#define CSW_PW (CSW_TI + 7)
#define CSW_PH (CSW_TJ + 7)
#define CSW_PLANE (CSW_PW * CSW_PH)
#ifdef PACE_EMU
#define CSW_ATTR
#else
#define CSW_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))  // two workgroups per CU (74 KB of LDS each): 128 registers
#endif
static_assert(10 * CSW_PLANE * 8 <= 80 * 1024, "two workgroups of k_csw_tile per CU");

__global__ void __launch_bounds__(CSW_NT) CSW_ATTR
k_csw_tile(Geo g, Met m, const real* __restrict__ delp, const real* __restrict__ pt, const real* __restrict__ w,
           const real* __restrict__ u, const real* __restrict__ v, real* __restrict__ delpc, real* __restrict__ ptc,
           real* __restrict__ omga, real* __restrict__ uc, real* __restrict__ vc, real* __restrict__ ua, real* __restrict__ va,
           real* __restrict__ ut, real* __restrict__ vt, real* __restrict__ divgd, double dt2, int do_divg, int ntx, int nty, int it0, int jt0) {
  __shared__ double L[10][CSW_PLANE];
  enum { PU = 0, PV, PUT, PVT, PUA, PVA, PUC, PVC, PUTC, PVTC };  // PUT / PVT: utmp / vtmp; PUTC / PVTC: ut / vt (contravariant, adjusted)
  constexpr int TI = CSW_TI, TJ = CSW_TJ, PW = CSW_PW, NT = CSW_NT;
  const int tid = (int)threadIdx.x;
  // Workgroups are dealt to the eight XCDs round-robin in launch order and each XCD has an L2 of its own: workgroup 8 q + x takes
  // tile x + 8 (q mod T8) at level q / T8, so that an XCD works on an eighth of the tiles at all levels -- the ~30 two-dimensional
  // metric arrays of its tiles (10 MB for the plane) stay in its 4 MB L2 instead of coming from the memory side every level.
  const int ntiles = ntx * nty, t8 = (ntiles + 7) / 8;
  const int q = (int)blockIdx.x >> 3, tile = ((int)blockIdx.x & 7) + 8 * (q % t8), lev = q / t8;
  if (tile >= ntiles || lev >= g.nk) return;
  const int by = tile / ntx, bx = tile - by * ntx;
  const int i0 = it0 + bx * TI, j0 = jt0 + by * TJ, i1 = i0 + TI - 1, j1 = j0 + TJ - 1;
  const int ib = i0 - 3, jb = j0 - 3, sj = g.sj;
  const long kb = (long)lev * g.sk;
#define CSW_LP(i, j) (((j)-jb) * PW + ((i)-ib))
  CSW_STAMP(0);
  // ---- the footprints: u on [i0-3, i1+2] x [j0-2, j1+3], v on [i0-2, i1+3] x [j0-3, j1+2]; delp, pt, w on the own cells + 1, kept
  // in registers until u and v are dead
  for (int e = tid; e < (TI + 6) * (TJ + 6); e += NT) {
    const int jj = e / (TI + 6), ii = e - jj * (TI + 6);
    L[PU][CSW_LP(i0 - 3 + ii, j0 - 2 + jj)] = u[kb + IDX2(g, i0 - 3 + ii, j0 - 2 + jj)];
    L[PV][CSW_LP(i0 - 2 + ii, j0 - 3 + jj)] = v[kb + IDX2(g, i0 - 2 + ii, j0 - 3 + jj)];
  }
  constexpr int NC1 = (TI + 2) * (TJ + 2), NS = (NC1 + NT - 1) / NT;
  double dpr[NS], ptr_[NS], wr[NS];
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    const int e = tid + NT * t;
    const int jj = e / (TI + 2), ii = e - jj * (TI + 2);
    const long c = kb + IDX2(g, i0 - 1 + ii, j0 - 1 + (e < NC1 ? jj : 0));
    dpr[t] = delp[c];
    ptr_[t] = pt[c];
    wr[t] = w[c];
  }
  __syncthreads();
  CSW_STAMP(1);
  // ---- pass A: utmp on [i0-3, i1+2] x [j0-1, j1+1], vtmp on [i0-1, i1+1] x [j0-3, j1+2]
  for (int e = tid; e < (TI + 6) * (TJ + 3); e += NT) {
    const int jj = e / (TI + 6), ii = e - jj * (TI + 6);
    const int p = CSW_LP(i0 - 3 + ii, j0 - 1 + jj);
    L[PUT][p] = A2 * (L[PU][p - PW] + L[PU][p + 2 * PW]) + A1 * (L[PU][p] + L[PU][p + PW]);
  }
  for (int e = tid; e < (TI + 3) * (TJ + 6); e += NT) {
    const int jj = e / (TI + 3), ii = e - jj * (TI + 3);
    const int p = CSW_LP(i0 - 1 + ii, j0 - 3 + jj);
    L[PVT][p] = A2 * (L[PV][p - 1] + L[PV][p + 2]) + A1 * (L[PV][p] + L[PV][p + 1]);
  }
  __syncthreads();
  CSW_STAMP(2);
  // ---- pass A's ua, va on [i0-1, i1] x [j0-1, j1]; pass B's uc, vc, ut, vt on [i0-1, i1+1] x [j0-1, j1+1]
  for (int e = tid; e < NC1; e += NT) {
    const int jj = e / (TI + 2), ii = e - jj * (TI + 2);
    const int i = i0 - 1 + ii, j = j0 - 1 + jj;
    const int p = CSW_LP(i, j);
    const long c2 = IDX2(g, i, j);
    const long c = kb + c2;
    const bool own = i >= i0 && i <= i1 && j >= j0 && j <= j1;
    if (i <= i1 && j <= j1) {
      const double ut_raw = L[PUT][p], vt_raw = L[PVT][p];
      const double cs = m.cosa_s[c2], rs2 = m.rsin2[c2];
      const double ua_ = contra2(ut_raw, vt_raw, cs, rs2);
      const double va_ = contra2(vt_raw, ut_raw, cs, rs2);
      L[PUA][p] = ua_;
      L[PVA][p] = va_;
      if (own) ua[c] = ua_, va[c] = va_;
    }
    const double ucv = A2 * (L[PUT][p - 2] + L[PUT][p + 1]) + A1 * (L[PUT][p - 1] + L[PUT][p]);
    double utv = contra2(ucv, L[PV][p], m.cosa_u[c2], m.rsin_u[c2]);
    utv = (utv > 0.0) ? dt2 * utv * m.dy[c2] * m.sin_sg3[c2 - 1] : dt2 * utv * m.dy[c2] * m.sin_sg1[c2];
    L[PUC][p] = ucv;
    L[PUTC][p] = utv;
    const double vcv = A2 * (L[PVT][p - 2 * PW] + L[PVT][p + PW]) + A1 * (L[PVT][p - PW] + L[PVT][p]);
    double vtv = contra2(vcv, L[PU][p], m.cosa_v[c2], m.rsin_v[c2]);
    vtv = (vtv > 0.0) ? dt2 * vtv * m.dx[c2] * m.sin_sg4[c2 - sj] : dt2 * vtv * m.dx[c2] * m.sin_sg2[c2];
    L[PVC][p] = vcv;
    L[PVTC][p] = vtv;
    if (own) ut[c] = utv, vt[c] = vtv;
  }
  __syncthreads();
  CSW_STAMP(3);
  // ---- the corner divergence at the own points, ke on [i0-1, i1] x [j0-1, j1] (over utmp), the vorticity on [i0, i1+1] x [j0, j1+1]
  // (over vtmp); the own u, v to registers
  constexpr int NOWN = TI * TJ, NO = (NOWN + NT - 1) / NT;
  double ur[NO], vr[NO];
#pragma unroll
  for (int t = 0; t < NO; ++t) {
    const int e = tid + NT * t;
    const int jj = e / TI, ii = e - jj * TI;
    const int i = i0 + ii, j = j0 + (e < NOWN ? jj : 0);
    const int p = CSW_LP(i, j);
    ur[t] = L[PU][p];
    vr[t] = L[PV][p];
    if (do_divg && e < NOWN) {
      const long c2 = IDX2(g, i, j);
      const double cu = m.cos_sg4[c2 - sj] + m.cos_sg2[c2], su = m.sin_sg4[c2 - sj] + m.sin_sg2[c2], dyc0 = m.dyc[c2];
      const double cu1 = m.cos_sg4[c2 - 1 - sj] + m.cos_sg2[c2 - 1], su1 = m.sin_sg4[c2 - 1 - sj] + m.sin_sg2[c2 - 1], dyc1 = m.dyc[c2 - 1];
      const double cv = m.cos_sg3[c2 - 1] + m.cos_sg1[c2], sv = m.sin_sg3[c2 - 1] + m.sin_sg1[c2], dxc0 = m.dxc[c2];
      const double cv1 = m.cos_sg3[c2 - 1 - sj] + m.cos_sg1[c2 - sj], sv1 = m.sin_sg3[c2 - 1 - sj] + m.sin_sg1[c2 - sj], dxc1 = m.dxc[c2 - sj];
      const double uf = (L[PU][p] - 0.25 * (L[PVA][p - PW] + L[PVA][p]) * cu) * dyc0 * 0.5 * su;
      const double uf1 = (L[PU][p - 1] - 0.25 * (L[PVA][p - 1 - PW] + L[PVA][p - 1]) * cu1) * dyc1 * 0.5 * su1;
      const double vf = (L[PV][p] - 0.25 * (L[PUA][p - 1] + L[PUA][p]) * cv) * dxc0 * 0.5 * sv;
      const double vf1 = (L[PV][p - PW] - 0.25 * (L[PUA][p - 1 - PW] + L[PUA][p - PW]) * cv1) * dxc1 * 0.5 * sv1;
      divgd[kb + c2] = (vf1 - vf + uf1 - uf) * m.rarea_c[c2];
    }
  }
  for (int e = tid; e < (TI + 1) * (TJ + 1); e += NT) {
    const int jj = e / (TI + 1), ii = e - jj * (TI + 1);
    {
      const int p = CSW_LP(i0 - 1 + ii, j0 - 1 + jj);
      const double uav = L[PUA][p], vav = L[PVA][p];
      const double kev = (uav > 0.0) ? L[PUC][p] : L[PUC][p + 1];
      const double vov = (vav > 0.0) ? L[PVC][p] : L[PVC][p + PW];
      L[PUT][p] = 0.5 * dt2 * (uav * kev + vav * vov);
    }
    {
      const int i = i0 + ii, j = j0 + jj;
      const int p = CSW_LP(i, j);
      const long c2 = IDX2(g, i, j);
      const double fxc = m.dxc[c2] * L[PUC][p];
      const double fyc = m.dyc[c2] * L[PVC][p];
      const double fx1c = m.dxc[c2 - sj] * L[PUC][p - PW];
      const double fy1c = m.dyc[c2 - 1] * L[PVC][p - 1];
      L[PVT][p] = m.fC[c2] + m.rarea_c[c2] * (fx1c - fxc - fy1c + fyc);
    }
  }
  __syncthreads();
  CSW_STAMP(4);
  // ---- delp, pt, w take the planes of u, v, ua
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    const int e = tid + NT * t;
    if (e < NC1) {
      const int jj = e / (TI + 2), ii = e - jj * (TI + 2);
      const int p = CSW_LP(i0 - 1 + ii, j0 - 1 + jj);
      L[PU][p] = dpr[t];
      L[PV][p] = ptr_[t];
      L[PUA][p] = wr[t];
    }
  }
  __syncthreads();
  CSW_STAMP(5);
  // ---- pass C's transport and pass D at the own cells
#pragma unroll
  for (int t = 0; t < NO; ++t) {
    const int e = tid + NT * t;
    if (e >= NOWN) break;
    const int jj = e / TI, ii = e - jj * TI;
    const int i = i0 + ii, j = j0 + jj;
    const int p = CSW_LP(i, j);
    const long c2 = IDX2(g, i, j);
    const long c = kb + c2;
    double fx1[2], fx[2], fx2[2], fy1[2], fy[2], fy2[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const double utv = L[PUTC][p + q];
      const int s = (utv > 0.0) ? p + q - 1 : p + q;
      fx1[q] = utv * L[PU][s];
      fx[q] = fx1[q] * L[PV][s];
      fx2[q] = fx1[q] * L[PUA][s];
      const double vtv = L[PVTC][p + q * PW];
      const int s2 = (vtv > 0.0) ? p + (q - 1) * PW : p + q * PW;
      fy1[q] = vtv * L[PU][s2];
      fy[q] = fy1[q] * L[PV][s2];
      fy2[q] = fy1[q] * L[PUA][s2];
    }
    const double ra = m.rarea[c2];
    const double dp = L[PU][p];
    const double dpc = dp + (fx1[0] - fx1[1] + fy1[0] - fy1[1]) * ra;
    delpc[c] = dpc;
    ptc[c] = (L[PV][p] * dp + (fx[0] - fx[1] + fy[0] - fy[1]) * ra) / dpc;
    omga[c] = (L[PUA][p] * dp + (fx2[0] - fx2[1] + fy2[0] - fy2[1]) * ra) / dpc;
    {
      const double vc0 = L[PVC][p];
      const double tmp = dt2 * (ur[t] - vc0 * m.cosa_v[c2]) / m.sina_v[c2];
      const double flux = (tmp > 0.0) ? L[PVT][p] : L[PVT][p + 1];
      vc[c] = vc0 - tmp * flux + m.rdyc[c2] * (L[PUT][p - PW] - L[PUT][p]);
    }
    {
      const double uc0 = L[PUC][p];
      const double tmp = dt2 * (vr[t] - uc0 * m.cosa_u[c2]) / m.sina_u[c2];
      const double flux = (tmp > 0.0) ? L[PVT][p] : L[PVT][p + PW];
      uc[c] = uc0 + tmp * flux + m.rdxc[c2] * (L[PUT][p - 1] - L[PUT][p]);
    }
  }
  CSW_STAMP(7);
#undef CSW_LP
}

// the tiles of k_csw_tile on a grid: first own cell, counts (0 x 0: none -- the four passes take the whole plane)
struct CswTiles {
  int it0, jt0, ntx, nty;
  int i1() const { return it0 + ntx * CSW_TI - 1; }
  int j1() const { return jt0 + nty * CSW_TJ - 1; }
  bool any() const { return ntx > 0 && nty > 0; }
  // the box of cells at least `d` inside the tiled area (empty: mode 0 = skips nothing)
  SplitBox hole(int d) const {
    if (!any() || i1() - it0 + 1 <= 2 * d || j1() - jt0 + 1 <= 2 * d) return SplitBox{0, 0, 0, 0, 0};
    return SplitBox{it0 + d, i1() - d, jt0 + d, j1() - d, 2};
  }
};
static CswTiles csw_tiles(const Geo& g) {
  CswTiles t{g.is + 6, g.js + 6, (g.n - 11) / CSW_TI, (g.n - 11) / CSW_TJ};
  if (g.n < 12 || !t.any() || getenv("PACE_CSW_NO_TILES")) t.ntx = t.nty = 0;
  return t;
}
// the plain region [a0, a1]^2 of a pass minus the hole, as up to four rectangles (the first ones of `r`: nplain of them)
static void add_plain_with_hole(Regions& r, int a0, int a1, const SplitBox& h) {
  if (h.mode == 0) {
    add_region(r, a0, a1, a0, a1);
  } else {
    add_region(r, a0, a1, a0, h.j0 - 1);
    add_region(r, a0, a1, h.j1 + 1, a1);
    add_region(r, a0, h.i0 - 1, h.j0, h.j1);
    add_region(r, h.i1 + 1, a1, h.j0, h.j1);
  }
  r.nplain = r.n;
}

// interior box (every output point uses the plain 4-point formulas) + the four frame strips of the domain
// [is-1, ie+2] x [js-1, je+2] of pass B
static Regions d2a2c_regions(const Geo& g, const SplitBox& hole) {
  Regions r{};
  add_plain_with_hole(r, g.is + 2, g.ie - 1, hole);
  add_region(r, g.is - 1, g.is + 1, g.js - 1, g.je + 2);
  add_region(r, g.ie, g.ie + 2, g.js - 1, g.je + 2);
  add_region(r, g.is + 2, g.ie - 1, g.js - 1, g.js + 1);
  add_region(r, g.is + 2, g.ie - 1, g.je, g.je + 2);
  return r;
}

#define CSW_NFIELDS 6  // utmp, vtmp, ke, the vorticity, uc and vc before the update
int64_t csw_workspace_bytes(const Geo& g) { return (int64_t)g.sk * (g.nk + 1) * CSW_NFIELDS * (int64_t)sizeof(real); }

int launch_d2a2c_vect(const Geo& g, const Met& m, void* ws, real* uc, real* vc, const real* u, const real* v,
                      real* ua, real* va, real* utc, real* vtc, hipStream_t st) {
  if (g.n < 8) return PACE_ERR_UNSUPPORTED;  // npt = 4 branch of d2a2c_vect.py:421-424 only
  const long field = g.sk * (g.nk + 1);
  real* utmp = (real*)ws;
  real* vtmp = utmp + field;
  const dim3 grid = plane_grid(g, g.nk), block(256);
  hipLaunchKernelGGL(k_d2a2c_a, grid, block, 0, st, g, m, u, v, utmp, vtmp, ua, va, SplitBox{0, 0, 0, 0, 0});
  const Regions rb = d2a2c_regions(g, SplitBox{0, 0, 0, 0, 0});
  launch_d2a2c_b(g, m, u, v, utmp, vtmp, ua, va, uc, vc, utc, vtc, (real*)nullptr, 0.0, 0, 0, rb, st);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}

int launch_c_sw(const Geo& g, const Met& m, void* ws, real* delpc, real* ptc, real* delp, real* pt,
                const real* u, const real* v, real* w, real* uc, real* vc, real* ua, real* va,
                real* ut, real* vt, real* divgd, real* omga, double dt2, int nord, hipStream_t st, int part) {
  // part 0: everything.  Around the u / v halo exchange (dyn_core.py:744-745): 1 = what reads no halo value of u / v -- the
  // interior tiles (complete) and the points of the band's pass A in the box [is+1, ie-1] x [js+1, je-1]; 2 = the rest of pass A
  // and the band's passes B, C, D.
  if (g.n < 8) return PACE_ERR_UNSUPPORTED;
  const long field = g.sk * (g.nk + 1);
  real* utmp = (real*)ws;
  real* vtmp = utmp + field;
  real* ke = utmp + 2 * field;
  real* vort = utmp + 3 * field;
  real* ucw = utmp + 4 * field;  // pass B's uc, vc: the caller's arrays take the updated winds (pass D on the band, the tiles)
  real* vcw = utmp + 5 * field;
  const CswTiles tiles = csw_tiles(g);
  // A pass's band reaches as far into the tiled area as the passes after it read: A 4 cells, B 2, C 1 (what they write there --
  // ua, va, ut, vt, divgd, delpc, ptc, omga -- are the values the tiles write, too)
  const dim3 grid = plane_grid(g, g.nk), block(256);
  const SplitBox halo_box{g.is + 1, g.ie - 1, g.js + 1, g.je - 1, part};  // (pass A's points that read no halo value of u / v)
  auto pass_a = [&](hipStream_t sb) {
    const SplitBox h4 = part != 2 ? tiles.hole(4) : SplitBox{0, 0, 0, 0, 0};
    if (h4.mode != 0) {
      Regions ra{};  // the plane [0, ni-2] x [0, nj-2] without the hole
      add_plain_with_hole(ra, 0, g.ni - 2, h4);
      hipLaunchKernelGGL(k_d2a2c_a_band, regions_grid(ra, g.nk), dim3(64, 4), 0, sb, g, m, u, v, utmp, vtmp, ua, va, halo_box, ra);
    } else {
      hipLaunchKernelGGL(k_d2a2c_a, grid, block, 0, sb, g, m, u, v, utmp, vtmp, ua, va, halo_box);
    }
  };
  auto tile_kernel = [&](hipStream_t sb) {
    if (tiles.any())
      hipLaunchKernelGGL(k_csw_tile, dim3((unsigned)(8 * ((tiles.ntx * tiles.nty + 7) / 8) * g.nk)), dim3(CSW_NT), 0, sb, g, m, delp, pt, w, u, v,
                         delpc, ptc, omga, uc, vc, ua, va, ut, vt, divgd, dt2, nord > 0 ? 1 : 0, tiles.ntx, tiles.nty, tiles.it0, tiles.jt0);
  };
  auto passes_bcd = [&](hipStream_t sb) {
    const Regions rb = d2a2c_regions(g, tiles.hole(2));
    launch_d2a2c_b(g, m, u, v, utmp, vtmp, ua, va, ucw, vcw, ut, vt, divgd, dt2, nord > 0 ? 1 : 0, 1, rb, sb);
    Regions rt{};  // the plain box [is+1, ie-1]^2 (without the tiles) + the width-2 frame of the domain [is-1, ie+1] x [js-1, je+1]
    add_plain_with_hole(rt, g.is + 1, g.ie - 1, tiles.hole(1));
    add_region(rt, g.is - 1, g.is, g.js - 1, g.je + 1);
    add_region(rt, g.ie, g.ie + 1, g.js - 1, g.je + 1);
    add_region(rt, g.is + 1, g.ie - 1, g.js - 1, g.js);
    add_region(rt, g.is + 1, g.ie - 1, g.je, g.je + 1);
    hipLaunchKernelGGL(k_csw_transport, regions_grid(rt, g.nk), dim3(64, 4), 0, sb, g, m, delp, pt, w, u, v, ua, va, ucw, vcw, ut, vt,
                       delpc, ptc, omga, ke, vort, dt2, rt);
    const int nchunk = (g.nk + UV_CH - 1) / UV_CH;
    if (tiles.any()) {
      const int nchunk_band = (g.nk + UV_CH_BAND - 1) / UV_CH_BAND;
      Regions rd{};  // pass D's domain [is-1, ie+2]^2 without the tiles
      add_plain_with_hole(rd, g.is - 1, g.ie + 2, tiles.hole(0));
      hipLaunchKernelGGL(k_csw_update_uc_vc_band, regions_grid(rd, nchunk_band), dim3(64, 4), 0, sb, g, m, u, v, ke, vort, ucw, vcw, uc, vc, dt2, rd);
    } else {
      hipLaunchKernelGGL(k_csw_update_uc_vc, dim3(grid.x, (unsigned)nchunk, 1), block, 0, sb, g, m, u, v, ke, vort, ucw, vcw, uc, vc, dt2);
    }
    // (the cells next to the corners of delp, pt, w as the reference's in-place corner fills leave them: nothing here reads them)
    hipLaunchKernelGGL(k_csw_corner_cells, dim3((unsigned)g.nk), dim3(64), 0, sb, g, delp, pt, w);
  };
#ifndef PACE_EMU
  // The band's four passes are chains of dependent reads on a few thousand points (their edge forms: ~140 us at C192 however few
  // points there are) and need no LDS and few registers; the tile kernel is bound by what it issues and by its LDS.  Neither reads
  // what the other writes (pass B's winds go to the workspace; where both write -- the rim of the tiled area -- they write the same
  // values): the band runs on a stream of this thread's own, beside the tiles.
  const PaceSideStream* side = (part == 0 && tiles.any() && !getenv("PACE_CSW_ONE_STREAM")) ? pace_side_stream() : nullptr;
  if (side != nullptr) {
    if (!side->fork(st)) return PACE_ERR_LAUNCH;
    pass_a(side->s);
    passes_bcd(side->s);
    tile_kernel(st);
    if (!side->join(st)) return PACE_ERR_LAUNCH;
    PACE_CHECK_LAUNCH();
    return PACE_OK;
  }
#endif
  pass_a(st);
  if (part != 2) tile_kernel(st);
  if (part == 1) {
    PACE_CHECK_LAUNCH();
    return PACE_OK;
  }
  passes_bcd(st);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}
