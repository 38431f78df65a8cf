// The remaining operators of the acoustic substep (everything in dyn_core.py's loop that is not c_sw, d_sw or a
// Riemann solver).  All are HBM-bound streaming passes or column sweeps with lanes along i.
//   updatedzc.py:15-207      UpdateGeopotentialHeightOnCGrid   (4 launches -> 2)
//   updatedzd.py:70-356      UpdateHeightOnDGrid               (29 launches -> 4: spline, fvtp2d, delnflux, apply)
//   dyn_core.py:83-171       gz_from_surface_height_and_thicknesses, compute_geopotential, p_grad_c_stencil
//   nh_p_grad.py:11-255      NonHydrostaticPressureGradient    (51 launches -> 9)
//   pe_halo.py:6-34, pk3_halo.py:11-69                          ring column scans
//   ray_fast.py:48-206       RayleighDamping
//   del2cubed.py:16-194      HyperdiffusionDamping             (18 launches -> 3)
//   temperature_adjust.py:8-43 apply_diffusive_heating
#include <cmath>
#include <cstring>

#include "common.h"
#include "kernels.h"

#define GRAV 9.80665
#define RDGAS 287.05
#define CP_AIR 1004.6
#define CV_AIR (CP_AIR - RDGAS)
#define RDG (-RDGAS / GRAV)
#define DZ_MIN 2.0

// fill_corners_2cells_{x,y} with unit multipliers as read-side maps (corners.py:129-305); same maps as k_csw.hip
__device__ __forceinline__ long zc_xfill(const Geo& g, int i, int j) {
  if (j == g.js - 1) {
    if (i < g.is && i >= g.is - 2) return IDX2(g, g.is - 1, g.js + (g.is - i) - 1);
    if (i > g.ie && i <= g.ie + 2) return IDX2(g, g.ie + 1, g.js + (i - g.ie) - 1);
  } else if (j == g.je + 1) {
    if (i < g.is && i >= g.is - 2) return IDX2(g, g.is - 1, g.je + 1 - (g.is - i));
    if (i > g.ie && i <= g.ie + 2) return IDX2(g, g.ie + 1, g.je + 1 - (i - g.ie));
  }
  return IDX2(g, i, j);
}
__device__ __forceinline__ long zc_yfill(const Geo& g, int i, int j) {
  if (i == g.is - 1) {
    if (j < g.js && j >= g.js - 2) return IDX2(g, g.is + (g.js - j) - 1, g.js - 1);
    if (j > g.je && j <= g.je + 2) return IDX2(g, g.is + (j - g.je) - 1, g.je + 1);
  } else if (i == g.ie + 1) {
    if (j < g.js && j >= g.js - 2) return IDX2(g, g.ie + 1 - (g.js - j), g.js - 1);
    if (j > g.je && j <= g.je + 2) return IDX2(g, g.ie + 1 - (j - g.je), g.je + 1);
  }
  return IDX2(g, i, j);
}

// ------------------------------------------------------------------------------------------------
// updatedzc: interface winds by dp_ref-weighted averages (p_weighted_average_*, updatedzc.py:15-31), first-order
// upwind advection of gz (xy_flux :34-52, update_dz_c :61-117) on compute +- 1, interface levels
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double pavg(const real* __restrict__ f, long c, long sk, int k, int km,
                                       const real* __restrict__ dp) {
  if (k == 0) {
    const double ratio = dp[0] / (dp[0] + dp[1]);
    return f[c] + (f[c] - f[c + sk]) * ratio;
  }
  if (k == km) {
    const double ratio = dp[km - 1] / (dp[km - 2] + dp[km - 1]);
    return f[c - sk] + (f[c - sk] - f[c - 2 * sk]) * ratio;
  }
  const double int_ratio = 1.0 / (dp[k - 1] + dp[k]);
  return (dp[k] * f[c - sk] + dp[k - 1] * f[c]) * int_ratio;
}

// A thread takes ZC_CH consecutive interfaces of its column: the winds of layer k are the "layer above" of interface k + 1 and
// stay in registers (one thread per interface read every layer of ut / vt twice, from workgroups a whole plane apart).
#define ZC_CH 8
__global__ void __launch_bounds__(256)
k_updatedzc_advect(Geo g, Met m, const real* __restrict__ dp_ref, const real* __restrict__ ut,
                   const real* __restrict__ vt, const real* __restrict__ gz, real* __restrict__ gz_new) {
  const int i = (int)blockIdx.x * PATCH_W + (int)threadIdx.x;
  const int j = (int)blockIdx.y * PATCH_H + (int)threadIdx.y;
  const int k0 = (int)blockIdx.z * ZC_CH;
  if (j >= g.nj || i >= g.ni) return;
  if (i < g.is - 1 || i > g.ie + 1 || j < g.js - 1 || j > g.je + 1) return;
  const long c2 = IDX2(g, i, j);
  const int km = g.nk;
  const long sk = g.sk;
  const double area = m.area[c2];
  // (the four gz neighbours: the same places on every level)
  const long zxm = zc_xfill(g, i - 1, j), zx0 = zc_xfill(g, i, j), zxp = zc_xfill(g, i + 1, j);
  const long zym = zc_yfill(g, i, j - 1), zy0 = zc_yfill(g, i, j), zyp = zc_yfill(g, i, j + 1);
  double up[2] = {0.0, 0.0}, vp[2] = {0.0, 0.0};  // ut at (i, i + 1), vt at (j, j + 1) of the layer above the interface
  bool have = false;
#pragma unroll
  for (int t = 0; t < ZC_CH; ++t) {
    const int k = k0 + t;
    if (k > km) break;
    const long kb = (long)k * sk;
    const long c = c2 + kb;
    double xfx[2], yfx[2], fx[2], fy[2];
    if (k == 0 || k == km) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        xfx[q] = pavg(ut, c + q, sk, k, km, dp_ref);
        yfx[q] = pavg(vt, c + (long)q * g.sj, sk, k, km, dp_ref);
      }
      have = false;
    } else {
      double un[2], vn[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (!have) {
          up[q] = ut[c + q - sk];
          vp[q] = vt[c + (long)q * g.sj - sk];
        }
        un[q] = ut[c + q];
        vn[q] = vt[c + (long)q * g.sj];
      }
      const double int_ratio = 1.0 / (dp_ref[k - 1] + dp_ref[k]);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        xfx[q] = (dp_ref[k] * up[q] + dp_ref[k - 1] * un[q]) * int_ratio;  // (pavg's interior expression)
        yfx[q] = (dp_ref[k] * vp[q] + dp_ref[k - 1] * vn[q]) * int_ratio;
        up[q] = un[q];
        vp[q] = vn[q];
      }
      have = true;
    }
    fx[0] = xfx[0] * ((xfx[0] > 0.0) ? gz[kb + zxm] : gz[kb + zx0]);
    fx[1] = xfx[1] * ((xfx[1] > 0.0) ? gz[kb + zx0] : gz[kb + zxp]);
    fy[0] = yfx[0] * ((yfx[0] > 0.0) ? gz[kb + zym] : gz[kb + zy0]);
    fy[1] = yfx[1] * ((yfx[1] > 0.0) ? gz[kb + zy0] : gz[kb + zyp]);
    gz_new[c] = (gz[c] * area + fx[0] - fx[1] + fy[0] - fy[1]) / (area + xfx[0] - xfx[1] + yfx[0] - yfx[1]);
  }
}

// ws and the monotonicity sweep (updatedzc.py:108-117), columns of compute +- 1
__global__ void __launch_bounds__(64)
k_updatedzc_column(Geo g, const real* __restrict__ zs, const real* __restrict__ gz_new, real* __restrict__ gz,
                   real* __restrict__ ws, double rdt) {
  const int i = g.is - 1 + blockIdx.x * 64 + threadIdx.x;
  const int j = g.js - 1 + blockIdx.y;
  if (i > g.ie + 1 || j > g.je + 1) return;
  const long c0 = IDX2(g, i, j);
  const int km = g.nk;
  double below = gz_new[c0 + (long)km * g.sk];
  gz[c0 + (long)km * g.sk] = below;
  ws[c0] = (zs[c0] - below) * rdt;
  // forty levels' loads in flight: the kernel is one wave per 64 columns and a handful of waves per CU -- its time is the number
  // of load round trips of a column (one per level without the chunks, ten with chunks of eight: 24 us at C192 x 79)
  constexpr int CHC = 40;
  for (int k0 = km - 1; k0 >= 0; k0 -= CHC) {
    double v[CHC];
#pragma unroll
    for (int t = 0; t < CHC; ++t) v[t] = gz_new[c0 + (long)(k0 - t >= 0 ? k0 - t : 0) * g.sk];
#pragma unroll
    for (int t = 0; t < CHC; ++t)
      if (k0 - t >= 0) {
        const double lim = below + DZ_MIN;
        below = (v[t] > lim) ? v[t] : lim;
        gz[c0 + (long)(k0 - t) * g.sk] = below;
      }
  }
}

int64_t updatedzc_workspace_bytes(const Geo& g) { return (int64_t)g.sk * (g.nk + 1) * (int64_t)sizeof(real); }

int launch_updatedzc(const Geo& g, const Met& m, void* ws_, const real* dp_ref, const real* zs, const real* ut,
                     const real* vt, real* gz, real* ws, double dt, hipStream_t st) {
  if (g.nk < 3) return PACE_ERR_UNSUPPORTED;
  real* gz_new = (real*)ws_;
  hipLaunchKernelGGL(k_updatedzc_advect, patch_grid(g, (g.nk + 1 + ZC_CH - 1) / ZC_CH), PATCH_BLOCK, 0, st, g, m, dp_ref, ut, vt, gz, gz_new);
  hipLaunchKernelGGL(k_updatedzc_column, dim3((g.n + 2 + 63) / 64, g.n + 2), dim3(64), 0, st, g, zs, gz_new, gz, ws, 1.0 / dt);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}

// ------------------------------------------------------------------------------------------------
// updatedzd
// ------------------------------------------------------------------------------------------------
struct SplineK {
  const real *gk, *beta, *gamma;  // device K-arrays (updatedzd.py:129-154)
  double xt1_top, a_bot, xt1_bot, xt2_bot;
};

// cubic_spline_interpolation_from_layer_center_to_interfaces (updatedzd.py:157-196), full domain; blockIdx.z picks
// one of the four fields
__global__ void __launch_bounds__(64)
k_spline_to_interfaces(Geo g, SplineK s, const real* __restrict__ q0, const real* __restrict__ q1,
                       const real* __restrict__ q2, const real* __restrict__ q3, real* __restrict__ o0,
                       real* __restrict__ o1, real* __restrict__ o2, real* __restrict__ o3) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  const int j = blockIdx.y;
  if (i > g.ni - 2 || j > g.nj - 2) return;
  const real* qc = (blockIdx.z == 0) ? q0 : (blockIdx.z == 1) ? q1 : (blockIdx.z == 2) ? q2 : q3;
  real* qi = (blockIdx.z == 0) ? o0 : (blockIdx.z == 1) ? o1 : (blockIdx.z == 2) ? o2 : o3;
  const long c0 = IDX2(g, i, j);
  const long sk = g.sk;
  const int km = g.nk;
  // both sweeps in register chunks: the loads of a chunk are issued together, the recurrence then runs out of registers
  constexpr int CHS = 16;
  double prev_c = qc[c0];
  double v = (s.xt1_top * prev_c + qc[c0 + sk]) / s.beta[0];
  qi[c0] = v;
  for (int k0 = 1; k0 < km; k0 += CHS) {
    double c_[CHS];
#pragma unroll
    for (int t = 0; t < CHS; ++t) c_[t] = qc[c0 + (long)((k0 + t < km) ? k0 + t : km - 1) * sk];
#pragma unroll
    for (int t = 0; t < CHS; ++t) {
      const int k = k0 + t;
      if (k < km) {
        v = (3.0 * (prev_c + s.gk[k] * c_[t]) - v) / s.beta[k];
        qi[c0 + (long)k * sk] = v;
        prev_c = c_[t];
      }
    }
  }
  v = (s.xt1_bot * qc[c0 + (long)(km - 1) * sk] + qc[c0 + (long)(km - 2) * sk] - s.a_bot * v) / s.xt2_bot;
  qi[c0 + (long)km * sk] = v;
  for (int k0 = km - 1; k0 >= 0; k0 -= CHS) {
    double i_[CHS];
#pragma unroll
    for (int t = 0; t < CHS; ++t) i_[t] = qi[c0 + (long)((k0 - t >= 0) ? k0 - t : 0) * sk];
#pragma unroll
    for (int t = 0; t < CHS; ++t) {
      const int k = k0 - t;
      if (k >= 0) {
        v = i_[t] - s.gamma[k] * v;
        qi[c0 + (long)k * sk] = v;
      }
    }
  }
}

// The same with the forward sweep's values parked in LDS instead of in the output field: the backward sweep then reads them
// from there, and the output is written once (2 field passes per field instead of 4: 407 -> ~210 MB for the four fields).
#define SPL_MAXK 96
__global__ void __launch_bounds__(64)
k_spline_to_interfaces_lds(Geo g, SplineK s, const real* __restrict__ q0, const real* __restrict__ q1,
                           const real* __restrict__ q2, const real* __restrict__ q3, real* __restrict__ o0,
                           real* __restrict__ o1, real* __restrict__ o2, real* __restrict__ o3) {
  __shared__ double sv[SPL_MAXK][64];
  const int tx = threadIdx.x;
  const int i = blockIdx.x * 64 + tx;
  const int j = blockIdx.y;
  if (i > g.ni - 2 || j > g.nj - 2) return;  // (no barrier in this kernel: a thread only ever reads its own LDS column)
  const real* qc = (blockIdx.z == 0) ? q0 : (blockIdx.z == 1) ? q1 : (blockIdx.z == 2) ? q2 : q3;
  real* qi = (blockIdx.z == 0) ? o0 : (blockIdx.z == 1) ? o1 : (blockIdx.z == 2) ? o2 : o3;
  const long c0 = IDX2(g, i, j);
  const long sk = g.sk;
  const int km = g.nk;
  constexpr int CHS = 16;
  double prev_c = qc[c0];
  double v = (s.xt1_top * prev_c + qc[c0 + sk]) / s.beta[0];
  sv[0][tx] = v;
  double c_last = prev_c, c_last2 = prev_c;  // qc[km-1], qc[km-2] for the bottom closure
  for (int k0 = 1; k0 < km; k0 += CHS) {
    double c_[CHS];
#pragma unroll
    for (int t = 0; t < CHS; ++t) c_[t] = qc[c0 + (long)((k0 + t < km) ? k0 + t : km - 1) * sk];
#pragma unroll
    for (int t = 0; t < CHS; ++t) {
      const int k = k0 + t;
      if (k < km) {
        v = (3.0 * (prev_c + s.gk[k] * c_[t]) - v) / s.beta[k];
        sv[k][tx] = v;
        c_last2 = prev_c;
        prev_c = c_[t];
        c_last = c_[t];
      }
    }
  }
  v = (s.xt1_bot * c_last + c_last2 - s.a_bot * v) / s.xt2_bot;
  qi[c0 + (long)km * sk] = v;
  for (int k = km - 1; k >= 0; --k) {
    v = sv[k][tx] - s.gamma[k] * v;
    qi[c0 + (long)k * sk] = v;
  }
}


// The same with the forward sweep's values in REGISTERS (km <= 80): the LDS form holds 48 KB per 64 columns -- three waves per
// CU, each a serial chain of km divisions: 86 us at C192 x 79 for 205 MB moved.  160 registers of forward values leave two
// waves per SIMD (eight per CU), and the levels are read sixteen at a time ahead of the chain.
#define SPL_REGK 80
__global__ void __launch_bounds__(64, 2)
k_spline_to_interfaces_regs(Geo g, SplineK s, const real* __restrict__ q0, const real* __restrict__ q1,
                            const real* __restrict__ q2, const real* __restrict__ q3, real* __restrict__ o0,
                            real* __restrict__ o1, real* __restrict__ o2, real* __restrict__ o3) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  const int j = blockIdx.y;
  if (i > g.ni - 2 || j > g.nj - 2) return;
  const real* qc = (blockIdx.z == 0) ? q0 : (blockIdx.z == 1) ? q1 : (blockIdx.z == 2) ? q2 : q3;
  real* qi = (blockIdx.z == 0) ? o0 : (blockIdx.z == 1) ? o1 : (blockIdx.z == 2) ? o2 : o3;
  const long c0 = IDX2(g, i, j);
  const long sk = g.sk;
  const int km = g.nk;
  constexpr int CHS = 32;
  double sv[SPL_REGK];
  double prev_c = qc[c0];
  double v = (s.xt1_top * prev_c + qc[c0 + sk]) / s.beta[0];
  sv[0] = v;
  double c_last = prev_c, c_last2 = prev_c;  // qc[km-1], qc[km-2] for the bottom closure
#pragma unroll
  for (int k0 = 1; k0 < SPL_REGK; k0 += CHS) {
    if (k0 < km) {
      double c_[CHS];
#pragma unroll
      for (int t = 0; t < CHS; ++t) c_[t] = qc[c0 + (long)((k0 + t < km) ? k0 + t : km - 1) * sk];
#pragma unroll
      for (int t = 0; t < CHS; ++t) {
        const int k = k0 + t;
        if (k < SPL_REGK && k < km) {
          v = (3.0 * (prev_c + s.gk[k] * c_[t]) - v) / s.beta[k];
          sv[k] = v;
          c_last2 = prev_c;
          prev_c = c_[t];
          c_last = c_[t];
        }
      }
    }
  }
  v = (s.xt1_bot * c_last + c_last2 - s.a_bot * v) / s.xt2_bot;
  qi[c0 + (long)km * sk] = v;
#pragma unroll
  for (int k = SPL_REGK - 1; k >= 0; --k) {
    if (k < km) {
      v = sv[k] - s.gamma[k] * v;
      qi[c0 + (long)k * sk] = v;
    }
  }
}

// apply_height_fluxes (updatedzd.py:70-126) in two steps: the advective + diffusive update is a point function (all
// levels in parallel, in place: a cell reads only its own zh); ws and the bottom-up monotonicity sweep are the only
// column-sequential part and touch one field.
__global__ void __launch_bounds__(256)
k_apply_height_fluxes(Geo g, Met m, real* __restrict__ zh, const real* __restrict__ fx, const real* __restrict__ fy,
                      const real* __restrict__ xfx, const real* __restrict__ yfx, const real* __restrict__ fx2,
                      const real* __restrict__ fy2) {
  PATCH_IJK(g);
  if (i < g.is || i > g.ie || j < g.js || j > g.je) return;
  const long c = IDX3(g, i, j, k);
  const double area = m.area[IDX2(g, i, j)];
  const double area_after = (area + xfx[c] - xfx[c + 1]) + (area + yfx[c] - yfx[c + g.sj]) - area;
  const double adv = (zh[c] * area + fx[c] - fx[c + 1] + fy[c] - fy[c + g.sj]) / area_after;
  zh[c] = adv + (fx2[c] - fx2[c + 1] + fy2[c] - fy2[c + g.sj]) / area;
}

// column part of update_dz_d (updatedzd.py:56-67 after the flux update): ws from the bottom interface, then the heights kept
// at least DZ_MIN apart from the bottom up.  `zin` holds the flux-updated heights (the transport's epilogue wrote them to a
// scratch field), `zh` receives the result.
__global__ void __launch_bounds__(64)
k_height_column(Geo g, const real* __restrict__ zs, const real* zin, real* zh, real* __restrict__ ws, double dt) {
  // (zin may be zh itself: no __restrict__ on the two)
  const int i = g.is + blockIdx.x * 64 + threadIdx.x;
  const int j = g.js + blockIdx.y;
  if (i > g.ie || j > g.je) return;
  const long c0 = IDX2(g, i, j);
  const int km = g.nk;
  constexpr int CHZ = 40;  // (levels' loads in flight: see k_updatedzc_column)
  double below = zin[c0 + (long)km * g.sk];
  zh[c0 + (long)km * g.sk] = below;
  ws[c0] = (zs[c0] - below) / dt;
  for (int k0 = km - 1; k0 >= 0; k0 -= CHZ) {
    double z_[CHZ];
#pragma unroll
    for (int t = 0; t < CHZ; ++t) z_[t] = zin[c0 + (long)((k0 - t >= 0) ? k0 - t : 0) * g.sk];
#pragma unroll
    for (int t = 0; t < CHZ; ++t) {
      const int k = k0 - t;
      if (k >= 0) {
        const double other = below + DZ_MIN;
        const double v = (z_[t] > other) ? z_[t] : other;
        zh[c0 + (long)k * g.sk] = v;
        below = v;
      }
    }
  }
}

#define UZD_NFIELDS 8
int64_t updatedzd_workspace_bytes(const Geo& g) {
  return (int64_t)g.sk * (g.nk + 1) * UZD_NFIELDS * (int64_t)sizeof(real);
}

int launch_updatedzd(const Geo& g, const Met& m, void* ws_, const pace_updatedzd_k_t* kc, const real* zs, real* zh,
                     const real* crx, const real* cry, const real* xfx, const real* yfx, real* wsd, double dt,
                     int hord_tm, hipStream_t st) {
  if (g.nk < 3) return PACE_ERR_UNSUPPORTED;
  const long field = g.sk * (g.nk + 1);
  real* p = (real*)ws_;
  real *crx_i = p, *cry_i = p + field, *xfx_i = p + 2 * field, *yfx_i = p + 3 * field, *fx = p + 4 * field,
         *fy = p + 5 * field, *fx2 = p + 6 * field, *fy2 = p + 7 * field;
  SplineK s{kc->gk, kc->beta, kc->gamma, kc->xt1_top, kc->a_bot, kc->xt1_bot, kc->xt2_bot};
  if (g.nk <= SPL_REGK && g.nk >= 3)
    hipLaunchKernelGGL(k_spline_to_interfaces_regs, dim3((g.ni - 1 + 63) / 64, g.nj - 1, 4), dim3(64), 0, st, g, s, crx, cry, xfx,
                       yfx, crx_i, cry_i, xfx_i, yfx_i);
  else if (g.nk <= SPL_MAXK && g.nk >= 3)
    hipLaunchKernelGGL(k_spline_to_interfaces_lds, dim3((g.ni - 1 + 63) / 64, g.nj - 1, 4), dim3(64), 0, st, g, s, crx, cry, xfx,
                       yfx, crx_i, cry_i, xfx_i, yfx_i);
  else
    hipLaunchKernelGGL(k_spline_to_interfaces, dim3((g.ni - 1 + 63) / 64, g.nj - 1, 4), dim3(64), 0, st, g, s, crx, cry, xfx, yfx,
                       crx_i, cry_i, xfx_i, yfx_i);
  int rc;
  if (hord_tm == 5 || hord_tm == 6) {
    // transport of the heights, del-n damping of the same field and apply_height_fluxes in ONE kernel (the transport kernel's
    // height epilogue): the four flux fields are never written; the updated heights go to a scratch field (`fx`) because
    // neighbouring tiles still read zh, and the column kernel moves them back (14 field passes -> 4)
    FvDamp dp{};
    dp.damp_k = kc->damp; dp.nord_k = kc->nord; dp.nmax = kc->nmax; dp.mass_given = 0;
    dp.qout = fx;
    if ((rc = launch_transport(g, m, zh, crx_i, cry_i, xfx_i, yfx_i, nullptr, nullptr, nullptr, nullptr, hord_tm, g.nk + 1, 0, 3, dp,
                               st)))
      return rc;
    hipLaunchKernelGGL(k_height_column, dim3((g.n + 63) / 64, g.n), dim3(64), 0, st, g, zs, fx, zh, wsd, dt);
  } else {
    if ((rc = launch_fvtp2d(g, m, zh, crx_i, cry_i, xfx_i, yfx_i, fx, fy, nullptr, nullptr, hord_tm, g.nk + 1, st))) return rc;
    if ((rc = launch_delnflux(g, m, 0, zh, fx2, fy2, nullptr, kc->damp, kc->nord, kc->nmax, 0, g.nk + 1, st))) return rc;
    hipLaunchKernelGGL(k_apply_height_fluxes, patch_grid(g, g.nk + 1), PATCH_BLOCK, 0, st, g, m, zh, fx, fy, xfx_i, yfx_i, fx2, fy2);
    hipLaunchKernelGGL(k_height_column, dim3((g.n + 63) / 64, g.n), dim3(64), 0, st, g, zs, zh, zh, wsd, dt);
  }
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}

// ------------------------------------------------------------------------------------------------
// dyn_core.py small stencils
// ------------------------------------------------------------------------------------------------
// gz_from_surface_height_and_thicknesses (dyn_core.py:83-96), compute domain
__global__ void __launch_bounds__(64)
k_gz_from_surface(Geo g, const real* __restrict__ zs, const real* __restrict__ delz, real* __restrict__ gz) {
  const int i = g.is + blockIdx.x * 64 + threadIdx.x;
  const int j = g.js + blockIdx.y;
  if (i > g.ie || j > g.je) return;
  const long c0 = IDX2(g, i, j);
  double z = zs[c0];
  gz[c0 + (long)g.nk * g.sk] = z;
  for (int k = g.nk - 1; k >= 0; --k) {
    z = z - delz[c0 + (long)k * g.sk];
    gz[c0 + (long)k * g.sk] = z;
  }
}

// zero_data (dyn_core.py:51-80): the flux accumulators on the full domain, the heat terms on the compute domain
__global__ void __launch_bounds__(256)
k_zero_data(Geo g, real* __restrict__ mfxd, real* __restrict__ mfyd, real* __restrict__ cxd, real* __restrict__ cyd,
            real* __restrict__ heat_source, real* __restrict__ diss_estd, int first_timestep) {
  PLANE_IJK(g);
  if (i > g.ni - 2 || j > g.nj - 2) return;
  const long c = IDX3(g, i, j, k);
  mfxd[c] = 0.0;
  mfyd[c] = 0.0;
  cxd[c] = 0.0;
  cyd[c] = 0.0;
  if (first_timestep && i >= g.is && i <= g.ie && j >= g.js && j <= g.je) {
    heat_source[c] = 0.0;
    diss_estd[c] = 0.0;
  }
}

// interface_pressure_from_toa_pressure_and_thickness (dyn_core.py:99-112), compute domain +- 1
__global__ void __launch_bounds__(64)
k_interface_pressure(Geo g, const real* __restrict__ delp, real* __restrict__ pem, double ptop) {
  const int i = g.is - 1 + blockIdx.x * 64 + threadIdx.x;
  const int j = g.js - 1 + blockIdx.y;
  if (i > g.ie + 1 || j > g.je + 1) return;
  const long c0 = IDX2(g, i, j);
  double p = ptop;
  pem[c0] = p;
  for (int k = 1; k < g.nk; ++k) {
    p = p + delp[c0 + (long)k * g.sk];
    pem[c0 + (long)k * g.sk] = p;
  }
}

int launch_zero_data(const Geo& g, real* mfxd, real* mfyd, real* cxd, real* cyd, real* heat_source, real* diss_estd,
                     int first_timestep, hipStream_t st) {
  hipLaunchKernelGGL(k_zero_data, plane_grid(g, g.nk), dim3(256), 0, st, g, mfxd, mfyd, cxd, cyd, heat_source, diss_estd,
                     first_timestep);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}
int launch_interface_pressure(const Geo& g, const real* delp, real* pem, double ptop, hipStream_t st) {
  hipLaunchKernelGGL(k_interface_pressure, dim3((g.n + 2 + 63) / 64, g.n + 2), dim3(64), 0, st, g, delp, pem, ptop);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}

// dst = src * factor on the window [i0, i1] x [j0, j1], nlev levels: copy_defn (basic_operations.py:7, factor 1)
// and compute_geopotential (dyn_core.py:115-117, factor GRAV)
// (two neighbouring elements per thread: one 16-byte load and store per lane -- this repo's stream kernels move 6.2 TB/s that way
// against 5.2 with 8 bytes per lane, profiles/r05_ubench_streams.txt; the rows are 16-byte aligned: sj is even)
struct alignas(2 * sizeof(real)) ScPair {
  real x, y;
};
__global__ void __launch_bounds__(256)
k_scale_copy(Geo g, const real* __restrict__ src, real* __restrict__ dst, double factor, int scale, int i0, int i1,
             int j0, int j1) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;  // pair index in the plane
  const int half = (g.sj + 1) / 2;  // (an odd row stride: the last pair of a row is its last element alone)
  const int j = (int)(p / half);
  const int i = 2 * (int)(p - (long)j * half);
  const int k = (int)blockIdx.y;
  if (j >= g.nj || j < j0 || j > j1 || i + 1 < i0 || i > i1) return;
  const long c = IDX3(g, i, j, k);
  const bool lo = i >= i0, hi = i + 1 <= i1 && i + 1 < g.sj;
  // the pair form needs the ELEMENT aligned, not only the bases: even row and level strides (the C ABI accepts odd ones)
  if (lo && hi && (((uintptr_t)src | (uintptr_t)dst) & (2 * sizeof(real) - 1)) == 0 && ((g.sj | g.sk) & 1) == 0) {
    ScPair v = *(const ScPair*)(src + c);
    if (scale) v.x = (real)(v.x * factor), v.y = (real)(v.y * factor);
    *(ScPair*)(dst + c) = v;
  } else {
    if (lo) dst[c] = scale ? src[c] * factor : src[c];
    if (hi) dst[c + 1] = scale ? src[c + 1] * factor : src[c + 1];
  }
}
static inline dim3 pair_grid(const Geo& g, int nlev) {
  return dim3((unsigned)(((long)((g.sj + 1) / 2) * g.nj + 255) / 256), (unsigned)nlev, 1);
}

// p_grad_c_stencil (dyn_core.py:120-171), non-hydrostatic; compute + 1.  A thread takes PG_CH consecutive layers of its point:
// the interface values below a layer are the ones above the next and stay in registers (k_nh_uv's arrangement).
#define PG_CH 8
__global__ void __launch_bounds__(256)
k_p_grad_c(Geo g, Met m, real* __restrict__ uc, real* __restrict__ vc, const real* __restrict__ delpc,
           const real* __restrict__ pkc, const real* __restrict__ gz, double dt2) {
  const int i = (int)blockIdx.x * PATCH_W + (int)threadIdx.x;
  const int j = (int)blockIdx.y * PATCH_H + (int)threadIdx.y;
  const int k0 = (int)blockIdx.z * PG_CH;
  if (j >= g.nj || i >= g.ni) return;
  if (i < g.is || i > g.ie + 1 || j < g.js || j > g.je + 1) return;
  const long c2 = IDX2(g, i, j);
  const long sk = g.sk;
  const int sj = g.sj;
  const double rdxc = m.rdxc[c2], rdyc = m.rdyc[c2];
  long c = IDX3(g, i, j, k0);
  // gz and pkc at (i, j), (i - 1, j), (i, j - 1) on the interface above the layer
  double gz_0 = gz[c], gzx_0 = gz[c - 1], gzy_0 = gz[c - sj];
  double pk_0 = pkc[c], pkx_0 = pkc[c - 1], pky_0 = pkc[c - sj];
#pragma unroll
  for (int t = 0; t < PG_CH; ++t) {
    if (k0 + t >= g.nk) break;
    const double gz_1 = gz[c + sk], gzx_1 = gz[c - 1 + sk], gzy_1 = gz[c - sj + sk];
    const double pk_1 = pkc[c + sk], pkx_1 = pkc[c - 1 + sk], pky_1 = pkc[c - sj + sk];
    const double d0 = delpc[c];
    uc[c] = uc[c] + dt2 * rdxc / (delpc[c - 1] + d0) * ((gzx_1 - gz_0) * (pk_1 - pkx_0) + (gzx_0 - gz_1) * (pkx_1 - pk_0));
    vc[c] = vc[c] + dt2 * rdyc / (delpc[c - sj] + d0) * ((gzy_1 - gz_0) * (pk_1 - pky_0) + (gzy_0 - gz_1) * (pky_1 - pk_0));
    gz_0 = gz_1, gzx_0 = gzx_1, gzy_0 = gzy_1;
    pk_0 = pk_1, pkx_0 = pkx_1, pky_0 = pky_1;
    c += sk;
  }
}

int launch_gz_from_surface(const Geo& g, const real* zs, const real* delz, real* gz, hipStream_t st) {
  hipLaunchKernelGGL(k_gz_from_surface, dim3((g.n + 63) / 64, g.n), dim3(64), 0, st, g, zs, delz, gz);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}
int launch_scale_copy(const Geo& g, const real* src, real* dst, double factor, int scale, int halo, int nlev,
                      hipStream_t st) {
  hipLaunchKernelGGL(k_scale_copy, pair_grid(g, nlev), dim3(256), 0, st, g, src, dst, factor, scale, g.is - halo, g.ie + halo,
                     g.js - halo, g.je + halo);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}
int launch_p_grad_c(const Geo& g, const Met& m, real* uc, real* vc, const real* delpc, const real* pkc,
                    const real* gz, double dt2, hipStream_t st) {
  hipLaunchKernelGGL(k_p_grad_c, patch_grid(g, (g.nk + PG_CH - 1) / PG_CH), PATCH_BLOCK, 0, st, g, m, uc, vc, delpc, pkc, gz, dt2);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}

// ------------------------------------------------------------------------------------------------
// nh_p_grad
// ------------------------------------------------------------------------------------------------
// set_k0_and_calc_wk (nh_p_grad.py:11-26), level 0 only; wk itself is recomputed where it is used
__global__ void __launch_bounds__(256)
k_nh_set_top(Geo g, real* __restrict__ pp, real* __restrict__ pk3, double top_value) {
  PLANE_IJK(g);
  (void)k;
  if (i < g.is || i > g.ie + 1 || j < g.js || j > g.je + 1) return;
  const long c = IDX2(g, i, j);
  pp[c] = 0.0;
  pk3[c] = top_value;
}

// calc_u (nh_p_grad.py:29-69), calc_v (:72-112).  The B-grid values of gz, pk3, pp (and of delp: wk1) come from the
// a2b_ord4 launches before it; with WRITE_BACK they are stored over the caller's gz / pk3 / pp here (a2b_ord4's replace = True,
// a2b_ord4.py:505-506: the reference leaves the interpolated values in its arguments) instead of by three copy kernels.
// A thread takes NH_CH consecutive levels of its corner: the interface values of level k + 1 are level k's of the next
// turn and stay in registers (one thread per level read every interface of gz / pk3 / pp twice, from workgroups a whole plane
// apart: 324 MB counted for 14 field passes of 23.6 MB).
#define NH_CH 8
template <bool WRITE_BACK>
__global__ void __launch_bounds__(256)
k_nh_uv(Geo g, Met m, real* __restrict__ u, real* __restrict__ v, const real* __restrict__ wk1,
        const real* __restrict__ gz, const real* __restrict__ pk3, const real* __restrict__ pp, double dt,
        real* __restrict__ gz_out, real* __restrict__ pk3_out, real* __restrict__ pp_out) {
  const int i = (int)blockIdx.x * PATCH_W + (int)threadIdx.x;
  const int j = (int)blockIdx.y * PATCH_H + (int)threadIdx.y;
  const int k0 = (int)blockIdx.z * NH_CH;
  if (j >= g.nj || i >= g.ni) return;
  if (i < g.is || i > g.ie + 1 || j < g.js || j > g.je + 1) return;
  const long c2 = IDX2(g, i, j);
  const long sk = g.sk;
  const int sj = g.sj;
  const bool do_u = i <= g.ie, do_v = j <= g.je;
  const double rdx = do_u ? (double)m.rdx[c2] : 0.0, rdy = do_v ? (double)m.rdy[c2] : 0.0;
  // (places a thread does not use -- beyond the last row / column of corners -- read its own corner again)
  const long dxo = do_u ? 1 : 0, dyo = do_v ? sj : 0;
  long c = IDX3(g, i, j, k0);
  double pk_0 = pk3[c], gz_0 = gz[c], pp_0 = pp[c];
  double pkx_0 = pk3[c + dxo], gzx_0 = gz[c + dxo], ppx_0 = pp[c + dxo];
  double pky_0 = pk3[c + dyo], gzy_0 = gz[c + dyo], ppy_0 = pp[c + dyo];
#pragma unroll
  for (int t = 0; t < NH_CH; ++t) {
    const int k = k0 + t;
    if (k >= g.nk) break;
    const double pk_1 = pk3[c + sk], gz_1 = gz[c + sk], pp_1 = pp[c + sk];
    const double pkx_1 = pk3[c + dxo + sk], gzx_1 = gz[c + dxo + sk], ppx_1 = pp[c + dxo + sk];
    const double pky_1 = pk3[c + dyo + sk], gzy_1 = gz[c + dyo + sk], ppy_1 = pp[c + dyo + sk];
    const double wk0 = pk_1 - pk_0;
    const double w0 = wk1[c];
    if (do_u) {
      const double wkx = pkx_1 - pkx_0;
      const double du = dt / (wk0 + wkx) * ((gz_1 - gzx_0) * (pkx_1 - pk_0) + (gz_0 - gzx_1) * (pk_1 - pkx_0));
      u[c] = (u[c] + du + dt / (w0 + wk1[c + 1]) * ((gz_1 - gzx_0) * (ppx_1 - pp_0) + (gz_0 - gzx_1) * (pp_1 - ppx_0))) * rdx;
    }
    if (do_v) {
      const double wky = pky_1 - pky_0;
      const double dv = dt / (wk0 + wky) * ((gz_1 - gzy_0) * (pky_1 - pk_0) + (gz_0 - gzy_1) * (pk_1 - pky_0));
      v[c] = (v[c] + dv + dt / (w0 + wk1[c + sj]) * ((gz_1 - gzy_0) * (ppy_1 - pp_0) + (gz_0 - gzy_1) * (pp_1 - ppy_0))) * rdy;
    }
    if (WRITE_BACK) {
      gz_out[c] = gz_0;
      pk3_out[c] = pk_0;
      pp_out[c] = pp_0;
      if (k == g.nk - 1) {
        gz_out[c + sk] = gz_1;
        pk3_out[c + sk] = pk_1;
        pp_out[c + sk] = pp_1;
      }
    }
    pk_0 = pk_1, gz_0 = gz_1, pp_0 = pp_1;
    pkx_0 = pkx_1, gzx_0 = gzx_1, ppx_0 = ppx_1;
    pky_0 = pky_1, gzy_0 = gzy_1, ppy_0 = ppy_1;
    c += sk;
  }
}

// workspace: the four B-grid fields (pp, pk3, gz, delp interpolated to the cell corners)
int64_t nh_p_grad_workspace_bytes(const Geo& g) { return 4 * (int64_t)g.sk * (g.nk + 1) * (int64_t)sizeof(real); }

int launch_nh_p_grad(const Geo& g, const Met& m, void* ws_, real* u, real* v, real* pp, real* gz, real* pk3,
                     real* delp, double dt, double ptop, double akap, hipStream_t st) {
  const long field = (long)g.sk * (g.nk + 1);
  real* pp_b = (real*)ws_;
  real* pk3_b = pp_b + field;
  real* gz_b = pk3_b + field;
  real* wk1 = gz_b + field;
  const int K = g.nk + 1;
  const real* in[4] = {pp, pk3, gz, delp};
  real* out[4] = {pp_b, pk3_b, gz_b, wk1};
  const int k0[4] = {1, 1, 0, 0}, k1[4] = {K, K, K, g.nk};
  const int rc = launch_a2b_ord4_batch(g, m, in, out, k0, k1, 4, st);
  if (rc) return rc;
  const double top_value = pow(ptop, akap);
  hipLaunchKernelGGL(k_nh_set_top, plane_grid(g, 1), dim3(256), 0, st, g, pp_b, pk3_b, top_value);
  hipLaunchKernelGGL(k_nh_uv<true>, patch_grid(g, (g.nk + NH_CH - 1) / NH_CH), PATCH_BLOCK, 0, st, g, m, u, v, wk1, gz_b, pk3_b, pp_b, dt, gz, pk3, pp);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}

// ------------------------------------------------------------------------------------------------
// pe_halo.edge_pe (pe_halo.py:6-34) and PK3Halo (pk3_halo.py:11-69): forward k scans in the ring of width
// `width` around the compute domain.  One thread per ring column.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool ring_cell(const Geo& g, int width, int t, int& i, int& j) {
  // enumerate the (n + 2 width)^2 - n^2 ring cells: `width` full rows south, `width` north, then west/east strips
  const int span = g.n + 2 * width;
  const int nrow = span * width;
  if (t < nrow) { j = g.js - width + t / span; i = g.is - width + t % span; return true; }
  t -= nrow;
  if (t < nrow) { j = g.je + 1 + t / span; i = g.is - width + t % span; return true; }
  t -= nrow;
  const int ncol = g.n * width;
  if (t < ncol) { j = g.js + t / width; i = g.is - width + t % width; return true; }
  t -= ncol;
  if (t < ncol) { j = g.js + t / width; i = g.ie + 1 + t % width; return true; }
  return false;
}

// (the level loop loads RING_CH levels ahead of the running sum: a plain load - add - store loop exposes one memory latency per
// level, 79 of them per ring column: 44 us for PK3Halo's 1 552 columns at C192)
#define RING_CH 40
__global__ void __launch_bounds__(64)
k_edge_pe(Geo g, real* __restrict__ pe, const real* __restrict__ delp, double ptop) {
  int i, j;
  if (!ring_cell(g, 1, blockIdx.x * 64 + threadIdx.x, i, j)) return;
  const long c0 = IDX2(g, i, j);
  double p = ptop;
  pe[c0] = p;
  for (int k0 = 1; k0 <= g.nk; k0 += RING_CH) {
    double d[RING_CH];
#pragma unroll
    for (int t = 0; t < RING_CH; ++t) d[t] = delp[c0 + (long)((k0 + t <= g.nk ? k0 + t : g.nk) - 1) * g.sk];
#pragma unroll
    for (int t = 0; t < RING_CH; ++t)
      if (k0 + t <= g.nk) {
        p = p + d[t];
        pe[c0 + (long)(k0 + t) * g.sk] = p;
      }
  }
}

// PK3Halo in two steps: the pressure scan is sequential per ring column (cheap), the power is not -- ~1600 threads
// doing 79 dependent pow() each was 68 us; all (column, level) pairs in parallel is a few.
__global__ void __launch_bounds__(64)
k_pk3_halo_scan(Geo g, real* __restrict__ pk3, const real* __restrict__ delp, double ptop) {
  int i, j;
  if (!ring_cell(g, 2, blockIdx.x * 64 + threadIdx.x, i, j)) return;
  const long c0 = IDX2(g, i, j);
  double p = ptop;
  for (int k0 = 1; k0 <= g.nk; k0 += RING_CH) {
    double d[RING_CH];
#pragma unroll
    for (int t = 0; t < RING_CH; ++t) d[t] = delp[c0 + (long)((k0 + t <= g.nk ? k0 + t : g.nk) - 1) * g.sk];
#pragma unroll
    for (int t = 0; t < RING_CH; ++t)
      if (k0 + t <= g.nk) {
        p = p + d[t];
        pk3[c0 + (long)(k0 + t) * g.sk] = p;  // pe for now
      }
  }
}

__global__ void __launch_bounds__(64)
k_pk3_halo_pow(Geo g, real* __restrict__ pk3, double akap) {
  int i, j;
  if (!ring_cell(g, 2, blockIdx.x * 64 + threadIdx.x, i, j)) return;
  const long c = IDX2(g, i, j) + (long)(blockIdx.y + 1) * g.sk;
  pk3[c] = pow(pk3[c], akap);
}

int launch_edge_pe(const Geo& g, real* pe, const real* delp, double ptop, hipStream_t st) {
  const int cells = 4 * (g.n + 1);
  hipLaunchKernelGGL(k_edge_pe, dim3((cells + 63) / 64), dim3(64), 0, st, g, pe, delp, ptop);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}
int launch_pk3_halo(const Geo& g, real* pk3, const real* delp, double ptop, double akap, hipStream_t st) {
  const int cells = (g.n + 4) * (g.n + 4) - g.n * g.n;
  hipLaunchKernelGGL(k_pk3_halo_scan, dim3((cells + 63) / 64), dim3(64), 0, st, g, pk3, delp, ptop);
  hipLaunchKernelGGL(k_pk3_halo_pow, dim3((cells + 63) / 64, g.nk), dim3(64), 0, st, g, pk3, akap);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}

// ------------------------------------------------------------------------------------------------
// RayleighDamping (ray_fast.py:48-141).  Everything that depends on the level only (rf, p_ref, which level's
// running momentum sum a level finally sees) is evaluated on the host, literally following the stencil's
// FORWARD/BACKWARD passes; the kernel does the per-column sums.  Only the top `kmax` levels are touched.
// ------------------------------------------------------------------------------------------------
#define RAY_MAXK 48
struct RayK {
  double rf[RAY_MAXK], dp[RAY_MAXK], p_ref[RAY_MAXK];
  signed char act[RAY_MAXK], nudge[RAY_MAXK], msrc[RAY_MAXK];
  int kmax;
};

__device__ __forceinline__ void ray_wind(const RayK& r, real* __restrict__ wind, long c0, long sk) {
  double s_[RAY_MAXK];
  double s = 0.0;
  for (int k = 0; k < r.kmax; ++k) {
    if (r.act[k]) {
      const double wv = wind[c0 + (long)k * sk];
      const double layer = (1.0 - r.rf[k]) * r.dp[k] * wv;
      s = (k == 0) ? layer : s + layer;
      wind[c0 + (long)k * sk] = wv * r.rf[k];
    }
    s_[k] = s;
  }
  for (int k = 0; k < r.kmax; ++k) {
    if (r.nudge[k]) wind[c0 + (long)k * sk] = wind[c0 + (long)k * sk] + s_[r.msrc[k]] / r.p_ref[k];
  }
}

// The sponge is a few levels deep (kmax ~ 10 at 79 levels): with all of a column's levels loaded at once (RAY_REG of them, in
// registers) and the three fields in different workgroups (blockIdx.z), a column costs one load round trip per field instead
// of two per level and field in a row -- the kernel is a few hundred waves, its time is the latency of one of them.
#define RAY_REG 16
__device__ __forceinline__ void ray_wind_regs(const RayK& r, real* __restrict__ wind, long c0, long sk) {
  double wv[RAY_REG], s_[RAY_REG];
#pragma unroll
  for (int k = 0; k < RAY_REG; ++k)
    if (k < r.kmax && (r.act[k] || r.nudge[k])) wv[k] = wind[c0 + (long)k * sk];
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < RAY_REG; ++k) {
    if (k < r.kmax && r.act[k]) {
      const double layer = (1.0 - r.rf[k]) * r.dp[k] * wv[k];
      s = (k == 0) ? layer : s + layer;
      wv[k] = wv[k] * r.rf[k];
    }
    s_[k] = s;
  }
#pragma unroll
  for (int k = 0; k < RAY_REG; ++k) {
    if (k < r.kmax && (r.act[k] || r.nudge[k])) {
      double out = wv[k];
      if (r.nudge[k]) {
        double src = 0.0;
#pragma unroll
        for (int q = 0; q < RAY_REG; ++q)
          if (r.msrc[k] == q) src = s_[q];  // (block-uniform: a scalar branch per candidate)
        out = out + src / r.p_ref[k];
      }
      wind[c0 + (long)k * sk] = out;
    }
  }
}

__global__ void __launch_bounds__(64)
k_ray_fast(Geo g, RayK r, real* __restrict__ u, real* __restrict__ v, real* __restrict__ w, int hydrostatic) {
  const int i = g.is + blockIdx.x * 64 + threadIdx.x;
  const int j = g.js + blockIdx.y;
  if (i > g.ie + 1 || j > g.je + 1) return;
  const long c0 = IDX2(g, i, j);
  const int field = (int)blockIdx.z;  // 0: u, 1: v, 2: w
  const bool regs = r.kmax <= RAY_REG;
  if (field == 0 && i <= g.ie) {
    if (regs) ray_wind_regs(r, u, c0, g.sk);
    else ray_wind(r, u, c0, g.sk);
  }
  if (field == 1 && j <= g.je) {
    if (regs) ray_wind_regs(r, v, c0, g.sk);
    else ray_wind(r, v, c0, g.sk);
  }
  if (field == 2 && !hydrostatic && i <= g.ie && j <= g.je) {
    if (regs) {
      double wv[RAY_REG];
#pragma unroll
      for (int k = 0; k < RAY_REG; ++k)
        if (k < r.kmax && r.act[k]) wv[k] = w[c0 + (long)k * g.sk];
#pragma unroll
      for (int k = 0; k < RAY_REG; ++k)
        if (k < r.kmax && r.act[k]) w[c0 + (long)k * g.sk] = wv[k] * r.rf[k];
    } else {
      for (int k = 0; k < r.kmax; ++k)
        if (r.act[k]) w[c0 + (long)k * g.sk] = w[c0 + (long)k * g.sk] * r.rf[k];
    }
  }
}

int launch_ray_fast(const Geo& g, real* u, real* v, real* w, const double* dp, const double* pfull, double dt,
                    double ptop, double rf_cutoff, double tau, int hydrostatic, hipStream_t st) {
  const int nk = g.nk;
  const double SDAY = 86400.0, PI = 3.14159265358979323846;
  const double nudge = rf_cutoff + fmin(100.0, 10.0 * ptop);
  RayK r;
  memset(&r, 0, sizeof(r));
  int kmax = 0;
  for (int k = 0; k < nk; ++k)
    if (pfull[k] < rf_cutoff || pfull[k] < nudge) kmax = k + 1;
  if (kmax > RAY_MAXK) return PACE_ERR_UNSUPPORTED;
  if (kmax == 0) return PACE_OK;
  if (!(pfull[0] < rf_cutoff)) return PACE_ERR_UNSUPPORTED;  // the reference reads an undefined temporary then
  r.kmax = kmax;
  // p_ref: FORWARD accumulate over nudged levels, then BACKWARD copy (ray_fast.py:79-89) -- over all nk levels
  {
    double* pr = new double[nk];
    double run = 0.0;
    for (int k = 0; k < nk; ++k) {
      const bool nz = pfull[k] < nudge;
      if (k == 0) run = nz ? dp[0] : NAN;
      else if (nz) run = run + dp[k];
      pr[k] = run;
    }
    for (int k = nk - 2; k >= 0; --k)
      if (pfull[k] < nudge) pr[k] = pr[k + 1];
    for (int k = 0; k < kmax; ++k) r.p_ref[k] = pr[k];
    delete[] pr;
  }
  // msrc: after the BACKWARD pass "if active: dmdir = dmdir[k+1]" level k holds the forward sum of level msrc[k]
  {
    int* ms = new int[nk];
    ms[nk - 1] = nk - 1;
    for (int k = nk - 2; k >= 0; --k) ms[k] = (pfull[k] < rf_cutoff) ? ms[k + 1] : k;
    for (int k = 0; k < kmax; ++k) r.msrc[k] = (signed char)(ms[k] < kmax ? ms[k] : kmax - 1);
    delete[] ms;
  }
  for (int k = 0; k < kmax; ++k) {
    r.act[k] = pfull[k] < rf_cutoff;
    r.nudge[k] = pfull[k] < nudge;
    r.dp[k] = dp[k];
    if (r.act[k]) {
      // compute_rff_vals (ray_fast.py:24-40)
      const double s = sin(0.5 * PI * log(rf_cutoff / pfull[k]) / log(rf_cutoff / ptop));
      const double rffvals = dt / (tau * SDAY) * (s * s);
      r.rf[k] = 1.0 / (1.0 + rffvals);
    }
  }
  hipLaunchKernelGGL(k_ray_fast, dim3((g.n + 1 + 63) / 64, g.n + 1, 3), dim3(64), 0, st, g, r, u, v, w, hydrostatic);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}

// ------------------------------------------------------------------------------------------------
// HyperdiffusionDamping (del2cubed.py:16-194): one launch per iteration, ping-pong between qdel and a
// scratch field.  corner_fill is a function of the input; copy_corners_{x,y} are read-side maps.
// ------------------------------------------------------------------------------------------------
struct Del2 {
  const Geo& g;
  const real* q;  // level base applied
  // corner_fill (del2cubed.py:31-68): the inner corner cell and its two out-of-tile neighbours all become
  // the mean of the three
  __device__ __forceinline__ double filled(int i, int j) const {
    const int ci = (i <= g.is) ? g.is : g.ie, cj = (j <= g.js) ? g.js : g.je;
    const int xo = (ci == g.is) ? g.is - 1 : g.ie + 1, yo = (cj == g.js) ? g.js - 1 : g.je + 1;
    const bool hit = (i == ci && j == cj) || (i == xo && j == cj) || (i == ci && j == yo);
    if (hit && (i <= g.is || i >= g.ie) && (j <= g.js || j >= g.je))
      return (q[IDX2(g, ci, cj)] + q[IDX2(g, xo, cj)] + q[IDX2(g, ci, yo)]) * (1.0 / 3.0);
    return q[IDX2(g, i, j)];
  }
  __device__ __forceinline__ double qx(int i, int j, bool cc) const {
    if (cc) remap_agrid_x(g, i, j);
    return filled(i, j);
  }
  __device__ __forceinline__ double qy(int i, int j, bool cc) const {
    if (cc) remap_agrid_y(g, i, j);
    return filled(i, j);
  }
};

__global__ void __launch_bounds__(256)
k_del2cubed_iter(Geo g, Met m, const real* __restrict__ qin, real* __restrict__ qout, double cd, int nt) {
  PATCH_IJK(g);
  if (i > g.ni - 2 || j > g.nj - 2) return;
  const long kb = (long)k * g.sk;
  const long c2 = IDX2(g, i, j);
  Del2 d{g, qin + kb};
  double val = d.filled(i, j);
  if (i >= g.is - nt && i <= g.ie + nt && j >= g.js - nt && j <= g.je + nt) {
    const bool cc = nt > 0;
    const double fx0 = m.del6_v[c2] * (d.qx(i - 1, j, cc) - d.qx(i, j, cc));
    const double fx1 = m.del6_v[c2 + 1] * (d.qx(i, j, cc) - d.qx(i + 1, j, cc));
    const double fy0 = m.del6_u[c2] * (d.qy(i, j - 1, cc) - d.qy(i, j, cc));
    const double fy1 = m.del6_u[c2 + g.sj] * (d.qy(i, j, cc) - d.qy(i, j + 1, cc));
    val = val + cd * m.rarea[c2] * (fx0 - fx1 + fy0 - fy1);
  }
  qout[kb + c2] = val;
}

int64_t del2cubed_workspace_bytes(const Geo& g) { return (int64_t)g.sk * (g.nk + 1) * (int64_t)sizeof(real); }

int launch_del2cubed(const Geo& g, const Met& m, void* ws_, real* qdel, double cd, int nmax, hipStream_t st) {
  const int ntimes = nmax < 3 ? nmax : 3;
  real* scratch = (real*)ws_;
  real* src = qdel;
  real* dst = scratch;
  const dim3 grid = plane_grid(g, g.nk), block(256);
  for (int n = 0; n < ntimes; ++n) {
    hipLaunchKernelGGL(k_del2cubed_iter, patch_grid(g, g.nk), PATCH_BLOCK, 0, st, g, m, src, dst, cd, ntimes - (n + 1));
    real* t = src;
    src = dst;
    dst = t;
  }
  if (src != qdel) hipLaunchKernelGGL(k_scale_copy, pair_grid(g, (int)grid.y), block, 0, st, g, src, qdel, 1.0, 0, 0, g.ni - 2, 0, g.nj - 2);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}

// apply_diffusive_heating (temperature_adjust.py:8-43), compute domain, first nlev levels
__global__ void __launch_bounds__(256)
k_diffusive_heating(Geo g, const real* __restrict__ delp, const real* __restrict__ delz,
                    const real* __restrict__ cappa, const real* __restrict__ heat_source, real* __restrict__ pt,
                    double delt_time_factor) {
  PLANE_IJK(g);
  if (i < g.is || i > g.ie || j < g.js || j > g.je) return;
  const long c = IDX3(g, i, j, k);
  const double pkz = pow(RDG * delp[c] / delz[c] * pt[c], cappa[c] / (1.0 - cappa[c]));
  const double dtmp = heat_source[c] / (CV_AIR * delp[c]);
  const double fac = (k == 0) ? 0.1 : (k == 1) ? 0.5 : 1.0;
  const double lim = delt_time_factor * fac;
  const double mag = fmin(lim, fabs(dtmp));
  const double deltmin = (dtmp > 0.0) ? fabs(mag) : -fabs(mag);  // basic_operations.sign
  pt[c] = pt[c] + deltmin / pkz;
}

int launch_diffusive_heating(const Geo& g, const real* delp, const real* delz, const real* cappa,
                             const real* heat_source, real* pt, double delt_time_factor, int nlev, hipStream_t st) {
  if (nlev < 1) return PACE_OK;
  hipLaunchKernelGGL(k_diffusive_heating, plane_grid(g, nlev), dim3(256), 0, st, g, delp, delz, cappa, heat_source, pt,
                     delt_time_factor);
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}
