// DelnFluxNoSG / DelnFlux (Fortran deln_flux, del6_vt_flux): conservative del-(2n+2) damping fluxes.
// Reference: fv3core/pace/fv3core/stencils/delnflux.py:945-1261 -- 5 + 5*nmax stencil launches over
// full 3-D fields (d2, fx2, fy2) on shrinking domains, with corner copies between them.
// Here: ONE kernel.  A workgroup stages damp*q on its (TI+6) x (TJ+6) footprint in LDS and iterates
// flux -> divergence -> flux entirely in LDS; the validity of the LDS values shrinks by one ring per
// sweep, which is exactly the 3-cell footprint staged.  Corner copies (copy_corners_{x,y}_nord) are
// index remaps on the LDS reads (oracle/corner_ops.py); the dependency cone of a valid output never
// leaves the staged footprint ("delnflux cone").
// HBM-bound: algorithmic traffic = 1 read (+1 mass, +2 flux reads) + 2 writes of 3-D fields.
#include "common.h"
#include "kernels.h"

#ifndef DN_TI
#define DN_TI 32
#define DN_TJ 16
#endif
#define DW (DN_TI + 6)
#define DH (DN_TJ + 6)

// MODE 0: write fx2, fy2.  MODE 1: fx += fx2, fy += fy2.  MODE 2: mass-weighted add (delnflux.py:318-328).
//
// Each thread owns NE fixed points of the footprint for the whole kernel: their LDS slot, validity flags and the three
// metric values (del6_v, del6_u, rarea) are worked out once, so an iteration is LDS reads and a handful of flops.
#define DWP (DW + 1)
#define NE ((DW * DH + 255) / 256)
template <int MODE>
__global__ void __launch_bounds__(256) k_delnflux(Geo g, Met m, const double* __restrict__ q, double* fxo,
                                                  double* fyo, const double* __restrict__ mass,
                                                  const double* __restrict__ damp_k,
                                                  const double* __restrict__ nord_k, int nmax, int mass_given) {
  __shared__ double sd[DH * DWP];
  __shared__ double sfx[DH * DWP];
  __shared__ double sfy[DH * DWP];
  const int tid = threadIdx.x;
  const int i0 = g.is + blockIdx.x * DN_TI;
  const int j0 = g.js + blockIdx.y * DN_TJ;
  const int k = blockIdx.z;
  const long kb = (long)k * g.sk;
  const int ilo = i0 - 3, jlo = j0 - 3;
  const bool hi_order = nord_k[k] > 0.0;
  // the corner-copy index maps only matter to workgroups whose footprint reaches a corner of the halo (block-uniform)
  const bool rc = hi_order && (ilo < g.is || ilo + DW - 1 > g.ie) && (jlo < g.js || jlo + DH - 1 > g.je);
  const int iters = hi_order ? nmax : 0;
  const double damp = damp_k[k];
  const double d0 = mass_given ? 1.0 : damp;

  int lidx[NE], pgi[NE], pgj[NE];
  bool own[NE], flx[NE], cel[NE];
  double dv[NE], du[NE], ra[NE];
#pragma unroll
  for (int t = 0; t < NE; ++t) {
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
    const long c2 = IDX2(g, gi, gj);
    dv[t] = flx[t] ? m.del6_v[c2] : 0.0;
    du[t] = flx[t] ? m.del6_u[c2] : 0.0;
    ra[t] = cel[t] ? m.rarea[c2] : 0.0;
    if (own[t]) {
      double v = 0.0;
      if (stored) {
        v = q[kb + c2];
        if (!mass_given) v = d0 * v;
      }
      sd[lidx[t]] = v;
    }
  }
  __syncthreads();

  for (int it = 0;; ++it) {
#pragma unroll
    for (int t = 0; t < NE; ++t) {
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
    for (int t = 0; t < NE; ++t) {
      if (!own[t]) continue;
      const int l = lidx[t];
      double v = 0.0;
      if (cel[t]) v = (sfx[l] - sfx[l + 1] + sfy[l] - sfy[l + DWP]) * ra[t];
      sd[l] = v;
    }
    __syncthreads();
  }

  // ceil(N / TI) x ceil(N / TJ) workgroups: the faces ie+1 / je+1 are written by the workgroup owning cell ie / je
  for (int e = tid; e < (DN_TI + 1) * (DN_TJ + 1); e += 256) {
    const int jj = e / (DN_TI + 1), ii = e - jj * (DN_TI + 1);
    const int gi = i0 + ii, gj = j0 + jj;
    if (gi > g.ie + 1 || gj > g.je + 1) continue;
    if ((ii == DN_TI && gi != g.ie + 1) || (jj == DN_TJ && gj != g.je + 1)) continue;
    const long c = kb + IDX2(g, gi, gj);
    const double vx = sfx[(jj + 3) * DWP + ii + 3], vy = sfy[(jj + 3) * DWP + ii + 3];
    if (gj <= g.je && jj < DN_TJ) {
      if (MODE == 0) fxo[c] = vx;
      else if (MODE == 1) fxo[c] = fxo[c] + vx;
      else fxo[c] = fxo[c] + 0.5 * damp * (mass[c - 1] + mass[c]) * vx;
    }
    if (gi <= g.ie && ii < DN_TI) {
      if (MODE == 0) fyo[c] = vy;
      else if (MODE == 1) fyo[c] = fyo[c] + vy;
      else fyo[c] = fyo[c] + 0.5 * damp * (mass[c - g.sj] + mass[c]) * vy;
    }
  }
}

int launch_delnflux(const Geo& g, const Met& m, int mode, const double* q, double* fx, double* fy,
                    const double* mass, const double* damp_k, const double* nord_k, int nmax, int mass_given,
                    int nlev, hipStream_t st) {
  if (nmax > 2) return PACE_ERR_UNSUPPORTED;  // 3-cell halo (the reference would index out of range too)
  const dim3 grid((g.n + DN_TI - 1) / DN_TI, (g.n + DN_TJ - 1) / DN_TJ, nlev), block(256);
  if (mode == 0) {
    hipLaunchKernelGGL(k_delnflux<0>, grid, block, 0, st, g, m, q, fx, fy, mass, damp_k, nord_k, nmax, mass_given);
  } else if (mode == 1) {
    hipLaunchKernelGGL(k_delnflux<1>, grid, block, 0, st, g, m, q, fx, fy, mass, damp_k, nord_k, nmax, mass_given);
  } else {
    hipLaunchKernelGGL(k_delnflux<2>, grid, block, 0, st, g, m, q, fx, fy, mass, damp_k, nord_k, nmax, mass_given);
  }
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}
