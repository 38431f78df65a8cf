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

#include "delnflux_core.h"

// MODE 0: write fx2, fy2.  MODE 1: fx += fx2, fy += fy2.  MODE 2: mass-weighted add (delnflux.py:318-328).
template <int MODE>
__global__ void __launch_bounds__(256) k_delnflux(Geo g, Met m, const real* __restrict__ q, real* fxo,
                                                  real* fyo, const real* __restrict__ mass,
                                                  const real* __restrict__ damp_k,
                                                  const real* __restrict__ nord_k, int nmax, int mass_given) {
  __shared__ double sraw[DH * DWP];
  __shared__ double sd[DH * DWP];
  __shared__ double sfx[DH * DWP];
  __shared__ double sfy[DH * DWP];
  const int tid = threadIdx.x;
  const int i0 = g.is + blockIdx.x * DN_TI;
  const int j0 = g.js + blockIdx.y * DN_TJ;
  const int k = blockIdx.z;
  const long kb = (long)k * g.sk;
  const double damp = damp_k[k];
  for (int e = tid; e < DW * DH; e += 256) {
    const int jj = e / DW, ii = e - jj * DW;
    const int gi = i0 - 3 + ii, gj = j0 - 3 + jj;
    sraw[jj * DWP + ii] = (gi >= 0 && gi < g.ni && gj >= 0 && gj < g.nj) ? q[kb + IDX2(g, gi, gj)] : 0.0;
  }
  __syncthreads();
  delnflux_core(g, m, sraw, sd, sfx, sfy, i0, j0, mass_given ? 1.0 : damp, nord_k[k] > 0.0, nmax);

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

int launch_delnflux(const Geo& g, const Met& m, int mode, const real* q, real* fx, real* fy,
                    const real* mass, const real* damp_k, const real* nord_k, int nmax, int mass_given,
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
