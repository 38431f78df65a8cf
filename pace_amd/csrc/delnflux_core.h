// Tile-local core of DelnFluxNoSG / DelnFlux (delnflux.py:1209-1261): shared by the stand-alone kernel
// (k_delnflux.hip) and the fused transport kernel (k_fvtp2d.hip).  See k_delnflux.hip for the design notes.
#pragma once
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
  const int tid = threadIdx.x & 255;  // (256 threads per tile; k_fvtp2d_pair has two tiles per workgroup)
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
