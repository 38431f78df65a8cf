// Halo-exchange pack / unpack (replaces HaloDataTransformerGPU's pack/unpack kernels and the rotate + slice logic
// in front of them: util/pace/util/halo_data_transformer.py:150-921, rotate.py:4-50, _boundary_utils.py:58-95).
//
// A message strip is addressed in the RECEIVER's orientation: element (a, b, k) lands at field index
// (ri0 + a, rj0 + b, k) on the receiving tile.  The sender reads it from (i0 + a*di_a + b*di_b,
// j0 + a*dj_a + b*dj_b, k) -- the tile-to-tile rotation, the component swap of vector fields and their sign are all
// folded into that affine map and `sign` by the host (pace_amd/util/halo.py), so packing IS the rotation and the
// receiver only copies.  One launch moves every strip of every field of an updater (blockIdx.y = strip).
// Message layout [k][b][a]: rows of `a` (the receiver's i) are contiguous.  HBM/latency-bound, tiny.
#include "common.h"
#include "kernels.h"

#define HALO_MAX_DESC 16
struct HaloBatch {
  pace_halo_desc_t d[HALO_MAX_DESC];
};

template <int UNPACK>
__global__ void __launch_bounds__(256) k_halo_copy(Geo g, HaloBatch batch) {
  const pace_halo_desc_t& d = batch.d[blockIdx.y];
  const long total = (long)d.na * d.nb * d.nk;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int a = (int)(e % d.na);
    const long r = e / d.na;
    const int b = (int)(r % d.nb);
    const int k = (int)(r / d.nb);
    const long f = IDX3(g, d.i0 + a * d.di_a + b * d.di_b, d.j0 + a * d.dj_a + b * d.dj_b, k);
    if (UNPACK) d.field[f] = d.buf[e];
    else d.buf[e] = d.sign * d.field[f];
  }
}

int launch_halo_copy(const Geo& g, const pace_halo_desc_t* descs, int ndesc, int unpack, hipStream_t st) {
  for (int base = 0; base < ndesc; base += HALO_MAX_DESC) {
    const int n = (ndesc - base < HALO_MAX_DESC) ? ndesc - base : HALO_MAX_DESC;
    HaloBatch batch;
    long most = 1;
    for (int t = 0; t < n; ++t) {
      batch.d[t] = descs[base + t];
      const long tot = (long)descs[base + t].na * descs[base + t].nb * descs[base + t].nk;
      if (tot > most) most = tot;
    }
    for (int t = n; t < HALO_MAX_DESC; ++t) batch.d[t] = batch.d[0];
    long bx = (most + 255) / 256;
    if (bx > 1024) bx = 1024;
    const dim3 grid((unsigned)bx, (unsigned)n), block(256);
    if (unpack) hipLaunchKernelGGL(k_halo_copy<1>, grid, block, 0, st, g, batch);
    else hipLaunchKernelGGL(k_halo_copy<0>, grid, block, 0, st, g, batch);
  }
  PACE_CHECK_LAUNCH();
  return PACE_OK;
}
