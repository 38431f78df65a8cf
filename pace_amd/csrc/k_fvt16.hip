// The lean transport / scalar-phase kernels of k_fvt.hip for a second tile shape, 16 x 24: C48 (BASELINE configuration 2) is a
// multiple of 16 and of 24 but not of 32.  Builds that set the tile shape themselves (the CPU emulation's small / canonical
// tilings, tests/emu) have one shape only: the functions of this file then answer "not covered".
#ifndef FV_TI
#define FV_TI 16
#define FV_TJ 24
#define FVT_NS fvt16
#define FVT_SHAPE 16
#include "k_fvt.hip"
#else
#include "common.h"
#include "kernels.h"
bool fvt16_covers(const Geo&, int) { return false; }
bool fvt16_take_winds() { return false; }
int fvt16_launch_transport(const Geo&, const Met&, const real*, const real*, const real*, const real*, const real*, real*, real*, const real*,
                           const real*, int, int, int, int, const FvDamp&, hipStream_t) {
  return PACE_ERR_UNSUPPORTED;
}
int fvt16_launch_scalars(const Geo&, const Met&, const real*, const real*, const real*, const real*, real* const*, const real*, const real*,
                         const real*, const real*, real*, real*, real*, real*, real*, const real*, int, int, int, int, double, hipStream_t,
                         const DswWinds*) {
  return PACE_ERR_UNSUPPORTED;
}
#endif
