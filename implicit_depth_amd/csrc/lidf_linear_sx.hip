// lidf_linear_sx.hip — the instantiations of lidf_linear_kernel (lidf_linear_kernel.inc) with SPLIT and XCOL:
// a translation unit per family for the build's wall clock (eight instantiations compile for ~45 s).
#include "lidf_device.h"
#include "lidf_linear_kernel.inc"

extern "C" void lidf_launch_linear_sx(int nt, dim3 g, dim3 b, hipStream_t st, const LinearArgs& a) {
    launch_linear_nt<true, true>(nt, g, b, st, a);
}
