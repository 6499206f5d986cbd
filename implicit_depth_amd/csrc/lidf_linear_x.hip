// lidf_linear_x.hip — the instantiations of lidf_linear_kernel (lidf_linear_kernel.inc) with one more output column through the vector unit (XCOL):
// a translation unit per family for the build's wall clock (eight instantiations compile for ~45 s).
#include "lidf_device.h"
#include "lidf_linear_kernel.inc"

extern "C" void lidf_launch_linear_x(int nt, dim3 g, dim3 b, hipStream_t st, const LinearArgs& a) {
    launch_linear_nt<false, true>(nt, g, b, st, a);
}
