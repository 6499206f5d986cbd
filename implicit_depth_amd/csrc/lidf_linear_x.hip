// lidf_linear_x.hip — the instantiations of lidf_linear_kernel (lidf_linear_kernel.inc) whose operand rows come from two
// buffers (SPLIT: the decoder pair's joint input gradient, K = 512) and / or that carry one more output column
// through the vector unit (XCOL: 385 = 12 x 32 + 1). A translation unit of its own for the build's wall clock.
#include "lidf_device.h"
#include "lidf_linear_kernel.inc"

extern "C" void lidf_launch_linear_x(int nt, int split, int xcol, dim3 g, dim3 b, hipStream_t st, const LinearArgs& a) {
    if (split) {
        if (xcol) launch_linear_nt<true, true>(nt, g, b, st, a);
        else launch_linear_nt<true, false>(nt, g, b, st, a);
    } else {
        launch_linear_nt<false, true>(nt, g, b, st, a);
    }
}
