// Development probe: how many bits below the largest product does v_mfma_f32_32x32x16_f16 keep
// when it sums the 16 products of one instruction? One product is 1.0, another is 2^-j (placed in
// the same half-wave or in the other one); C = 0. Prints the computed sum minus 1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, int j, int other_half, int via_c) {
    const int l = threadIdx.x, h = l >> 5;
    h8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 c = {0};
    const float small = ldexpf(1.f, -j);
    // small = 2^-8 (f16 normal) times 2^-(j-8) split over A and B to stay in f16 range
    const _Float16 sa = (_Float16)ldexpf(1.f, -(j / 2)), sb = (_Float16)ldexpf(1.f, -(j - j / 2));
    if (h == 0) { a[0] = (_Float16)1.f; b[0] = (_Float16)1.f; }
    if (via_c) { for (int r = 0; r < 16; ++r) c[r] = 1.f; if (h == 0) { a[0] = 0; } }
    if (h == (other_half ? 1 : 0)) { a[1] = sa; b[1] = sb; }
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (l == 0) out[0] = c[0] - 1.f - 0.f * small;
}
int main() {
    float* d; hipMalloc(&d, 4);
    for (int mode = 0; mode < 3; ++mode) {
        printf("%s:\n", mode == 0 ? "big and small product in the same half-wave" : mode == 1 ? "in different half-waves" : "big value in C, small product");
        for (int j = 8; j <= 26; j += 2) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, j, mode == 1, mode == 2);
            float v; hipMemcpy(&v, d, 4, hipMemcpyDeviceToHost);
            printf("  small = 2^-%-2d  result-1 = %.3e (exact %.3e)\n", j, v, ldexp(1.0, -j));
        }
    }
    return 0;
}
