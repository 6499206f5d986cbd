// Development test: accuracy of v_sin_f32 / v_cos_f32 fed with an exact "revolutions" range
// reduction, against double-precision sin/cos, for positional-encoding arguments x * 2^o.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const float* x, float* s, float* c, int n, int L) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float C_HI = 0.15915494f;             // fl(1/(2pi))
    const float C_LO = 1.0f / 6.283185307179586f - 0.15915494f > 0 ? 0.f : 0.f;  // placeholder
    (void)C_LO;
    const float xv = x[i];
    // y = x/(2pi) as hi + lo:  hi = fl(x*C_HI), lo = fma(x, C_HI, -hi) + x*C_LO2
    const float C_LO2 = 6.4206383e-09f;          // 1/(2pi) - C_HI
    const float hi = xv * C_HI;
    const float lo = fmaf(xv, C_HI, -hi) + xv * C_LO2;
    float sc = 1.f;
    for (int o = 0; o < L; ++o) {
        const float t = __builtin_amdgcn_fractf(hi * sc) + lo * sc;   // exact scaling by 2^o
        s[i * L + o] = __builtin_amdgcn_sinf(t);
        c[i * L + o] = __builtin_amdgcn_cosf(t);
        sc *= 2.f;
    }
}
int main() {
    const int n = 1 << 20, L = 10;
    std::vector<float> hx(n);
    std::mt19937 g(1);
    std::uniform_real_distribution<float> u(-2.5f, 2.5f);
    for (int i = 0; i < n; ++i) hx[i] = u(g);
    hx[0] = 0.f; hx[1] = 1e-7f; hx[2] = 2.3f; hx[3] = -2.3f; hx[4] = 3.14159265f; hx[5] = 1.57079633f;
    float *dx, *ds, *dc;
    hipMalloc(&dx, n * 4); hipMalloc(&ds, n * L * 4); hipMalloc(&dc, n * L * 4);
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, ds, dc, n, L);
    std::vector<float> hs(n * L), hc(n * L);
    hipMemcpy(hs.data(), ds, n * L * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hc.data(), dc, n * L * 4, hipMemcpyDeviceToHost);
    for (int o = 0; o < L; ++o) {
        double es = 0, ec = 0;
        for (int i = 0; i < n; ++i) {
            double a = (double)hx[i] * (double)(1 << o);
            es = fmax(es, fabs(hs[i * L + o] - sin(a)));
            ec = fmax(ec, fabs(hc[i * L + o] - cos(a)));
        }
        printf("octave %d: max abs err sin %.3e cos %.3e\n", o, es, ec);
    }
    return 0;
}
