// Development microbenchmark: sustained v_mfma_f32_32x32x16_f16 rate of the whole chip (wall clock
// and shader-clock ticks) for zero / smooth / random operands, 1 and 2 wavefronts per SIMD. Shows
// how far the power budget lets the clock stay up under dense f16 matrix work.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

__global__ void __launch_bounds__(256, 2) k(const _Float16* src, float* out, long long* cyc, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    h8 a[4], b[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 8; ++i) { a[j][i] = src[(tid * 64 + j * 8 + i) & 0xfffff]; b[j][i] = src[(tid * 64 + 32 + j * 8 + i) & 0xfffff]; }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 32; ++s) acc[s % 4] = MFMA(a[s % 4], b[(s / 4) % 4], acc[s % 4]);
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[tid] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int N = 1 << 20;
    _Float16* h = (_Float16*)malloc(N * 2);
    _Float16* d; float* out; long long* cyc;
    hipMalloc(&d, N * 2); hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"zero", "smooth", "random"};
    for (int kind = 0; kind < 3; ++kind) {
        for (int i = 0; i < N; ++i)
            h[i] = kind == 0 ? (_Float16)0.f : kind == 1 ? (_Float16)(0.001f * (i % 1000)) : (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
        hipMemcpy(d, h, N * 2, hipMemcpyHostToDevice);
        for (int wgs = 256; wgs <= 512; wgs += 256) {
            const int iters = 20000;
            hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, d, out, cyc, 2000);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, d, out, cyc, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double flop = (double)wgs * 4 * iters * 32 * 32768.0;
            printf("%-7s %d waves/SIMD: %.2f ms  %.0f TFLOP/s  ticks/MFMA/wave %.1f  clock %.2f GHz\n", names[kind], wgs / 256, ms,
                   flop / ms / 1e9, c / (iters * 32.0), c / ms / 1e6);
        }
    }
    return 0;
}
