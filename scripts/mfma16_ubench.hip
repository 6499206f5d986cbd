// Development microbenchmark (round 4): sustained rate of v_mfma_f32_16x16x4_f32 against
// v_mfma_f32_32x32x2_f32 (1 wavefront per SIMD, 256 workgroups), independent accumulators and dependent
// chains. Prints TFLOP/s of each.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define M32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define M16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int V>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    float b = lane * 0.001f, a0 = 1.0f + lane * 0.01f;
    f32x16 c32[4];
    f32x4 c16[8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) c32[i][j] = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) c16[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 64; ++s) {
            if (V == 0) c32[s & 3] = M32(a0, b, c32[s & 3]);       // 32x32x2, 4 independent
            else if (V == 1) c32[0] = M32(a0, b, c32[0]);          // 32x32x2, dependent
            else if (V == 2) c16[s & 7] = M16(a0, b, c16[s & 7]);  // 16x16x4, 8 independent
            else if (V == 3) c16[s & 1] = M16(a0, b, c16[s & 1]);  // 16x16x4, 2 chains
            else c16[0] = M16(a0, b, c16[0]);                      // 16x16x4, dependent
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += c32[i][j];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += c16[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V>
void run(const char* name, float* out, double flop) {
    const int iters = 4000, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %.3f ms  %.1f TFLOP/s\n", name, ms, (double)grid * 4 * iters * 64 * flop / (ms * 1e-3) / 1e12);
}

int main() {
    float* out; hipMalloc(&out, 1024 * 256 * 4);
    run<0>("32x32x2 f32, 4 independent", out, 4096);
    run<1>("32x32x2 f32, dependent chain", out, 4096);
    run<2>("16x16x4 f32, 8 independent", out, 2048);
    run<3>("16x16x4 f32, 2 chains", out, 2048);
    run<4>("16x16x4 f32, dependent chain", out, 2048);
    return 0;
}
