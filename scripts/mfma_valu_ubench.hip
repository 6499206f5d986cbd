// Development microbenchmark: how many independent VALU instructions hide behind one
// v_mfma_f32_32x32x2_f32 (1 wavefront per SIMD), for a dependent accumulator chain vs 4 rotating
// accumulators. Prints shader cycles per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

template <int NACC, int NVALU>
__global__ void __launch_bounds__(256) k(float* out, long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float b = lane * 0.001f, a0 = 1.0f + lane * 0.01f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = lane * 0.5f + i;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            acc[s % NACC] = MFMA(a0, b, acc[s % NACC]);
#pragma unroll
            for (int j = 0; j < NVALU; ++j) v[j % 8] = fmaxf(v[j % 8], v[j % 8] * 0.99f + 0.001f);
            FENCE();
        }
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int NVALU>
void run(float* out, long long* cyc) {
    int iters = 1000;
    hipLaunchKernelGGL((k<NACC, NVALU>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((k<NACC, NVALU>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("acc=%d  VALU ops per MFMA=%2d (x2 instr: mul-add+max)  cycles/MFMA %.1f\n", NACC, NVALU, c / (iters * 32.0));
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    run<1, 0>(out, cyc); run<1, 2>(out, cyc); run<1, 4>(out, cyc); run<1, 6>(out, cyc); run<1, 8>(out, cyc); run<1, 12>(out, cyc); run<1, 16>(out, cyc);
    run<4, 0>(out, cyc); run<4, 2>(out, cyc); run<4, 4>(out, cyc); run<4, 6>(out, cyc); run<4, 8>(out, cyc); run<4, 12>(out, cyc); run<4, 16>(out, cyc);
    return 0;
}
