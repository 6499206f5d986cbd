// Development microbenchmark: sustained issue rate of v_mfma_f32_32x32x2_f32 in the patterns the
// decoder kernel uses (1 wavefront per SIMD). Prints cycles per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

template <int VARIANT>
__global__ void __launch_bounds__(256) k(const float* w, float* out, long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 1 << 20, 0x00020000);
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float b = lane * 0.001f, a0 = 1.0f + lane * 0.01f;
    f32x4 ring[8];
    for (int i = 0; i < 8; ++i) ring[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, i * 1024, 0));
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 64; ++s) {
            if (VARIANT == 0) {  // 4 independent accumulators, register operands
                acc[s & 3] = MFMA(a0, b, acc[s & 3]);
            } else if (VARIANT == 1) {  // one dependent chain
                acc[0] = MFMA(a0, b, acc[0]);
            } else {  // dependent chain fed by the 8-deep buffer-load ring, 4 MFMAs per quad
                if ((s & 3) == 0) {
                    const int q = s / 4;
                    f32x4 v = ring[q % 8];
                    ring[q % 8] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, ((it * 16 + q + 8) % 512) * 1024, 0));
                    acc[0] = MFMA(v[0], b, acc[0]);
                    acc[0] = MFMA(v[1], b, acc[0]);
                    acc[0] = MFMA(v[2], b, acc[0]);
                    acc[0] = MFMA(v[3], b, acc[0]);
                    FENCE();
                }
            }
        }
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
void run(const char* name, const float* w, float* out, long long* cyc, int grid) {
    int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, w, out, cyc, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, w, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double nm = (double)iters * (V == 2 ? 64 : 64);
    double tf = (double)grid * 4 * nm * 4096 / (ms * 1e-3) / 1e12;
    printf("%-28s grid %d: %.3f ms  %.1f TFLOP/s  clock64 ticks/MFMA %.2f  (ms-derived ns/MFMA %.2f)\n", name, grid, ms, tf, c / nm, ms * 1e6 / nm);
}

int main() {
    float *w, *out; long long* cyc;
    hipMalloc(&w, 1 << 20); hipMemset(w, 0, 1 << 20);
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    for (int grid : {256, 512}) {
        run<0>("4 independent accumulators", w, out, cyc, grid);
        run<1>("1 dependent chain", w, out, cyc, grid);
        run<2>("dependent chain + ring loads", w, out, cyc, grid);
    }
    return 0;
}
