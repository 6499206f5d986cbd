// Development microbenchmark: cost of individual filler instructions beside
// v_mfma_f32_32x32x16_f16 with one wavefront per SIMD (cycles per MFMA for N fillers per gap).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

template <int KIND, int N>
__global__ void __launch_bounds__(256) k(float* out, long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.01f + i); b[i] = (_Float16)(0.5f + i * 0.1f); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = lane * 0.5f + i;
    float w = 0.25f * lane;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            acc[s % 4] = MFMA(a, b, acc[s % 4]);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                float x = v[j % 8];
                if (KIND == 0) asm volatile("v_mul_f32 %0, 0x3f7d70a4, %0" : "+v"(x));
                if (KIND == 1) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(x) : "v"(w));
                if (KIND == 2) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x) : "v"(w));
                if (KIND == 3) asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(x));
                if (KIND == 4) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(w));
                if (KIND == 5) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(x) : "v"(w));
                if (KIND == 6) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(w));
                if (KIND == 7) asm volatile("v_accvgpr_write_b32 a1, %0" :: "v"(x));
                if (KIND == 8) asm volatile("v_sin_f32 %0, %1" : "=v"(x) : "v"(w));
                if (KIND == 9) asm volatile("v_fract_f32 %0, %1" : "=v"(x) : "v"(w));
                if (KIND == 10) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(w));
                v[j % 8] = x;
            }
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

template <int KIND, int N>
float run(float* out, long long* cyc) {
    int iters = 300;
    hipLaunchKernelGGL((k<KIND, N>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((k<KIND, N>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    return c / (iters * 32.0);
}
template <int KIND>
void row(const char* name, float* out, long long* cyc) {
    printf("%-22s N=0 %.1f  N=2 %.1f  N=4 %.1f  N=5 %.1f  N=6 %.1f  N=8 %.1f\n", name, run<KIND, 0>(out, cyc),
           run<KIND, 2>(out, cyc), run<KIND, 4>(out, cyc), run<KIND, 5>(out, cyc), run<KIND, 6>(out, cyc), run<KIND, 8>(out, cyc));
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    row<0>("v_mul_f32", out, cyc); row<1>("v_fma_mix_f32", out, cyc); row<2>("v_cvt_pk_f16_f32", out, cyc);
    row<3>("v_accvgpr_read", out, cyc); row<4>("v_max_f32", out, cyc); row<5>("v_cvt_f32_f16", out, cyc);
    row<6>("v_sub_f32", out, cyc); row<7>("v_accvgpr_write", out, cyc); row<8>("v_sin_f32", out, cyc);
    row<9>("v_fract_f32", out, cyc); row<10>("v_fma_f32", out, cyc);
    return 0;
}
