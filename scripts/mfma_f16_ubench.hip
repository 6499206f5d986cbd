// Development microbenchmark for the split-f16 path (gfx950):
//  1. operand layout of v_mfma_f32_32x32x16_f16 against a host loop (asymmetric A and B)
//  2. whether f16 subnormal A/B inputs are honoured
//  3. how many VALU / LDS-read fillers hide behind one MFMA with one wavefront per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

__global__ void layout(const _Float16* A, const _Float16* B, float* D) {  // A[32][16], B[16][32]
    const int l = threadIdx.x, rc = l & 31, h = l >> 5;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = A[rc * 16 + 8 * h + i]; b[i] = B[(8 * h + i) * 32 + rc]; }
    f32x16 c = {0};
    c = MFMA(a, b, c);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + rc] = c[r];
}

template <int NVALU, int NLDS, int NACC = 4>
__global__ void __launch_bounds__(256) k(float* out, long long* cyc, int iters) {
    __shared__ f32x4 lds[1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = f32x4{1.f * i, 0.f, 1.f, 2.f};
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.01f + i); b[i] = (_Float16)(0.5f + i * 0.1f); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = lane * 0.5f + i;
    f32x4 q[4] = {};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            acc[s % NACC] = MFMA(a, b, acc[s % NACC]);
#pragma unroll
            for (int j = 0; j < NVALU; ++j) {
                float x = v[j % 8];
                asm volatile("v_mul_f32 %0, 0x3f7d70a4, %0\n" : "+v"(x));
                v[j % 8] = x;
            }
#pragma unroll
            for (int j = 0; j < NLDS; ++j) q[(s + j) % 4] += lds[(lane + 64 * ((s * NLDS + j) % 16))];
            FENCE();
        }
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += q[i][0] + q[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NVALU, int NLDS, int NACC = 4>
void run(float* out, long long* cyc) {
    int iters = 500;
    hipLaunchKernelGGL((k<NVALU, NLDS, NACC>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((k<NVALU, NLDS, NACC>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("acc=%d fillers per MFMA: VALU %2d  ds_read_b128 %d (+4 v_add each)  -> cycles/MFMA %.1f\n", NACC, NVALU, NLDS, c / (iters * 32.0));
}

int main() {
    std::vector<_Float16> A(512), B(512);
    for (int i = 0; i < 512; ++i) { A[i] = (_Float16)(((i * 37) % 101) * 0.01f - 0.5f); B[i] = (_Float16)(((i * 53) % 89) * 0.02f - 0.7f); }
    // subnormal probe: A[0][0] = 2^-20 (f16 subnormal), B[0][0] = 1024, every other k of row 0 / col 0 zero
    std::vector<_Float16> A2(512, (_Float16)0.f), B2(512, (_Float16)0.f);
    A2[0] = (_Float16)9.5367431640625e-07f; B2[0] = (_Float16)1024.f;
    _Float16 *dA, *dB; float* dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
    std::vector<float> D(1024);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
        double s = 0; for (int kk = 0; kk < 16; ++kk) s += (double)(float)A[r * 16 + kk] * (double)(float)B[kk * 32 + c];
        worst = fmax(worst, fabs(s - D[r * 32 + c]));
    }
    printf("layout check: max |D - ref| = %.3g (expect ~1e-6)\n", worst);
    hipMemcpy(dA, A2.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B2.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    printf("subnormal probe: 2^-20 * 1024 = %.6g (expect 0.000976562; 0 means flushed)\n", D[0]);

    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    run<0, 0>(out, cyc); run<2, 0>(out, cyc); run<4, 0>(out, cyc); run<5, 0>(out, cyc); run<6, 0>(out, cyc);
    run<8, 0>(out, cyc); run<12, 0>(out, cyc);
    run<0, 0, 1>(out, cyc); run<4, 0, 1>(out, cyc); run<0, 0, 2>(out, cyc); run<4, 0, 2>(out, cyc);
    run<0, 1>(out, cyc); run<0, 2>(out, cyc); run<2, 1>(out, cyc);
    return 0;
}
