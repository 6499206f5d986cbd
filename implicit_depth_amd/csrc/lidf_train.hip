// lidf_train.hip — kernels of the decoders' training path (first step): the weight
// gradient reduction and the small per-row pieces around lidf_linear_kernel. The forward of the
// training path keeps every layer's activation (lidf_decoder_forward_train_f32), the backward
// (lidf_decoder_backward_f32) runs the input-gradient chain through lidf_linear_kernel with the
// transposed weights and the leaky-ReLU mask as epilogue, and reduces the weight gradients here.
#include "lidf_device.h"

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// C[m, j] += sum over rows r of A[r, m] * B[r, j]   (m < M, j < N), db[m] += sum_r A[r, m].
// One wavefront per (32x32 tile of C, slice of rows): lane (c, h) feeds A[r+h, 32mt+c] and
// B[r+h, 32nt+c] — 128 contiguous bytes per half-wave and matrix — into v_mfma_f32_32x32x2_f32 with
// k = the row pair; partial tiles are added with float atomics (summation order is not fixed).
struct WgradArgs {
    const float* A; long long lda; int M;
    const float* B; long long ldb; int N;
    long long n;
    float* C; int ldc;
    float* db;            // optional: column N of B is taken as 1
    int mtiles, ntiles, splits;
    long long rows_per_split;
};

__global__ void __launch_bounds__(256) lidf_wgrad_kernel(WgradArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, c = lane & 31;
    int id = blockIdx.x;
    const int mt = id % a.mtiles; id /= a.mtiles;
    const int nt = id % a.ntiles; id /= a.ntiles;
    const long long r0 = (long long)id * a.rows_per_split;
    long long r1 = r0 + a.rows_per_split;
    if (r1 > a.n) r1 = a.n;
    const int am = 32 * mt + c, bj = 32 * nt + c;
    const bool a_ok = am < a.M, b_ok = bj < a.N, b_one = a.db && bj == a.N;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // the four wavefronts interleave blocks of 16 rows
    for (long long r = r0 + wave * 16; r < r1; r += 64) {
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long row = r + 2 * u + h;
            const bool in = row < r1;
            av[u] = (in && a_ok) ? a.A[(size_t)row * a.lda + am] : 0.f;
            bv[u] = in ? (b_ok ? a.B[(size_t)row * a.ldb + bj] : (b_one ? 1.f : 0.f)) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = MFMA(av[u], bv[u], acc);
    }
    // result register q of lane (c, h): C row 32mt + (q&3) + 8(q>>2) + 4h, column 32nt + c
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int m = 32 * mt + (q & 3) + 8 * (q >> 2) + 4 * h;
        if (m >= a.M || acc[q] == 0.f) continue;
        if (bj < a.N)
            atomicAdd(a.C + (size_t)m * a.ldc + bj, acc[q]);
        else if (b_one)
            atomicAdd(a.db + m, acc[q]);
    }
}

extern "C" hipError_t lidf_launch_wgrad(const float* A, long long lda, int M, const float* B,
                                        long long ldb, int N, long long n, float* C, int ldc,
                                        float* db, hipStream_t st) {
    if (n <= 0 || M <= 0) return hipSuccess;
    WgradArgs a;
    a.A = A; a.lda = lda; a.M = M; a.B = B; a.ldb = ldb; a.N = N; a.n = n; a.C = C; a.ldc = ldc;
    a.db = db;
    a.mtiles = (M + 31) / 32;
    a.ntiles = (N + (db ? 1 : 0) + 31) / 32;
    a.rows_per_split = 4096;
    a.splits = (int)((n + a.rows_per_split - 1) / a.rows_per_split);
    const long long blocks = (long long)a.mtiles * a.ntiles * a.splits;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(lidf_wgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ---- per-row pieces -------------------------------------------------------------------------------
// enc[r, j] = off[r] * wenc[j] + benc[j]   (IEF.offset_enc, implicit_net.py:107,139)
__global__ void lidf_enc_rows_kernel(const float* off, const float* wenc, const float* benc,
                                     long long n, float* enc) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 16) return;
    const int j = (int)(i & 15);
    enc[i] = off[i >> 4] * wenc[j] + benc[j];
}

__global__ void lidf_fill_kernel(float* x, long long n, float v) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = v;
}

// out = act(pre) (implicit_net.py:93-96 / :148-151); with g != NULL also gpre = g * act'(pre)
__global__ void lidf_out_act_kernel(const float* pre, long long n, int use_sigmoid, float* out,
                                    const float* g, float* gpre) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float y = pre[i];
    float o, d;
    if (use_sigmoid) {
        o = 1.f / (1.f + expf(-y));
        d = o * (1.f - o);
    } else {
        // max(min(y, 0.01 y + 0.99), 0.01 y): identity on [0, 1], slope 0.01 outside
        o = fmaxf(fminf(y, y * 0.01f + 0.99f), y * 0.01f);
        d = (y >= 0.f && y <= 1.f) ? 1.f : 0.01f;
    }
    if (out) out[i] = o;
    if (g) gpre[i] = g[i] * d;
}

extern "C" hipError_t lidf_launch_enc_rows(const float* off, const float* wenc, const float* benc,
                                           long long n, float* enc, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_enc_rows_kernel, dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), 0,
                       st, off, wenc, benc, n, enc);
    return hipGetLastError();
}
extern "C" hipError_t lidf_launch_fill(float* x, long long n, float v, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n,
                       v);
    return hipGetLastError();
}
extern "C" hipError_t lidf_launch_out_act(const float* pre, long long n, int use_sigmoid,
                                          float* out, const float* g, float* gpre,
                                          hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_out_act_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                       pre, n, use_sigmoid, out, g, gpre);
    return hipGetLastError();
}
