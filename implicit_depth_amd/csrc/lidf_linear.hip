// lidf_linear.hip — generic f32 MFMA linear layer with fused epilogues, used by the PointNet2Stage
// forward (models/pointnet.py:22-38) and by the stage-2 refinement query.
//
//   out[row, 0:32*NT] = epilogue( X[row, 0:D] W^T + b  [+ u * xoff[row]] )
//   epilogue: (+ addrows[addidx[row]])  ->  (relu / leaky relu)  ->  (* slope mask of another tensor)
//             ->  (store or accumulate)  and/or  (atomic max into a pool)
// The mask / accumulate / u-column options serve the decoders' training path (lidf_train.hip).
//
// Same transposed 32x32x2 f32 MFMA formulation and the same packed weight-stream format as the
// decoder kernel (lidf_device.h, rows mode with NT output tiles): one wavefront owns 32 rows, lane
// = row, the two half-waves split the K columns. The max-pool epilogue replaces
// torch_scatter.scatter(..., reduce='max') (models/pointnet.py:27,35): values are post-ReLU
// (>= 0), so integer atomicMax on the bit pattern is an exact, order-independent float max, and a
// zero-initialised pool reproduces torch_scatter's 0 for a voxel without points.
#include "lidf_device.h"
#include <cstdlib>

#include "lidf_linear_kernel.inc"

// ------------------------------------------------------------------------------------------------
// Two row-local layers over a handful of rows in ONE launch (round 4): the per-voxel layers of
// PointNet2Stage — vox_lin1 -> the voxel half of point_lin3 (models/pointnet.py:29-33), and vox_lin2
// (:37) -> the voxel columns of the stage-2 decoder's layer 1 (models/pipeline.py:1016) — were one
// launch each over V = 50..150 rows: ~10 us of dispatch + operand latency for ~2 us of work, 11 per
// frame. A workgroup owns 32 rows; its four wavefronts split the output tiles of layer 1, leave the
// activated tiles in LDS as rows [x | 1 | 0...] (the bias operand rides as column D), and split the
// output tiles of layer 2, whose operands are 16-byte LDS reads. Streams, k order and the sequence
// of matrix instructions per output are those of lidf_linear_kernel: results are bit-identical to
// the two launches.
// ------------------------------------------------------------------------------------------------

#define VOX2_MAX_K 128
#define VOX2_KQ 17   // k-quads per tile at most: (128 + 2 + 7) / 8
// These launches are a chain of dependent memory round trips and little else (a tile is 68 matrix
// instructions): every weight quad and every operand of a tile is requested before the first matrix
// instruction, so a tile waits for memory once instead of once per k-quad (one k-quad ahead: 12.5 us
// for vox_lin1 + point_lin3's voxel half over 74 rows, 28.8 us for vox_lin2 + the 256-wide tail).
__global__ void __launch_bounds__(256) lidf_vox2_kernel(Vox2Args a) {
    __shared__ float s_x[32 * (VOX2_MAX_K + 12)];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int h = lane >> 5;
    const int col = lane & 31;
    const long long AN = a.n_dev ? (long long)*a.n_dev : a.n;
    const long long row0 = (long long)blockIdx.x * 32;
    if (row0 >= AN) return;
    const long long p = row0 + col;
    const bool valid = p < AN;
    const long long pc = valid ? p : AN - 1;
    const int vq = lane * 16;
    const int D2 = 32 * a.nt1, lds = D2 + 12;
    // ---- layer 1: output tile t by wavefront t % 4
    {
        const __amdgpu_buffer_rsrc_t srs =
            __builtin_amdgcn_make_buffer_rsrc((void*)a.s1, 0, a.kq1 * a.nt1 * 1024, 0x00020000);
        const float* xrow = a.X + (size_t)pc * a.ldx + 4 * h;
        // operand columns of this lane in k-quad kq: 8kq + 4h + {0..3}; column D1 = bias
        f32x4 xb[VOX2_KQ];
#pragma unroll
        for (int kq = 0; kq < VOX2_KQ; ++kq) {
            xb[kq] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int x0 = 8 * kq + 4 * h;
            if (kq < a.kq1) {
                if (x0 + 3 < a.D1) {
                    const f32x4u v = *(const f32x4u*)(xrow + 8 * kq);
                    xb[kq] = f32x4{v[0], v[1], v[2], v[3]};
                } else {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                        xb[kq][jj] = x0 + jj < a.D1 ? xrow[8 * kq + jj] : ((x0 + jj == a.D1 && a.bias1) ? 1.f : 0.f);
                }
            }
        }
        for (int t = wave; t < a.nt1; t += 4) {
            f32x4 q[VOX2_KQ];
#pragma unroll
            for (int kq = 0; kq < VOX2_KQ; ++kq)   // (beyond the stream's kq1 k-quads: out of range reads give 0)
                q[kq] = LDQ(srs, vq, (kq * a.nt1 + t) * 1024);
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int kq = 0; kq < VOX2_KQ; ++kq) {
                if (kq < a.kq1) {
                    acc = MFMA(q[kq][0], xb[kq][0], acc);
                    acc = MFMA(q[kq][1], xb[kq][1], acc);
                    acc = MFMA(q[kq][2], xb[kq][2], acc);
                    acc = MFMA(q[kq][3], xb[kq][3], acc);
                }
            }
            // this lane holds, per group g, features 32t + 8g + 4h + {0..3} of row `col`
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = a.relu1 ? fmaxf(acc[4 * g + i], 0.f * acc[4 * g + i]) : acc[4 * g + i];
                *(f32x4*)(s_x + col * lds + 32 * t + 8 * g + 4 * h) = v;
                if (a.out1 && valid) *(f32x4*)(a.out1 + (size_t)p * a.ld1 + 32 * t + 8 * g + 4 * h) = v;
            }
        }
        if (wave == 0 && h == 0) {   // the bias operand and the padding of the last k-quad
            *(f32x4*)(s_x + col * lds + D2) = f32x4{a.bias2 ? 1.f : 0.f, 0.f, 0.f, 0.f};
            *(f32x4*)(s_x + col * lds + D2 + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    if (!a.s2) return;
    // ---- layer 2: output tile t by wavefront t % 4, operands from LDS; the first tile's weight quads are
    //      requested before the barrier
    {
        const __amdgpu_buffer_rsrc_t srs =
            __builtin_amdgcn_make_buffer_rsrc((void*)a.s2, 0, a.kq2 * a.nt2 * 1024, 0x00020000);
        const float* xr = s_x + col * lds + 4 * h;
        const int kmax = (D2 + 8) / 8;   // k-quads the LDS row holds operands for; beyond: zero operands
        f32x4 q[VOX2_KQ];
#pragma unroll
        for (int kq = 0; kq < VOX2_KQ; ++kq) q[kq] = LDQ(srs, vq, (kq * a.nt2 + wave) * 1024);
        __syncthreads();
        for (int t = wave; t < a.nt2; t += 4) {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int kq = 0; kq < VOX2_KQ; ++kq) {
                if (kq < a.kq2) {
                    f32x4 b = {0.f, 0.f, 0.f, 0.f};
                    if (kq < kmax) b = *(const f32x4*)(xr + 8 * kq);
                    acc = MFMA(q[kq][0], b[0], acc);
                    acc = MFMA(q[kq][1], b[1], acc);
                    acc = MFMA(q[kq][2], b[2], acc);
                    acc = MFMA(q[kq][3], b[3], acc);
                }
            }
            if (t + 4 < a.nt2) {   // the next tile's quads while this one is stored
#pragma unroll
                for (int kq = 0; kq < VOX2_KQ; ++kq) q[kq] = LDQ(srs, vq, (kq * a.nt2 + t + 4) * 1024);
            }
            if (valid) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = a.relu2 ? fmaxf(acc[4 * g + i], 0.f * acc[4 * g + i]) : acc[4 * g + i];
                    *(f32x4*)(a.out2 + (size_t)p * a.ld2 + 32 * t + 8 * g + 4 * h) = v;
                }
            }
        }
    }
}

extern "C" hipError_t lidf_launch_vox2(const Vox2Args& a, hipStream_t st) {
    if (a.n <= 0) return hipSuccess;
    if (a.nt1 < 1 || 32 * a.nt1 > VOX2_MAX_K || a.kq1 > VOX2_KQ || (a.s2 && (a.nt2 < 1 || a.kq2 > VOX2_KQ)))
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(lidf_vox2_kernel, dim3((unsigned)((a.n + 31) / 32)), dim3(256), 0, st, a);
    return hipGetLastError();
}

// the SPLIT / XCOL instantiations: lidf_linear_s.hip, lidf_linear_x.hip, lidf_linear_sx.hip
extern "C" void lidf_launch_linear_s(int nt, dim3 g, dim3 b, hipStream_t st, const LinearArgs& a);
extern "C" void lidf_launch_linear_x(int nt, dim3 g, dim3 b, hipStream_t st, const LinearArgs& a);
extern "C" void lidf_launch_linear_sx(int nt, dim3 g, dim3 b, hipStream_t st, const LinearArgs& a);

// nt = accumulator tiles; a.xcol: the stream carries nt + 1 quads per k-quad, the last one the extra column's
extern "C" hipError_t lidf_launch_linear(int nt, const LinearArgs& a_in, int grid, hipStream_t st) {
    if (a_in.n <= 0) return hipSuccess;
    if (nt < 1 || nt > 8) return hipErrorInvalidValue;
    LinearArgs a = a_in;
    a.nt_total = 0;
    // operand rows from two buffers: whole k-quads on either side of the split, no tail columns of X
    if (a.X2 && (a.kq_split <= 0 || 8 * a.kq_split >= a.D || a.D % 8 != 0 || a.ldx2 < a.D - 8 * a.kq_split))
        return hipErrorInvalidValue;
    // the extra column: a plain or accumulating store, whole tiles in front of it, a launch of its own size
    if (a.xcol && (a.addrows || a.addrows2 || a.relu || a.mask_src || a.pool || !a.out || a.nout != 32 * nt))
        return hipErrorInvalidValue;
    const long long ntile = (a.n + 127) / 128;
    if (nt > 1 && ntile <= 16 && !a.xcol) {   // few rows: one workgroup per (row tile, output tile)
        a.nt_total = nt;
        if (a.X2) lidf_launch_linear_s(1, dim3((unsigned)ntile, (unsigned)nt), dim3(256), st, a);
        else hipLaunchKernelGGL((lidf_linear_kernel<1, false, false>), dim3((unsigned)ntile, (unsigned)nt), dim3(256), 0, st, a);
        return hipGetLastError();
    }
    dim3 g(grid), b(256);
    // Three or more accumulator tiles: ONE workgroup per compute unit (one wavefront per SIMD). The operand rows are
    // gathered a row stride apart (32 cache lines per wavefront and k-quad, each used by four consecutive k-quads):
    // with two workgroups per CU the lines of eight wavefronts do not survive in the L1 between their uses and the
    // launch runs at 0.57 of the matrix peak (614,400 x 256 -> 256) against 0.71 with one (scripts/linear_ubench.py);
    // the k-loop hides its memory latency by itself (operands two k-quads ahead, weights one).
    if (nt >= 3) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0 &&
            (long long)g.x > cus)
            g = dim3((unsigned)cus);
    }
    if (a.X2 && a.xcol) lidf_launch_linear_sx(nt, g, b, st, a);
    else if (a.X2) lidf_launch_linear_s(nt, g, b, st, a);
    else if (a.xcol) lidf_launch_linear_x(nt, g, b, st, a);
    else launch_linear_nt<false, false>(nt, g, b, st, a);
    return hipGetLastError();
}
