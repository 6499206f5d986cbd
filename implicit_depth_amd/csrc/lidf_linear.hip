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

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define LDQ(rs, voff, soff) \
    __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128((rs), (voff), (soff), 0))

// SPLIT: the operand rows come from TWO buffers — k-quads [0, kq_split) from X, the ones behind them from X2 (the
// joint input gradient of the two decoders on materialised rows: d rows = [S_prob | S_off] [W1_prob ; W1_off], one
// K = 512 product and one store of the [n, 385] rows instead of two K = 256 products, two stores and autograd's add)
template <int NT, bool SPLIT>
__global__ void __launch_bounds__(256) lidf_linear_kernel(LinearArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int h = lane >> 5;
    const int col = lane & 31;
    // a.nt_total > 0: the launch is split over the output tiles (grid.y = tile; NT = 1): a layer over a
    // handful of rows (the per-voxel layers of the PointNet: 74 rows x 256 outputs) is one wavefront's
    // chain of NT x 4 matrix instructions per k-quad otherwise — 8 workgroups side by side instead
    const int NTS = a.nt_total > 0 ? a.nt_total : NT;      // tiles per k-quad in the stream
    const int T0 = a.nt_total > 0 ? (int)blockIdx.y : 0;   // first output tile of this workgroup
    const __amdgpu_buffer_rsrc_t srs =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.stream, 0, a.kq1 * NTS * 1024, 0x00020000);
    const int vq = lane * 16;
    const long long AN = a.n_dev ? (long long)*a.n_dev : a.n;   // device-side row count (frame path)
    const long long ntile = (AN + 127) / 128;
    // k-quads below kfull hold eight real operand columns for both half-waves: one 16-byte load per lane, no
    // condition. The k-quad(s) behind them carry the row's last columns, the bias and the u column.
    const int kfull = a.D / 8 < a.kq1 ? a.D / 8 : a.kq1;
    // Operand registers live across the tile loop: the next tile's first loads are issued in FRONT of this
    // tile's stores (vector memory completes in order — a load behind 32 stores waits for their acknowledgements,
    // docs/history.md §4.5), so the first matrix instruction of a tile waits for two loads, not for the previous
    // tile's output to reach memory.
    float b0[4], b1[4], b2[4], b3[4];   // operand ring: k-quads kq, kq + 1, (kq + 2, kq + 3)
    f32x4 q0[NT], q1[NT];               // weight quads of k-quad kq, kq + 1
    const float* xrow = nullptr;
    const float* xrow2 = nullptr;   // SPLIT: the second buffer's row, biased by -8 kq_split
    long long pc = 0;
    auto load_b_fast = [&](int kq, float (&b)[4]) {
        const float* xr = xrow;
        if (SPLIT) xr = kq >= a.kq_split ? xrow2 : xrow;   // (kq wave-uniform)
        const f32x4u v = *(const f32x4u*)(xr + 8 * kq);
        b[0] = v[0]; b[1] = v[1]; b[2] = v[2]; b[3] = v[3];
    };
    auto load_b_tail = [&](int kq, float (&b)[4]) {
        const int x0 = 8 * kq + 4 * h;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
            b[jj] = x0 + jj < a.D ? xrow[8 * kq + jj]
                                  : ((x0 + jj == a.D && a.has_bias)
                                         ? 1.f
                                         : ((x0 + jj == a.D + 1 && a.xoff) ? a.xoff[pc] : 0.f));
    };
    auto load_b = [&](int kq, float (&b)[4]) {   // (kq wave-uniform)
        if (kq < kfull) load_b_fast(kq, b);
        else if (kq < a.kq1) load_b_tail(kq, b);
        else { b[0] = 0.f; b[1] = 0.f; b[2] = 0.f; b[3] = 0.f; }
    };
    auto load_q = [&](int kq, f32x4 (&q)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; ++t) q[t] = LDQ(srs, vq, (kq * NTS + T0 + t) * 1024);
    };
    // rows of a tile for this lane; the tile's first requests: operands of k-quads 0 and 1, weight quads of k-quad 0
    auto tile_begin = [&](long long tile) {
        const long long p = tile * 128 + wave * 32 + col;
        pc = p < AN ? p : AN - 1;
        xrow = a.X + (size_t)pc * a.ldx + 4 * h;
        if (SPLIT) xrow2 = a.X2 + (size_t)pc * a.ldx2 + 4 * h - 8 * (long long)a.kq_split;
        load_b(0, b0);
        load_b(1, b1);
        load_q(0, q0);
    };
    long long tile = blockIdx.x;
    if (tile < ntile && tile * 128 + wave * 32 < AN) tile_begin(tile);
    for (; tile < ntile; tile += gridDim.x) {
        if (tile * 128 + wave * 32 >= AN) continue;   // (only the last tile can be short: no later tile for this wave)
        const long long p = tile * 128 + wave * 32 + col;
        const bool valid = p < AN;
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        }
        auto mfma_quad = [&](const f32x4 (&q)[NT], const float (&b)[4]) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x16 c = acc[t];
                c = MFMA(q[t][0], b[0], c);
                c = MFMA(q[t][1], b[1], c);
                c = MFMA(q[t][2], b[2], c);
                c = MFMA(q[t][3], b[3], c);
                acc[t] = c;
            }
        };
        int kq = 0;
        // main loop, four k-quads per trip through a ring of four operand sets and two weight sets (no copies):
        // the operand of k-quad kq + 2 and the weight quads of kq + 1 are requested BEFORE the 4 NT matrix
        // instructions of k-quad kq — the operand rows come from HBM (two k-quads of matrix time ahead), the
        // weight quads from L2 (one ahead). Left to itself the compiler sinks the requests behind the products
        // and every k-quad starts on a memory round trip.
        for (; kq + 5 < kfull; kq += 4) {
            load_b_fast(kq + 2, b2); load_q(kq + 1, q1); SCHED_FENCE(); mfma_quad(q0, b0); SCHED_FENCE();
            load_b_fast(kq + 3, b3); load_q(kq + 2, q0); SCHED_FENCE(); mfma_quad(q1, b1); SCHED_FENCE();
            load_b_fast(kq + 4, b0); load_q(kq + 3, q1); SCHED_FENCE(); mfma_quad(q0, b2); SCHED_FENCE();
            load_b_fast(kq + 5, b1); load_q(kq + 4, q0); SCHED_FENCE(); mfma_quad(q1, b3); SCHED_FENCE();
        }
        // the last k-quads (at most five whole ones + the tail k-quads), one per trip with register copies
        for (; kq < a.kq1; ++kq) {
            load_b(kq + 2, b2);
            load_q(kq + 1 < a.kq1 ? kq + 1 : kq, q1);
            SCHED_FENCE();
            mfma_quad(q0, b0);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) { b0[jj] = b1[jj]; b1[jj] = b2[jj]; }
#pragma unroll
            for (int t = 0; t < NT; ++t) q0[t] = q1[t];
            SCHED_FENCE();
        }
        // the next tile's first requests, in front of this tile's stores
        {
            const long long nx = tile + gridDim.x;
            if (nx < ntile && nx * 128 + wave * 32 < AN) tile_begin(nx);
            SCHED_FENCE();
        }
        // ---- epilogue: this lane holds, per tile t and group g, features 32t + 8g + 4h + {0..3}
        // a negative index = this row takes no part (its gathered term is absent, it is not pooled)
        const int ai = (valid && a.addrows) ? a.addidx[p] : -1;
        const float* ar = ai >= 0 ? a.addrows + (size_t)ai * a.ld_add + 4 * h : nullptr;
        const float* ar2 =
            (valid && a.addrows2) ? a.addrows2 + (size_t)a.addidx2[p] * a.ld_add2 + 4 * h : nullptr;
        float* op = (valid && a.out) ? a.out + (size_t)p * a.ld_out + 4 * h : nullptr;
        const float* mp = (valid && a.mask_src) ? a.mask_src + (size_t)p * a.ld_mask + 4 * h : nullptr;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = acc[t][4 * g + i];
                // (a.nout > 0: columns >= nout are padding of the last tile — no gathered term is read
                // for them, they are neither stored nor pooled)
                const int cg = (T0 + t) * 32 + 8 * g + 4 * h;
                const bool whole = a.nout <= 0 || cg + 3 < a.nout;
                if (ar) {
                    if (whole) {
                        const f32x4 r = *(const f32x4*)(ar + (T0 + t) * 32 + 8 * g);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] += r[i];
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (cg + i < a.nout) v[i] += ar[(T0 + t) * 32 + 8 * g + i];
                    }
                }
                if (ar2) {
                    if (whole) {
                        const f32x4 r = *(const f32x4*)(ar2 + (T0 + t) * 32 + 8 * g);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] += r[i];
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (cg + i < a.nout) v[i] += ar2[(T0 + t) * 32 + 8 * g + i];
                    }
                }
                if (a.relu) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], a.slope * v[i]);
                }
                const int c0 = (T0 + t) * 32 + 8 * g + 4 * h;  // first of this lane's four columns
                if (mp && (a.nout <= 0 || c0 + 3 < a.nout)) {
                    // dgrad through a leaky ReLU: the activation's output has the sign of its input
                    const f32x4 m = *(const f32x4*)(mp + (T0 + t) * 32 + 8 * g);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] *= m[i] > 0.f ? 1.f : a.mask_slope;
                } else if (mp && c0 < a.nout) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (c0 + i < a.nout) v[i] *= mp[(T0 + t) * 32 + 8 * g + i] > 0.f ? 1.f : a.mask_slope;
                }
                if (op) {
                    if (a.nout <= 0 || c0 + 3 < a.nout) {
                        if (a.accumulate) {
                            const f32x4 o = *(const f32x4*)(op + (T0 + t) * 32 + 8 * g);
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] += o[i];
                        }
                        *(f32x4*)(op + (T0 + t) * 32 + 8 * g) = v;
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (c0 + i < a.nout) {
                                float* q = op + (T0 + t) * 32 + 8 * g + i;
                                *q = a.accumulate ? *q + v[i] : v[i];
                            }
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[t][4 * g + i] = v[i];
            }
        }
        if (a.pool) {
            // max-pool into pool[poolidx[row]] (values are post-ReLU, >= 0, pool starts at 0).
            // Neighbouring rows mostly share their voxel, so the wave first reduces per distinct
            // voxel: members keep their value, the others contribute 0 (the identity), a 5-step
            // shuffle max over the 32 row lanes of each half, and one lane issues the atomics —
            // and only where a plain (possibly stale; entries only grow) read does not already
            // prove them unnecessary. More than 4 distinct voxels: per-lane atomics.
            const int vox = valid ? a.poolidx[p] : -1;
            unsigned todo = (unsigned)__ballot(valid && h == 0 && vox >= 0);
            int rounds = 0;
            while (todo && rounds < 4) {
                const int lead = __builtin_ctz(todo);
                const int vv = __builtin_amdgcn_readlane(vox, lead);
                const unsigned mem = (unsigned)__ballot(vox == vv) & todo;
                todo &= ~mem;
                ++rounds;
                const bool mine = vox == vv;
                int* pp = (int*)a.pool + (size_t)vv * a.ld_pool + 4 * h;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (a.nout > 0 && (T0 + t) * 32 >= a.nout) continue;   // padding tile (wave-uniform)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 m;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float x = mine ? acc[t][4 * g + i] : 0.f;
#pragma unroll
                            for (int sft = 16; sft >= 1; sft >>= 1) x = fmaxf(x, __shfl_xor(x, sft));
                            m[i] = x;
                        }
                        if (col == lead) {
                            const f32x4 seen = *(const f32x4*)(pp + (T0 + t) * 32 + 8 * g);
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (m[i] > seen[i])
                                    atomicMax(pp + (T0 + t) * 32 + 8 * g + i, __float_as_int(m[i]));
                        }
                    }
                }
            }
            if (todo && valid && ((todo >> col) & 1u)) {
                int* pp = (int*)a.pool + (size_t)vox * a.ld_pool + 4 * h;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (a.nout > 0 && (T0 + t) * 32 >= a.nout) continue;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 seen = *(const f32x4*)(pp + (T0 + t) * 32 + 8 * g);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (acc[t][4 * g + i] > seen[i])
                                atomicMax(pp + (T0 + t) * 32 + 8 * g + i, __float_as_int(acc[t][4 * g + i]));
                    }
                }
            }
        }
    }
}

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

extern "C" hipError_t lidf_launch_linear(int nt, const LinearArgs& a_in, int grid, hipStream_t st) {
    if (a_in.n <= 0) return hipSuccess;
    LinearArgs a = a_in;
    a.nt_total = 0;
    // operand rows from two buffers: whole k-quads on either side of the split, no tail columns of X
    if (a.X2 && (a.kq_split <= 0 || 8 * a.kq_split >= a.D || a.D % 8 != 0 || a.ldx2 < a.D - 8 * a.kq_split))
        return hipErrorInvalidValue;
    const long long ntile = (a.n + 127) / 128;
    if (nt > 1 && ntile <= 16) {   // few rows: one workgroup per (row tile, output tile)
        a.nt_total = nt;
        if (a.X2) hipLaunchKernelGGL((lidf_linear_kernel<1, true>), dim3((unsigned)ntile, (unsigned)nt), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((lidf_linear_kernel<1, false>), dim3((unsigned)ntile, (unsigned)nt), dim3(256), 0, st, a);
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
    if (a.X2) {
        switch (nt) {
            case 1: hipLaunchKernelGGL((lidf_linear_kernel<1, true>), g, b, 0, st, a); break;
            case 2: hipLaunchKernelGGL((lidf_linear_kernel<2, true>), g, b, 0, st, a); break;
            case 3: hipLaunchKernelGGL((lidf_linear_kernel<3, true>), g, b, 0, st, a); break;
            case 4: hipLaunchKernelGGL((lidf_linear_kernel<4, true>), g, b, 0, st, a); break;
            case 5: hipLaunchKernelGGL((lidf_linear_kernel<5, true>), g, b, 0, st, a); break;
            case 6: hipLaunchKernelGGL((lidf_linear_kernel<6, true>), g, b, 0, st, a); break;
            case 7: hipLaunchKernelGGL((lidf_linear_kernel<7, true>), g, b, 0, st, a); break;
            case 8: hipLaunchKernelGGL((lidf_linear_kernel<8, true>), g, b, 0, st, a); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (nt) {
        case 1: hipLaunchKernelGGL((lidf_linear_kernel<1, false>), g, b, 0, st, a); break;
        case 2: hipLaunchKernelGGL((lidf_linear_kernel<2, false>), g, b, 0, st, a); break;
        case 3: hipLaunchKernelGGL((lidf_linear_kernel<3, false>), g, b, 0, st, a); break;
        case 4: hipLaunchKernelGGL((lidf_linear_kernel<4, false>), g, b, 0, st, a); break;
        case 5: hipLaunchKernelGGL((lidf_linear_kernel<5, false>), g, b, 0, st, a); break;
        case 6: hipLaunchKernelGGL((lidf_linear_kernel<6, false>), g, b, 0, st, a); break;
        case 7: hipLaunchKernelGGL((lidf_linear_kernel<7, false>), g, b, 0, st, a); break;
        case 8: hipLaunchKernelGGL((lidf_linear_kernel<8, false>), g, b, 0, st, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
