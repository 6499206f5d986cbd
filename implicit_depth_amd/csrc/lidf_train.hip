// lidf_train.hip — kernels of the decoders' training path (first step): the weight
// gradient reduction and the small per-row pieces around lidf_linear_kernel. The forward of the
// training path keeps every layer's activation (lidf_decoder_forward_train_f32), the backward
// (lidf_decoder_backward_f32) runs the input-gradient chain through lidf_linear_kernel with the
// transposed weights and the leaky-ReLU mask as epilogue, and reduces the weight gradients here.
#include "lidf_device.h"
#include <stdlib.h>

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// C[m, j] += sum over rows r of A[r, m] * B[r, j]   (m < M, j < N), db[m] += sum_r A[r, m].
// One wavefront per (64x64 tile of C, slice of rows): lane (c, h) feeds A[r+h, 64mt+32t+c] and
// B[r+h, 64nt+32t+c] — 128 contiguous bytes per half-wave and matrix — into v_mfma_f32_32x32x2_f32
// with k = the row pair; partial tiles are added with float atomics (summation order is not fixed).
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
    // one wavefront: a 64 x 64 tile of C (2 x 2 matrix tiles), so every loaded value feeds two
    // instructions
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, c = lane & 31;
    long long id = (long long)blockIdx.x * 4 + wave;  // every wavefront is an independent unit
    if (id >= (long long)a.mtiles * a.ntiles * a.splits) return;
    const int mt = (int)(id % a.mtiles); id /= a.mtiles;
    const int nt = (int)(id % a.ntiles); id /= a.ntiles;
    const long long r0 = id * a.rows_per_split;
    long long r1 = r0 + a.rows_per_split;
    if (r1 > a.n) r1 = a.n;
    int am[2], bj[2];
    bool a_ok[2], b_ok[2], b_one[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        am[t] = 64 * mt + 32 * t + c;
        bj[t] = 64 * nt + 32 * t + c;
        a_ok[t] = am[t] < a.M;
        b_ok[t] = bj[t] < a.N;
        b_one[t] = a.db && bj[t] == a.N;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t >> 1][t & 1][i] = 0.f;
    }
    // 8 rows per step; the loads of the next step are in flight while this one multiplies.
    // Buffer loads over this unit's rows: the row advance is a scalar offset, rows past the end
    // and columns past the matrix fall outside the descriptor and read as 0 — no per-load
    // address arithmetic or bounds test on the vector unit.
    const int nrows = (int)(r1 - r0);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.A + (size_t)r0 * a.lda), 0, (int)((size_t)nrows * a.lda * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.B + (size_t)r0 * a.ldb), 0, (int)((size_t)nrows * a.ldb * 4), 0x00020000);
    int va[2], vb[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        va[t] = a_ok[t] ? (int)((h * a.lda + am[t]) * 4) : 0x7ffffff0;
        vb[t] = b_ok[t] ? (int)((h * a.ldb + bj[t]) * 4) : 0x7ffffff0;
    }
    const int sa = (int)(a.lda * 8), sbb = (int)(a.ldb * 8);  // two rows
    float av[4][2], bv[4][2], an[4][2], bn[4][2];
    auto load = [&](int r, float (&x)[4][2], float (&y)[4][2]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                x[u][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, va[t], (r / 2 + u) * sa, 0));
                const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, vb[t], (r / 2 + u) * sbb, 0));
                y[u][t] = b_one[t] ? ((r + 2 * u + h < nrows) ? 1.f : 0.f) : v;
            }
        }
    };
    load(0, av, bv);
    for (int r = 0; r < nrows; r += 8) {
        load(r + 8, an, bn);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[t >> 1][t & 1] = MFMA(av[u][t >> 1], bv[u][t & 1], acc[t >> 1][t & 1]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                av[u][t] = an[u][t];
                bv[u][t] = bn[u][t];
            }
        }
    }
    // result register q of lane (c, h) in tile (tm, tn): C row 64mt + 32tm + (q&3) + 8(q>>2) + 4h,
    // column 64nt + 32tn + c
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int tm = t >> 1, tn = t & 1;
        const int col = bj[tn];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int m = 64 * mt + 32 * tm + (q & 3) + 8 * (q >> 2) + 4 * h;
            const float v = acc[tm][tn][q];
            if (m >= a.M || v == 0.f) continue;
            if (col < a.N)
                atomicAdd(a.C + (size_t)m * a.ldc + col, v);
            else if (b_one[tn])
                atomicAdd(a.db + m, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient for the wide layers: one workgroup = a 128 x 256 block of C over a slice of rows,
// so A and B stream from HBM once per block instead of once per 64 x 64 tile (the 64-wide tiling
// re-reads A N/64 times and B M/64 times: 2.5 GB for the 128 x 256 layer at 614,400 rows).
// Lane (c, h) reads row r+h, columns 4c..4c+3 of A — element j of that float4 feeds the matrix
// instructions of row tile j, which therefore holds the C rows {4i + j} — and the B columns
// {4c + e} of its wavefront's column tiles (wavefront w: tiles {2w, 2w+1} of the 8, or tile w of 4):
// a permutation of rows and columns that the write-back undoes. Rows past the slice / matrix read as 0 through the buffer descriptors;
// columns past M / N accumulate garbage that is never stored.
// ------------------------------------------------------------------------------------------------
struct Wgrad2Args {
    const float* A; long long lda; int M;
    const float* B; long long ldb; int N;
    long long n;
    float* C; int ldc;
    float* db;
    long long rows_per_split;   // multiple of 8
    float* part;                // optional [splits, mb, nb, LIDF_WG_SLAB] partial blocks
    int bcol0, acol0;           // B / A point at this column of their rows: the last row ends that many floats earlier
    int swap;                   // 1 (SW instantiation): the caller handed the operands over SWAPPED — this launch's A is
                                // the product's B (<= 128 columns), its B the product's A (<= 256 columns): a 256 x 102
                                // or 256 x 128 gradient is ONE block of eight accumulator tiles per wavefront instead of
                                // two single-half blocks of four. C is addressed transposed (C[col * ldc + m]), db holds
                                // the column sums of B, xcol names an extra column of A.
    int xcol;                   // >= 0: column xcol of B (and of C) is one more column, beyond the block's N: its
                                // products with the block's A rows are 4 vector FMAs per row pair beside the matrix
                                // instructions (385 = 3 x 128 + 1 columns of the decoders' input rows: the one
                                // column would otherwise occupy a 128-column half block of its own). Not with HM.
};

typedef float f32x4w __attribute__((ext_vector_type(4)));
#define LDX4(rs, voff, soff) \
    __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128((rs), (voff), (soff), 0))

// TWO: the block has a second half of 128 columns (N - n0 > 128); without it the four accumulator
// tiles of that half and their matrix instructions are left out (N = 102 or 128 operands).
// HM (not TWO): M <= 64 (layer 3: dW3 = dZ3^T H2, 64 x 128) — the block is 64 x 128, lane (c, h) takes
// the A columns {2c, 2c+1}, wavefront w the column 2c + (w & 1) and the two column tiles
// {2 (w >> 1), 2 (w >> 1) + 1}: two matrix instructions per row pair instead of four on a half-empty
// block (16 rows per step, so that a step is still 16 matrix instructions between barriers).
template <bool TWO, bool HM, bool XC = false, bool SW = false>
__global__ void __launch_bounds__(256, 2) lidf_wgrad2_kernel(Wgrad2Args a) {
    static_assert(!SW || (TWO && !HM), "the swapped form is the two-half block");
    // RS rows of A (128 columns) and of B (256 columns) per step, staged once per workgroup through
    // LDS (double-buffered): the four wavefronts read the same B rows and the same A float4
    constexpr int RS = HM ? 16 : 8;
    __shared__ f32x4w sA[2][RS][HM ? 16 : 32];   // [buffer][row][float4 column]
    __shared__ f32x4w sB[2][RS][HM ? 32 : 64];
    __shared__ float sX[2][RS];                  // the extra column's RS values of a step
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, c = lane & 31;
    const int m0 = HM ? 0 : blockIdx.y * 128, n0 = HM ? 0 : blockIdx.z * 256;
    const long long r0 = (long long)blockIdx.x * a.rows_per_split;
    long long r1 = r0 + a.rows_per_split;
    if (r1 > a.n) r1 = a.n;
    const int nrows = r0 < r1 ? (int)(r1 - r0) : 0;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.A + (size_t)r0 * a.lda), 0, nrows > 0 ? (int)(((size_t)nrows * a.lda - a.acol0) * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.B + (size_t)r0 * a.ldb), 0, nrows > 0 ? (int)(((size_t)nrows * a.ldb - a.bcol0) * 4) : 0, 0x00020000);
    // staging role of this thread: row tr (0..7) of the step, float4 column tc (A), tc and tc+32 (B);
    // HM: rows tr and tr + 8, A only from the threads tc < 16
    const int tr = threadIdx.x >> 5, tc = threadIdx.x & 31;
    const bool second = TWO && n0 + 128 < a.N;
    const int ga = (HM && tc >= 16) ? 0x7ffffff0 : (int)((tr * a.lda + m0 + 4 * tc) * 4);
    const int gb0 = (int)((tr * a.ldb + n0 + 4 * tc) * 4);
    const int gb1 = second ? gb0 + 512 : 0x7ffffff0;
    const int sa = (int)(a.lda * 4 * RS), sbb = (int)(a.ldb * 4 * RS);  // RS rows
    const int ha = (int)(a.lda * 32), hb = (int)(a.ldb * 32);           // HM: the thread's second row
    constexpr bool xc = XC && !HM;   // (a launch with xcol >= 0 takes the XC instantiation)
    const int gx = (xc && tc == 0) ? (int)((tr * (SW ? a.lda : a.ldb) + a.xcol) * 4) : 0x7ffffff0;
    const __amdgpu_buffer_rsrc_t rx = SW ? ra : rb;   // the extra column lives in the operand that has <= 128 columns
    const int sx = SW ? (int)(a.lda * 4 * RS) : (int)(a.ldb * 4 * RS);
    f32x2 bs2 = {0.f, 0.f}, xs2 = {0.f, 0.f};          // swapped form: sums over this lane's two B columns
    constexpr int NT = HM ? 2 : (TWO ? 8 : 4);
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    }
    f32x4w asum = {0.f, 0.f, 0.f, 0.f}, xsum = {0.f, 0.f, 0.f, 0.f};
    float asum1 = 0.f;
    const int nstep = (nrows + RS - 1) / RS;
    f32x4w ga4 = LDX4(ra, ga, 0), gb4 = LDX4(rb, gb0, 0), gc4 = {0.f, 0.f, 0.f, 0.f};
    float gx1 = 0.f;
    if (xc) gx1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, gx, 0, 0));
    f32x4w ga5 = gc4, gb5 = gc4;
    if (TWO) gc4 = LDX4(rb, gb1, 0);
    if (HM) {
        ga5 = LDX4(ra, ga, ha);
        gb5 = LDX4(rb, gb0, hb);
    }
    auto stage = [&](int buf) {
        if (HM) {
            if (tc < 16) {
                sA[buf][tr][tc] = ga4;
                sA[buf][tr + 8][tc] = ga5;
            }
            sB[buf][tr][tc] = gb4;
            sB[buf][tr + 8][tc] = gb5;
        } else {
            sA[buf][tr][tc] = ga4;
            sB[buf][tr][tc] = gb4;
            if (TWO) sB[buf][tr][32 + tc] = gc4;
            if (xc && tc == 0) sX[buf][tr] = gx1;
        }
    };
    stage(0);
    __syncthreads();
    for (int st = 0; st < nstep; ++st) {
        const int cur = st & 1;
        if (st + 1 < nstep) {   // next step's rows: global -> registers while this step multiplies
            ga4 = LDX4(ra, ga, (st + 1) * sa);
            gb4 = LDX4(rb, gb0, (st + 1) * sbb);
            if (TWO) gc4 = LDX4(rb, gb1, (st + 1) * sbb);
            if (xc) gx1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, gx, (st + 1) * sx, 0));
            if (HM) {
                ga5 = LDX4(ra, ga, (st + 1) * sa + ha);
                gb5 = LDX4(rb, gb0, (st + 1) * sbb + hb);
            }
        }
        if constexpr (HM) {
            const float* fa = (const float*)&sA[cur][0][0];
            const float* fb = (const float*)&sB[cur][0][0];
#pragma unroll
            for (int u = 0; u < RS / 2; ++u) {
                const float aw = fa[(2 * u + h) * 64 + 2 * c + (wave & 1)];
                const f32x2 b2 = *(const f32x2*)(fb + (2 * u + h) * 128 + 4 * c + 2 * (wave >> 1));
                acc[0] = MFMA(aw, b2[0], acc[0]);
                acc[1] = MFMA(aw, b2[1], acc[1]);
                asum1 += aw;
            }
        } else {
            // wavefront w owns ALL 128 rows of the block (row tile j = element j of the A float4) and
            // the column tiles {2w, 2w+1} (TWO) or {w}: 24 or 20 bytes of LDS per lane and row pair
            // instead of the 48 of a row-owning split, where every wavefront reads every B value
            const float* fb = (const float*)&sB[cur][0][0];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4w a4 = sA[cur][2 * u + h][c];
                if constexpr (TWO) {
                    const f32x2 b2 = *(const f32x2*)(fb + (2 * u + h) * 256 + 128 * (wave >> 1) + 4 * c + 2 * (wave & 1));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[2 * j] = MFMA(a4[j], b2[0], acc[2 * j]);
                        acc[2 * j + 1] = MFMA(a4[j], b2[1], acc[2 * j + 1]);
                    }
                    if constexpr (SW) {
                        bs2 += b2;
                        if (xc) xs2 += b2 * sX[cur][2 * u + h];
                    }
                } else {
                    const float b1 = fb[(2 * u + h) * 256 + 4 * c + wave];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = MFMA(a4[j], b1, acc[j]);
                }
                if constexpr (!SW) {
                    asum += a4;
                    if (xc) xsum += a4 * sX[cur][2 * u + h];
                }
            }
        }
        if (st + 1 < nstep) stage(cur ^ 1);
        __syncthreads();
    }
    // register q of lane (c, h), i = (q&3) + 8(q>>2) + 4h, in tile e:
    //   TWO: C row m0 + 4i + (e>>1), column n0 + 128(wave>>1) + 4c + 2(wave&1) + (e&1)
    //   else: C row m0 + 4i + e,     column n0 + 4c + wave
    //   HM:   C row 2i + (wave&1),   column 4c + 2(wave>>1) + e
    auto c_row = [&](int q, int e) {
        const int i = (q & 3) + 8 * (q >> 2) + 4 * h;
        return HM ? 2 * i + (wave & 1) : TWO ? 4 * i + (e >> 1) : 4 * i + e;
    };
    auto c_col = [&](int e) {
        return HM ? 4 * c + 2 * (wave >> 1) + e : TWO ? 128 * (wave >> 1) + 4 * c + 2 * (wave & 1) + (e & 1) : 4 * c + wave;
    };
    if (HM) {   // column sums of A: lanes c and c+32 hold the two rows of a pair
        asum1 += __shfl_xor(asum1, 32);
    }
    if (a.part) {
        // deterministic path: this block's partial sums go to its own slab, summed by
        // lidf_wgrad_reduce_kernel in a fixed order
        float* slab = a.part + ((size_t)(blockIdx.x * gridDim.y + blockIdx.y) * gridDim.z + blockIdx.z) * LIDF_WG_SLAB;
#pragma unroll
        for (int e = 0; e < NT; ++e) {
#pragma unroll
            for (int q = 0; q < 16; ++q) slab[c_row(q, e) * 256 + c_col(e)] = acc[e][q];
        }
        if (SW) {   // every wavefront holds the sums of its own two B columns per lane
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float v = bs2[e], x = xs2[e];
                v += __shfl_xor(v, 32);
                x += __shfl_xor(x, 32);
                if (h == 0) {
                    slab[128 * 256 + c_col(e)] = v;
                    if (xc) slab[128 * 256 + 256 + c_col(e)] = x;
                }
            }
        } else if (HM) {
            if (wave < 2 && h == 0) slab[128 * 256 + 2 * c + wave] = asum1;
        } else if (wave == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = asum[e];
                v += __shfl_xor(v, 32);
                if (h == 0) slab[128 * 256 + 4 * c + e] = v;
            }
        } else if (wave == 1 && xc) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = xsum[e];
                v += __shfl_xor(v, 32);
                if (h == 0) slab[128 * 256 + 256 + 4 * c + e] = v;
            }
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < NT; ++e) {
        const int col = n0 + c_col(e);
        if (col >= a.N) continue;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int m = m0 + c_row(q, e);
            const float v = acc[e][q];
            if (m < a.M && v != 0.f) atomicAdd(SW ? a.C + (size_t)col * a.ldc + m : a.C + (size_t)m * a.ldc + col, v);
        }
    }
    if (SW) {   // column sums of B / products with the extra column of A: first row block only
        if (blockIdx.y == 0) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float v = bs2[e], x = xs2[e];
                v += __shfl_xor(v, 32);
                x += __shfl_xor(x, 32);
                const int col = n0 + c_col(e);
                if (h == 0 && col < a.N) {
                    if (a.db && v != 0.f) atomicAdd(a.db + col, v);
                    if (xc && x != 0.f) atomicAdd(a.C + (size_t)col * a.ldc + a.xcol, x);
                }
            }
        }
        return;
    }
    // bias gradient: column sums of A (first column block only); lanes c and c+32 hold the two rows
    if (HM) {
        const int m = 2 * c + wave;
        if (a.db && wave < 2 && h == 0 && m < a.M && asum1 != 0.f) atomicAdd(a.db + m, asum1);
    } else if (a.db && blockIdx.z == 0 && wave == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = asum[e];
            v += __shfl_xor(v, 32);
            const int m = m0 + 4 * c + e;
            if (h == 0 && m < a.M && v != 0.f) atomicAdd(a.db + m, v);
        }
    }
    if (!HM && xc && blockIdx.z == 0 && wave == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = xsum[e];
            v += __shfl_xor(v, 32);
            const int m = m0 + 4 * c + e;
            if (h == 0 && m < a.M && v != 0.f) atomicAdd(a.C + (size_t)m * a.ldc + a.xcol, v);
        }
    }
}

// C[m, col] += sum over the row slices of the partial blocks, in a fixed order (deterministic):
// thread (entry group g of 32 float4 entries, slice group k of 8) sums every 8th slice of its
// float4, the 8 slice groups are combined through LDS in order k = 0..7.
__global__ void __launch_bounds__(256) lidf_wgrad_reduce_kernel(const float* __restrict__ part,
                                                                int splits, int mb, int nb, int M,
                                                                int N, float* __restrict__ C, int ldc,
                                                                float* __restrict__ db, int xcol, int swap) {
    __shared__ f32x4w red[8][32];
    const int blk = blockIdx.y;                  // (m block, n block)
    const int bm = blk / nb, bn = blk % nb;
    const int e4 = blockIdx.x * 32 + (threadIdx.x & 31);   // float4 entry of the 128 x 256 (+ 128 + 128) slab
    const int k = threadIdx.x >> 5;
    constexpr int SLAB = LIDF_WG_SLAB;
    f32x4w s = {0.f, 0.f, 0.f, 0.f};
    // entries beyond the matrix (rows >= M, columns >= N of a partial block: a 64 x 128 or 256 x 102
    // operand fills a quarter or half of its slab; the single-half kernel never writes columns
    // 128..255) are neither read nor written
    bool used = e4 < SLAB / 4;
    if (used && 4 * e4 < 128 * 256) {
        const int m = bm * 128 + (4 * e4) / 256, col = bn * 256 + (4 * e4) % 256;
        used = m < M && col < N;
    } else if (used) {   // the bias sums, the extra column's sums: only where the producer wrote them
        const int i = (4 * e4 - 128 * 256) % 256;
        const bool own = swap ? (bm == 0 && bn * 256 + i < N) : (bn == 0 && i < 128 && bm * 128 + i < M);
        used = own && (4 * e4 < 128 * 256 + 256 ? db != nullptr : xcol >= 0);
    }
    if (used) {
        const float* p = part + ((size_t)bm * nb + bn) * SLAB + 4 * (size_t)e4;
        // (eight slices in flight per thread: the loop is a chain of independent 16-byte loads a slab apart)
        const size_t stride = (size_t)mb * nb * SLAB;
        int sp = k;
        for (; sp + 56 < splits; sp += 64) {
            f32x4w v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *(const f32x4w*)(p + (size_t)(sp + 8 * j) * stride);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; sp < splits; sp += 8) s += *(const f32x4w*)(p + (size_t)sp * stride);
    }
    red[k][threadIdx.x & 31] = s;
    __syncthreads();
    if (k != 0 || e4 >= SLAB / 4) return;
#pragma unroll
    for (int j = 1; j < 8; ++j) s += red[j][threadIdx.x & 31];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = 4 * e4 + i;
        if (e < 128 * 256) {
            const int m = bm * 128 + e / 256, col = bn * 256 + e % 256;
            if (m < M && col < N) C[swap ? (size_t)col * ldc + m : (size_t)m * ldc + col] += s[i];
        } else {
            // (entry j of a run of sums belongs to row bm * 128 + j of C — swapped: to column bn * 256 + j of B)
            const bool second = e >= 128 * 256 + 256;
            const int j = e - 128 * 256 - (second ? 256 : 0);
            const int at = swap ? bn * 256 + j : bm * 128 + j;
            const bool own = swap ? (bm == 0 && at < N) : (bn == 0 && j < 128 && at < M);
            if (!own) continue;
            if (!second) {
                if (db) db[at] += s[i];
            } else if (xcol >= 0) {
                C[(size_t)at * ldc + xcol] += s[i];
            }
        }
    }
}

// One launch of the block kernel over the columns [0, N) of B / C (+ optionally the single column xcol >= N through the
// vector unit) and its reduce. false: the operand does not fit the block kernel's 32-bit offsets.
static bool wgrad2_launch(const float* A, long long lda, int M, const float* B, long long ldb, int bcol0, int N,
                          int xcol, long long n, float* C, int ldc, float* db, float* g_wgrad_scratch,
                          size_t g_wgrad_scratch_floats, hipStream_t st) {
    Wgrad2Args w;
    w.A = A; w.lda = lda; w.M = M; w.B = B; w.ldb = ldb; w.N = N; w.n = n; w.C = C; w.ldc = ldc;
    w.db = db; w.xcol = xcol; w.bcol0 = bcol0; w.acol0 = 0; w.swap = 0;
    // A gradient of more than 128 rows and at most 128 (+ 1) columns — S^T PE (256 x 102), the 128 + 1 remainder of
    // the decoders' 385 input columns — with the operands swapped: one two-half block (eight accumulator tiles per
    // wavefront, 24 bytes of LDS per lane and 8 matrix instructions) where the product's own orientation is two
    // single-half blocks of four tiles (20 bytes per 4). Not for operands of a few hundred rows (one slice per block:
    // two blocks are the parallelism there).
    if (M > 128 && N <= 128 && n > 320) {
        w.A = B; w.lda = ldb; w.M = N; w.acol0 = bcol0;
        w.B = A; w.ldb = lda; w.N = M; w.bcol0 = 0;
        w.swap = 1;
    }
    const int mb = (w.M + 127) / 128, nb = (w.N + 255) / 256;
    // two workgroups per CU; with a scratch area for the partial blocks the slices are as long
    // as possible (fewer partial sums), without it shorter slices keep the atomics spread
    static int slab_budget = -1;   // development knob (A/B runs): LIDF_WGRAD_SPLITS = row slices per product, <= 512
    if (slab_budget < 0) {
        const char* e = getenv("LIDF_WGRAD_SPLITS");
        const int v = e ? atoi(e) : 0;
        slab_budget = (v >= 64 && v <= 512) ? v : 512;
    }
    long long splits = (g_wgrad_scratch ? slab_budget : 1024) / (mb * nb);
    // (slices of at least 64 rows: a per-ray or per-voxel operand of 76,800 or 729 rows still
    // fills the chip instead of walking its rows in a few workgroups)
    const long long max_splits = g_wgrad_scratch ? (n + 63) / 64 : (n + 1023) / 1024;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    // per-voxel operands (a frame has 50-150 occupied voxels, a training batch a few hundred rows): ONE slice
    // per block of C, added straight into C — every element receives exactly one contribution, so the
    // result is as deterministic as the slab path's, without the reduce launch that cost as much as the
    // product itself (10 + 5 us per per-voxel layer, a dozen of them per training step)
    const bool one_slice = g_wgrad_scratch && n <= 320;   // (measured: 729 rows in one slice 60 us, in 12 slices + reduce 35)
    if (one_slice) splits = 1;
    const bool half_m = !w.swap && M <= 64 && N <= 128 && mb == 1 && nb == 1 && xcol < 0;   // 64 x 128 block (layer 3)
    const int rs = half_m ? 16 : 8;
    w.rows_per_split = ((n + splits - 1) / splits + rs - 1) / rs * rs;
    const long long sp = (n + w.rows_per_split - 1) / w.rows_per_split;
    const size_t slice_bytes = (size_t)w.rows_per_split * (size_t)(lda > ldb ? lda : ldb) * 4;
    const size_t need = (size_t)sp * mb * nb * LIDF_WG_SLAB;
    w.part = (g_wgrad_scratch && need <= g_wgrad_scratch_floats && !one_slice) ? g_wgrad_scratch : nullptr;
    if (slice_bytes >= 0x7fffffffULL) return false;
    const dim3 grid((unsigned)sp, mb, nb), blk(256);
    const bool x = xcol >= 0;
    if (w.swap) {
        if (x) hipLaunchKernelGGL((lidf_wgrad2_kernel<true, false, true, true>), grid, blk, 0, st, w);
        else hipLaunchKernelGGL((lidf_wgrad2_kernel<true, false, false, true>), grid, blk, 0, st, w);
    } else if (half_m)
        hipLaunchKernelGGL((lidf_wgrad2_kernel<false, true>), grid, blk, 0, st, w);
    else if (N - (nb - 1) * 256 > 128 || nb > 1) {
        if (x) hipLaunchKernelGGL((lidf_wgrad2_kernel<true, false, true>), grid, blk, 0, st, w);
        else hipLaunchKernelGGL((lidf_wgrad2_kernel<true, false>), grid, blk, 0, st, w);
    } else if (x)
        hipLaunchKernelGGL((lidf_wgrad2_kernel<false, false, true>), grid, blk, 0, st, w);
    else
        hipLaunchKernelGGL((lidf_wgrad2_kernel<false, false>), grid, blk, 0, st, w);
    if (w.part)
        hipLaunchKernelGGL(lidf_wgrad_reduce_kernel, dim3((LIDF_WG_SLAB / 4 + 31) / 32, mb * nb),
                           dim3(256), 0, st, w.part, (int)sp, mb, nb, w.M, w.N, C, ldc, db, xcol, w.swap);
    return true;
}

extern "C" hipError_t lidf_launch_wgrad(const float* A, long long lda, int M, const float* B,
                                        long long ldb, int N, long long n, float* C, int ldc,
                                        float* db, float* g_wgrad_scratch,
                                        size_t g_wgrad_scratch_floats, hipStream_t st) {
    if (n <= 0 || M <= 0) return hipSuccess;
    // the wide layers: 128 x 256 blocks (float4 loads want lda/ldb*4 within the 32-bit offsets the
    // buffer instructions take, and a slice of rows below 2 GiB)
    if (M >= 32 && N >= 4) {
        // Column plan. The block kernel multiplies whole 256-column blocks (two 128-column halves per wavefront
        // pair) or one 128-column block: columns are padded to that. Instead of ceil(N / 256) blocks of 256
        // (385 columns of the decoders' input rows: 512 multiplied, a quarter of the matrix instructions on
        // zeros — and 127 of the last 128 for ONE column), the full 256-column blocks go in one launch and the
        // remainder in a second one of its own width: <= 128 columns as a single-half block, and a remainder
        // of 128 k + 1 columns hands its last column to the vector unit (Wgrad2Args.xcol: 4 FMAs per row pair
        // beside 16-32 matrix instructions). 385 = [256] + [128 + 1]: 384 columns multiplied.
        const int full = (N / 256) * 256, rem = N - full;
        const auto go = [&](const float* Bp, int Ncols, int xcol, float* Cp, float* dbp) {
            return wgrad2_launch(A, lda, M, Bp, ldb, (int)(Bp - B), Ncols, xcol, n, Cp, ldc, dbp, g_wgrad_scratch,
                                 g_wgrad_scratch_floats, st);
        };
        bool ok;
        if (full > 0 && rem == 1)                       // 257, 513: the leftover column rides with the full blocks
            ok = go(B, full, full, C, db);
        else if (full > 0 && rem > 1 && rem <= 129) {   // 385 = [256] + [128 + 1], 300 = [256] + [44]
            const bool x = rem == 129;
            ok = go(B, full, -1, C, db) && go(B + full, x ? 128 : rem, x ? 128 : -1, C + full, nullptr);
        } else if (full == 0 && rem == 129)             // 129 = [128 + 1]
            ok = go(B, 128, 128, C, db);
        else
            ok = go(B, N, -1, C, db);
        if (ok) return hipGetLastError();
    }
    WgradArgs a;
    a.A = A; a.lda = lda; a.M = M; a.B = B; a.ldb = ldb; a.N = N; a.n = n; a.C = C; a.ldc = ldc;
    a.db = db;
    a.mtiles = (M + 63) / 64;
    a.ntiles = (N + (db ? 1 : 0) + 63) / 64;
    // enough units to fill the chip (~8 wavefronts per SIMD), as few as possible beyond that: every
    // unit ends with 4096 atomic adds
    const long long tiles = (long long)a.mtiles * a.ntiles;
    long long splits = (8192 + tiles - 1) / tiles;
    const long long max_splits = (n + 511) / 512;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    a.rows_per_split = ((n + splits - 1) / splits + 7) / 8 * 8;
    a.splits = (int)((n + a.rows_per_split - 1) / a.rows_per_split);
    const long long blocks = (tiles * a.splits + 3) / 4;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(lidf_wgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ---- per-row pieces -------------------------------------------------------------------------------
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

extern "C" hipError_t lidf_launch_out_act(const float* pre, long long n, int use_sigmoid,
                                          float* out, const float* g, float* gpre,
                                          hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_out_act_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                       pre, n, use_sigmoid, out, g, gpre);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Narrow ends of the decoder backward as single passes over their wide operand (each was a chain of
// rank-deficient GEMM launches that re-read it):
//
//  * layer 4 (64 -> 1): dZ3 = (goff (x) w4) * lrelu'(H3), d w4 = sum_p goff[p] H3[p,:],
//    d b4 = sum_p goff[p] — one read of H3.
//  * the 16 offset-encoding columns of IEF's layer 1 (implicit_net.py:107,139-141). With
//    enc[p,j] = wenc[j] off[p] + benc[j] every product with `enc` collapses onto two column sums
//    of dZ1:   A[c] = sum_p dZ1[p,c] off[p],  B[c] = sum_p dZ1[p,c]
//      d W1[c, D+j] = wenc[j] A[c] + benc[j] B[c]
//      d wenc[j]    = sum_c W1[c, D+j] A[c] ;  d benc[j] = sum_c W1[c, D+j] B[c]
//      d off[p]    += dZ1[p,:] . u ,  u[c] = sum_j W1[c, D+j] wenc[j]
//    — one read of dZ1, and the same pass keeps the running sum S of dZ1 over the passes
//    (s_mode 1: S = dZ1, 2: S += dZ1) that the query's pass-independent layer-1 operands need.
//
// Column sums: per-workgroup partial vectors, summed in a fixed order by lidf_colsum_reduce_kernel
// (deterministic).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

__global__ void __launch_bounds__(256) lidf_l4_backward_kernel(
    const float* __restrict__ goff, const float* __restrict__ h3, const float* __restrict__ w4,
    float slope, long long n, long long rows_per_wg, float* __restrict__ dz3,
    float* __restrict__ part) {
    __shared__ f32x4w red[4][16];
    __shared__ float redb[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane >> 4, c4 = lane & 15;
    const f32x4w w = *(const f32x4w*)(w4 + 4 * c4);
    const long long r0 = (long long)blockIdx.x * rows_per_wg;
    const long long r1 = r0 + rows_per_wg < n ? r0 + rows_per_wg : n;
    f32x4w acc = {0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    // 16 rows per workgroup step: wavefront `wave` takes rows 4 wave + sub
    for (long long r = r0 + 4 * wave + sub; r < r1; r += 16) {
        const float g = goff[r];
        const f32x4w h = *(const f32x4w*)(h3 + (size_t)r * 64 + 4 * c4);
        f32x4w d;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            d[i] = __fmul_rn(__fmul_rn(w[i], g), h[i] > 0.f ? 1.f : slope);
            acc[i] = fmaf(g, h[i], acc[i]);
        }
        *(f32x4w*)(dz3 + (size_t)r * 64 + 4 * c4) = d;
        bsum += g;
    }
    // the four row sub-groups of the wavefront, then the four wavefronts, in a fixed order
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        acc[i] += __shfl_xor(acc[i], 16);
        acc[i] += __shfl_xor(acc[i], 32);
    }
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    if (sub == 0) red[wave][c4] = acc;
    if (lane == 0) redb[wave] = bsum;
    __syncthreads();
    if (threadIdx.x < 16) {
        f32x4w s = red[0][threadIdx.x];
#pragma unroll
        for (int k = 1; k < 4; ++k) s += red[k][threadIdx.x];
        *(f32x4w*)(part + (size_t)blockIdx.x * 65 + 4 * threadIdx.x) = s;
    }
    if (threadIdx.x == 16) part[(size_t)blockIdx.x * 65 + 64] = redb[0] + redb[1] + redb[2] + redb[3];
}

__global__ void __launch_bounds__(256) lidf_ief_tail_kernel(
    const float* __restrict__ dz1, const float* __restrict__ off, const float* __restrict__ w1enc,
    int ld1, const float* __restrict__ wenc, long long n, long long rows_per_wg, int s_mode,
    float* __restrict__ S, float* __restrict__ goff, float* __restrict__ part) {
    __shared__ float su[256];
    __shared__ f32x4w red[4][2][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        float u = 0.f;
        if (w1enc) {
#pragma unroll
            for (int j = 0; j < 16; ++j) u = fmaf(w1enc[(size_t)threadIdx.x * ld1 + j], wenc[j], u);
        }
        su[threadIdx.x] = u;
    }
    __syncthreads();
    const f32x4w u4 = *(const f32x4w*)(su + 4 * lane);
    const long long r0 = (long long)blockIdx.x * rows_per_wg;
    const long long r1 = r0 + rows_per_wg < n ? r0 + rows_per_wg : n;
    f32x4w A = {0.f, 0.f, 0.f, 0.f}, B = {0.f, 0.f, 0.f, 0.f};
    // one row per wavefront and step (1 KiB), four steps in flight
    for (long long r = r0 + wave; r < r1; r += 16) {
        f32x4w v[4], sv[4];
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long rr = r + 4 * j;
            if (rr < r1) {
                v[j] = *(const f32x4w*)(dz1 + (size_t)rr * 256 + 4 * lane);
                o[j] = off ? off[rr] : 0.f;
                if (s_mode == 2) sv[j] = *(const f32x4w*)(S + (size_t)rr * 256 + 4 * lane);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long rr = r + 4 * j;
            if (rr >= r1) break;
            if (s_mode == 1) *(f32x4w*)(S + (size_t)rr * 256 + 4 * lane) = v[j];
            if (s_mode == 2) *(f32x4w*)(S + (size_t)rr * 256 + 4 * lane) = sv[j] + v[j];
            if (w1enc) {
                float d = v[j][0] * u4[0];
                d = fmaf(v[j][1], u4[1], d);
                d = fmaf(v[j][2], u4[2], d);
                d = fmaf(v[j][3], u4[3], d);
                d = wave_sum(d);
                if (lane == 0) goff[rr] += d;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    A[i] = fmaf(v[j][i], o[j], A[i]);
                    B[i] += v[j][i];
                }
            }
        }
    }
    if (!w1enc) return;
    red[wave][0][lane] = A;
    red[wave][1][lane] = B;
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, l = threadIdx.x & 63;
        f32x4w s = red[0][which][l];
#pragma unroll
        for (int k = 1; k < 4; ++k) s += red[k][which][l];
        *(f32x4w*)(part + (size_t)blockIdx.x * 512 + 256 * which + 4 * l) = s;
    }
}

// out[c] = sum over the G partial vectors (stride ncol) in a fixed order: a workgroup owns 16
// columns, thread (k, c) sums every 16th partial (eight loads in flight), the sixteen are combined
// in order.
__global__ void __launch_bounds__(256) lidf_colsum_reduce_kernel(const float* __restrict__ part, int G,
                                                                 int ncol, float* __restrict__ out) {
    __shared__ float red[16][16];
    const int k = threadIdx.x >> 4, cl = threadIdx.x & 15, c = blockIdx.x * 16 + cl;
    float s = 0.f;
    if (c < ncol) {
        int g = k;
        for (; g + 112 < G; g += 128) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = part[(size_t)(g + 16 * j) * ncol + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += x[j];
        }
        for (; g < G; g += 16) s += part[(size_t)g * ncol + c];
    }
    red[k][cl] = s;
    __syncthreads();
    if (k == 0 && c < ncol) {
        float t = red[0][cl];
#pragma unroll
        for (int j = 1; j < 16; ++j) t += red[j][cl];
        out[c] = t;
    }
}

// d w4 += sums[0:64], d b4 += sums[64]
__global__ void lidf_l4_finish_kernel(const float* __restrict__ sums, float* __restrict__ dw4,
                                      float* __restrict__ db4) {
    const int c = threadIdx.x;
    if (c < 64) dw4[c] += sums[c];
    if (c == 64) db4[0] += sums[64];
}
// lidf_colsum_reduce_kernel (ncol = 65) + lidf_l4_finish_kernel in one launch: the column's sum over the G
// partial vectors, in the same fixed order, added straight into d w4 / d b4
__global__ void __launch_bounds__(256) lidf_l4_reduce_finish_kernel(const float* __restrict__ part, int G,
                                                                    float* __restrict__ dw4,
                                                                    float* __restrict__ db4) {
    __shared__ float red[16][16];
    const int k = threadIdx.x >> 4, cl = threadIdx.x & 15, c = blockIdx.x * 16 + cl;
    float s = 0.f;
    if (c < 65) {
        int g = k;
        for (; g + 112 < G; g += 128) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = part[(size_t)(g + 16 * j) * 65 + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += x[j];
        }
        for (; g < G; g += 16) s += part[(size_t)g * 65 + c];
    }
    red[k][cl] = s;
    __syncthreads();
    if (k == 0 && c < 65) {
        float t = red[0][cl];
#pragma unroll
        for (int j = 1; j < 16; ++j) t += red[j][cl];
        if (c < 64) dw4[c] += t;
        else db4[0] += t;
    }
}

// sums = [A | B] (256 each): d W1[c, D+j] += wenc[j] A[c] + benc[j] B[c] ; d wenc[j] += W1[:, D+j] . A ;
// d benc[j] += W1[:, D+j] . B      (one workgroup of 256 threads)
// `bacc` (optional, 256): the B of the passes finished so far. `btot` != NULL = the pass whose
// offset-in is the constant initial offset `init` for every row (the first pass of the IEF, the
// last one the backward reaches): its sums follow from the column sums of the running sum of dZ1
// over ALL passes, B = btot - bacc, A = init B — no sweep over its dZ1 is needed.
__global__ void __launch_bounds__(256) lidf_ief_finish_kernel(
    const float* __restrict__ sums, const float* __restrict__ w1enc, int ld1,
    const float* __restrict__ wenc, const float* __restrict__ benc, float* __restrict__ dw1enc,
    float* __restrict__ dwenc, float* __restrict__ dbenc, float* __restrict__ bacc,
    const float* __restrict__ btot, float init) {
    __shared__ float sA[256], sB[256];
    const int c = threadIdx.x;
    float A, B;
    if (btot) {
        B = btot[c] - bacc[c];
        A = init * B;
    } else {
        A = sums[c];
        B = sums[256 + c];
        if (bacc) bacc[c] += B;
    }
    sA[c] = A;
    sB[c] = B;
#pragma unroll
    for (int j = 0; j < 16; ++j) dw1enc[(size_t)c * ld1 + j] += fmaf(wenc[j], A, benc[j] * B);
    __syncthreads();
    // 32 dot products of length 256 (j, A|B): thread (piece of 32 rows, which, j), pieces in order
    __shared__ float sp[8][32];
    {
        const int j = c & 15, which = (c >> 4) & 1, piece = c >> 5;
        const float* v = which ? sB : sA;
        float t = 0.f;
#pragma unroll 8
        for (int i = 32 * piece; i < 32 * piece + 32; ++i) t = fmaf(w1enc[(size_t)i * ld1 + j], v[i], t);
        sp[piece][c & 31] = t;
    }
    __syncthreads();
    if (c < 32) {
        float t = sp[0][c];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += sp[k][c];
        if (c < 16) dwenc[c] += t;
        else dbenc[c - 16] += t;
    }
}

static inline long long narrow_rows_per_wg(long long n, int* G) {
    long long g = (n + 255) / 256;           // at least 256 rows per workgroup
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    const long long rows = ((n + g - 1) / g + 15) / 16 * 16;
    *G = (int)((n + rows - 1) / rows);
    return rows;
}

// scratch: at least 1024 * 65 + 65 floats
extern "C" hipError_t lidf_launch_l4_backward(const float* goff, const float* h3, const float* w4,
                                              float slope, long long n, float* dz3, float* dw4,
                                              float* db4, float* scratch, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    int G;
    const long long rows = narrow_rows_per_wg(n, &G);
    float* sums = scratch + (size_t)G * 65;
    hipLaunchKernelGGL(lidf_l4_backward_kernel, dim3(G), dim3(256), 0, st, goff, h3, w4, slope, n,
                       rows, dz3, scratch);
    (void)sums;
    hipLaunchKernelGGL(lidf_l4_reduce_finish_kernel, dim3(5), dim3(256), 0, st, scratch, G, dw4, db4);
    return hipGetLastError();
}

// scratch: at least 1024 * 512 + 512 floats. w1enc = W1 + D (NULL: no offset encoding, only the
// running sum S is kept); s_mode 0: S untouched.
extern "C" hipError_t lidf_launch_ief_tail(const float* dz1, const float* off, const float* w1enc,
                                           int ld1, const float* wenc, const float* benc,
                                           long long n, int s_mode, float* S, float* goff,
                                           float* dw1enc, float* dwenc, float* dbenc,
                                           float* bacc, float* scratch, hipStream_t st) {
    if (n <= 0 || (!w1enc && s_mode == 0)) return hipSuccess;
    int G;
    const long long rows = narrow_rows_per_wg(n, &G);
    float* sums = scratch + (size_t)G * 512;
    hipLaunchKernelGGL(lidf_ief_tail_kernel, dim3(G), dim3(256), 0, st, dz1, off, w1enc, ld1, wenc,
                       n, rows, s_mode, S, goff, scratch);
    if (w1enc) {
        hipLaunchKernelGGL(lidf_colsum_reduce_kernel, dim3(32), dim3(256), 0, st, scratch, G, 512, sums);
        hipLaunchKernelGGL(lidf_ief_finish_kernel, dim3(1), dim3(256), 0, st, sums, w1enc, ld1, wenc,
                           benc, dw1enc, dwenc, dbenc, bacc, (const float*)nullptr, 0.f);
    }
    return hipGetLastError();
}

// The IEF's first pass without a sweep over its dZ1 (see lidf_ief_finish_kernel): btot = column sums
// of the running sum of dZ1 over all passes, bacc = the B sums of the other passes.
extern "C" hipError_t lidf_launch_ief_first_pass(const float* btot, float* bacc, float init,
                                                 const float* w1enc, int ld1, const float* wenc,
                                                 const float* benc, float* dw1enc, float* dwenc,
                                                 float* dbenc, hipStream_t st) {
    hipLaunchKernelGGL(lidf_ief_finish_kernel, dim3(1), dim3(256), 0, st, (const float*)nullptr, w1enc,
                       ld1, wenc, benc, dw1enc, dwenc, dbenc, bacc, btot, init);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Training path of the query (LIDF.get_embedding, models/pipeline.py:338-420, with gradients):
// the decoder input rows are materialised — the reference's own formulation — so that the
// decoders' training path above applies; their gradient is reduced back to the voxel features and
// the per-ray features, and the ROI part of the latter through RoIAlign to the feature map.
// ------------------------------------------------------------------------------------------------
// row p = [ vox_feat[pair_vox] (128) | rayfeat[pair_ray][0:128] | embed(enter) | embed(leave) |
//           rayfeat[pair_ray][128:128+Ed] ]; one wavefront per row, lanes stride the columns.
__global__ void __launch_bounds__(256) lidf_build_rows_kernel(
    const int* __restrict__ pair_ray, const int* __restrict__ pair_vox, const float* __restrict__ pair_t,
    const float* __restrict__ ray_dir, const float* __restrict__ vox_center, int pos_rel,
    const float* __restrict__ vox_feat, const float* __restrict__ rayfeat, int ld_rf, int L, int Ed,
    long long P, float* __restrict__ rows, int D) {
    const int lane = threadIdx.x & 63;
    const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    const int r = pair_ray[p], v = pair_vox[p];
    const float te = pair_t[2 * p], tl = pair_t[2 * p + 1];
    const float d[3] = {ray_dir[3 * (size_t)r], ray_dir[3 * (size_t)r + 1], ray_dir[3 * (size_t)r + 2]};
    float c[3] = {0.f, 0.f, 0.f};
    if (pos_rel) {
        c[0] = vox_center[3 * (size_t)v];
        c[1] = vox_center[3 * (size_t)v + 1];
        c[2] = vox_center[3 * (size_t)v + 2];
    }
    const int E = 3 + 6 * L;
    float* o = rows + (size_t)p * D;
    for (int j = lane; j < D; j += 64) {
        float val;
        if (j < 128) {
            val = vox_feat[(size_t)v * 128 + j];
        } else if (j < 256) {
            val = rayfeat[(size_t)r * ld_rf + (j - 128)];
        } else if (j < 256 + 2 * E) {
            const int k = (j - 256) % E;
            const float t = j - 256 < E ? te : tl;
            // positional encoding element k of the position d*t (- voxel centre): models/implicit_net.py:30-39
            const int i = k < 3 ? k : (k - 3) % 3;
            const float x = __fmul_rn(d[i], t) - c[i];
            if (k < 3) {
                val = x;
            } else {
                const int oct = (k - 3) / 6;
                const float a = x * (float)(1 << oct);
                val = ((k - 3) % 6) < 3 ? sinf(a) : cosf(a);
            }
        } else {
            val = rayfeat[(size_t)r * ld_rf + 128 + (j - 256 - 2 * E)];
        }
        o[j] = val;
    }
}

// d rayfeat[r] = sum of the row gradients of the ray's (contiguous) pairs; one wavefront per ray
__global__ void __launch_bounds__(256) lidf_rows_ray_backward_kernel(
    const float* __restrict__ d_rows, int D, int E2, const int* __restrict__ pair_off, long long R,
    int Ed, float* __restrict__ d_rayfeat, int ld_rf) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int beg = pair_off[r], end = pair_off[r + 1];
    for (int j = lane; j < 128 + Ed; j += 64) {
        const int colr = j < 128 ? 128 + j : 256 + E2 + (j - 128);
        float acc = 0.f;
        for (int p = beg; p < end; ++p) acc += d_rows[(size_t)p * D + colr];
        d_rayfeat[(size_t)r * ld_rf + j] = acc;
    }
}

// d vox_feat[v] += row gradient columns 0..127 (pairs of a voxel are scattered over the rays:
// float atomics, summation order not fixed)
__global__ void __launch_bounds__(256) lidf_rows_vox_backward_kernel(
    const float* __restrict__ d_rows, int D, const int* __restrict__ pair_vox, long long P,
    float* __restrict__ d_vox_feat) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= P * 128) return;
    const long long p = i >> 7;
    const int j = (int)(i & 127);
    const float g = d_rows[(size_t)p * D + j];
    if (g != 0.f) atomicAdd(d_vox_feat + (size_t)pair_vox[p] * 128 + j, g);
}

// rays per pixel (the zeroed [B,H,W] table pix_rays)
__global__ void lidf_rayfeat_count_kernel(const int* __restrict__ ray_pix, const int* __restrict__ ray_bid,
                                          long long R, int H, int W, int* __restrict__ pix_rays) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    atomicAdd(pix_rays + ((size_t)ray_bid[r] * H + ray_pix[2 * r + 1]) * W + ray_pix[2 * r], 1);
}

// The bilinear weights of the gh x gw samples of a bin are a product of a row and a column factor,
// so a pixel's share of the bin is wy(py) wx(px).
__device__ __forceinline__ void roi_tap(float x, int LIM, int& lo, int& hi, float& l, bool& ok) {
    ok = !(x < -1.0f || x > (float)LIM);
    if (x <= 0.f) x = 0.f;
    lo = (int)x;
    if (lo >= LIM - 1) { hi = lo = LIM - 1; x = (float)lo; } else { hi = lo + 1; }
    l = x - (float)lo;
}
__device__ __forceinline__ float roi_axis_weight(float start, float bin, int n, int LIM, int px) {
    float wsum = 0.f;
    for (int i = 0; i < n; ++i) {
        int lo, hi; float l; bool ok;
        roi_tap(start + ((float)i + .5f) * bin / (float)n, LIM, lo, hi, l, ok);
        if (!ok) continue;
        if (lo == px) wsum += 1.f - l;
        if (hi == px) wsum += l;
    }
    return wsum;
}

// The boxes clamped at the image border (fractional samples), from the parked image: a workgroup
// takes a 16 x 16 tile of source pixels on the ring of clamped boxes and one channel, adds the
// shares of its pixels' four bins into an LDS patch of (18 + 2 half)^2 pixels (LDS float adds) and
// flushes the patch with one global add per touched pixel — the boxes of neighbouring border rays
// overlap almost entirely, a thread per (ray, column) with global adds spent 0.27 ms on 4,416 rays.
// grid.x = ring tiles (rayfeat_ring_tile), grid.y = B * 32.
__device__ __forceinline__ void rayfeat_ring_tile(int t, int tx_n, int ty_n, int& tx, int& ty) {
    // top row, bottom row, then the left and right columns between them
    if (t < tx_n) { ty = 0; tx = t; return; }
    t -= tx_n;
    if (ty_n > 1) {
        if (t < tx_n) { ty = ty_n - 1; tx = t; return; }
        t -= tx_n;
    }
    const int rows = ty_n - 2;   // (with a single tile column the launcher counts these once)
    if (t < rows) { ty = 1 + t; tx = 0; return; }
    t -= rows;
    ty = 1 + t; tx = tx_n - 1;
}
__global__ void __launch_bounds__(256) lidf_rayfeat_backward_ring_kernel(
    const float* __restrict__ gimg, int half, int H, int W, float* __restrict__ d_feat) {
    extern __shared__ float patch[];
    // (the box of a ray starts half a pixel before column u1: its first samples may fall into
    // column u1 - 1 — one pixel of margin beyond the boxes on every side)
    const int PW = 18 + 2 * half;
    for (int i = threadIdx.x; i < PW * PW; i += 256) patch[i] = 0.f;
    __syncthreads();
    const int tx_n = (W + 15) / 16, ty_n = (H + 15) / 16;
    int tx, ty;
    rayfeat_ring_tile(blockIdx.x, tx_n, ty_n, tx, ty);
    const int b = blockIdx.y >> 5, c = blockIdx.y & 31;
    const int qx = tx * 16 + (threadIdx.x & 15), qy = ty * 16 + (threadIdx.x >> 4);
    const int ox = tx * 16 - half - 1, oy = ty * 16 - half - 1;   // patch origin
    const bool clampedbox = qx < W && qy < H && (qx < half || qx > W - 1 - half || qy < half || qy > H - 1 - half);
    if (clampedbox) {
        const int u1 = min(max(qx - half, 0), W - 1), u2 = min(max(qx + half, 0), W - 1);
        const int v1 = min(max(qy - half, 0), H - 1), v2 = min(max(qy + half, 0), H - 1);
        const float rsw = (float)u1 - 0.5f, rsh = (float)v1 - 0.5f;
        const float roi_w = ((float)u2 - 0.5f) - rsw, roi_h = ((float)v2 - 0.5f) - rsh;
        const float bin_w = roi_w / 2.f, bin_h = roi_h / 2.f;
        const int gw = (int)ceilf(roi_w / 2.f), gh = (int)ceilf(roi_h / 2.f);
        if (gw > 0 && gh > 0) {
#pragma unroll
            for (int bin = 0; bin < 4; ++bin) {
                const int ph = bin >> 1, pw = bin & 1;
                const float g = gimg[(((size_t)b * 128 + c * 4 + bin) * H + qy) * W + qx];
                if (g == 0.f) continue;
                const float gs = g / (float)(gh * gw);
                const float xs = rsw + (float)pw * bin_w, ys = rsh + (float)ph * bin_h;
                int x0, x1, y0, y1, t0; float tl; bool tk;
                roi_tap(xs + .5f * bin_w / (float)gw, W, x0, t0, tl, tk);
                roi_tap(xs + ((float)(gw - 1) + .5f) * bin_w / (float)gw, W, t0, x1, tl, tk);
                roi_tap(ys + .5f * bin_h / (float)gh, H, y0, t0, tl, tk);
                roi_tap(ys + ((float)(gh - 1) + .5f) * bin_h / (float)gh, H, t0, y1, tl, tk);
                // column factors once per bin (x1 - x0 <= gw <= half <= 16)
                float wxv[17];
#pragma unroll
                for (int i = 0; i < 17; ++i)
                    wxv[i] = x0 + i <= x1 ? roi_axis_weight(xs, bin_w, gw, W, x0 + i) : 0.f;
                for (int py = y0; py <= y1; ++py) {
                    const float wy = roi_axis_weight(ys, bin_h, gh, H, py);
                    if (wy == 0.f) continue;
                    float* prow = patch + (py - oy) * PW + (x0 - ox);
#pragma unroll
                    for (int i = 0; i < 17; ++i)
                        if (wxv[i] != 0.f) atomicAdd(prow + i, gs * wy * wxv[i]);
                }
            }
        }
    }
    __syncthreads();
    float* img = d_feat + ((size_t)b * 32 + c) * H * W;
    for (int i = threadIdx.x; i < PW * PW; i += 256) {
        const float v = patch[i];
        const int py = oy + i / PW, px = ox + i % PW;
        if (v != 0.f && py >= 0 && py < H && px >= 0 && px < W) atomicAdd(img + (size_t)py * W + px, v);
    }
}

// RoIAlign backward (torchvision roi_align, output 2x2, aligned): every sample of bin (ph, pw)
// passes g / count to its four bilinear taps.
// A workgroup takes 64 rays: their 128 gradient columns are read as rows (coalesced) into LDS, then
// a wavefront walks the columns with lane = ray — consecutive rays are neighbouring pixels, so the
// 64 adds of a column land in one or two lines of its image plane (a thread per (ray, column) put
// every add of a wavefront into a different plane: 0.87 ms for 76,800 rays).
// `gimg` != NULL: rays whose box is not clamped are left to the gather pair below.
__global__ void __launch_bounds__(256) lidf_rayfeat_backward_kernel(
    const float* __restrict__ d_rayfeat, int ld_rf, const int* __restrict__ ray_pix,
    const int* __restrict__ ray_bid, long long R, int half, int H, int W, float* __restrict__ d_feat,
    float* __restrict__ gimg, const int* __restrict__ pix_rays, int park_all) {
    __shared__ float tile[64][129];
    const long long r0 = (long long)blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 128; i += 256) {
        const long long r = r0 + (i >> 7);
        tile[i >> 7][i & 127] = r < R ? d_rayfeat[(size_t)r * ld_rf + (i & 127)] : 0.f;
    }
    __syncthreads();
    const int rl = threadIdx.x & 63;
    const long long r = r0 + rl;
    if (r >= R) return;
    const int qx = ray_pix[2 * r], qy = ray_pix[2 * r + 1], bid = ray_bid[r];
    const int u1 = min(max(qx - half, 0), W - 1), u2 = min(max(qx + half, 0), W - 1);
    const int v1 = min(max(qy - half, 0), H - 1), v2 = min(max(qy + half, 0), H - 1);
    // park_all: the clamped boxes are parked too (lidf_rayfeat_backward_ring_kernel takes them)
    const bool parked = gimg && half > 0 && (park_all || (u2 - u1 == 2 * half && v2 - v1 == 2 * half));
    // pix_rays (rays parked per pixel, lidf_rayfeat_count_kernel): the only ray of its pixel stores
    const bool alone = parked && pix_rays && pix_rays[((size_t)bid * H + qy) * W + qx] == 1;
    const float rsw = (float)u1 - 0.5f, rsh = (float)v1 - 0.5f;
    const float roi_w = ((float)u2 - 0.5f) - rsw, roi_h = ((float)v2 - 0.5f) - rsh;
    const float bin_w = roi_w / 2.f, bin_h = roi_h / 2.f;
    const int gw = (int)ceilf(roi_w / 2.f), gh = (int)ceilf(roi_h / 2.f);
    for (int cb = threadIdx.x >> 6; cb < 128; cb += 4) {
        const float g = tile[rl][cb];
        if (g == 0.f) continue;
        if (parked) {
            // unclamped box: every sample sits on a pixel centre, bin (ph, pw) spreads g / half^2
            // over a half x half pixel block; parked at the ray's pixel, gathered by
            // lidf_rayfeat_gather_kernel (into the zeroed image; added atomically where two rays
            // name the same pixel)
            float* dst = gimg + (((size_t)bid * 128 + cb) * H + qy) * W + qx;
            if (alone) *dst = g;
            else atomicAdd(dst, g);
            continue;
        }
        const int c = cb >> 2, ph = (cb >> 1) & 1, pw = cb & 1;
        const float gs = g / (float)max(gh * gw, 1);
        float* img = d_feat + ((size_t)bid * 32 + c) * H * W;
        for (int iy = 0; iy < gh; ++iy) {
            float y = rsh + (float)ph * bin_h + ((float)iy + .5f) * bin_h / (float)gh;
            for (int ix = 0; ix < gw; ++ix) {
                float x = rsw + (float)pw * bin_w + ((float)ix + .5f) * bin_w / (float)gw;
                float yy = y;
                // the taps of bilinear() in lidf_aux.hip
                if (yy < -1.0f || yy > (float)H || x < -1.0f || x > (float)W) continue;
                if (yy <= 0.f) yy = 0.f;
                if (x <= 0.f) x = 0.f;
                int y_low = (int)yy, x_low = (int)x, y_high, x_high;
                if (y_low >= H - 1) { y_high = y_low = H - 1; yy = (float)y_low; } else { y_high = y_low + 1; }
                if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else { x_high = x_low + 1; }
                const float ly = yy - (float)y_low, lx = x - (float)x_low, hy = 1.f - ly, hx = 1.f - lx;
                if (hy * hx != 0.f) atomicAdd(img + y_low * W + x_low, gs * hy * hx);
                if (hy * lx != 0.f) atomicAdd(img + y_low * W + x_high, gs * hy * lx);
                if (ly * hx != 0.f) atomicAdd(img + y_high * W + x_low, gs * ly * hx);
                if (ly * lx != 0.f) atomicAdd(img + y_high * W + x_high, gs * ly * lx);
            }
        }
    }
}

extern "C" hipError_t lidf_launch_build_rows(const int* pair_ray, const int* pair_vox,
                                             const float* pair_t, const float* ray_dir,
                                             const float* vox_center, int pos_rel,
                                             const float* vox_feat, const float* rayfeat, int ld_rf,
                                             int L, int Ed, long long P, float* rows, int D,
                                             hipStream_t st) {
    if (P <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_build_rows_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, st,
                       pair_ray, pair_vox, pair_t, ray_dir, vox_center, pos_rel, vox_feat, rayfeat,
                       ld_rf, L, Ed, P, rows, D);
    return hipGetLastError();
}

extern "C" hipError_t lidf_launch_rows_backward(const float* d_rows, int D, int E2,
                                                const int* pair_off, const int* pair_vox,
                                                long long R, long long P, int Ed, float* d_vox_feat,
                                                float* d_rayfeat, int ld_rf, hipStream_t st) {
    if (d_rayfeat && R > 0)
        hipLaunchKernelGGL(lidf_rows_ray_backward_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0,
                           st, d_rows, D, E2, pair_off, R, Ed, d_rayfeat, ld_rf);
    if (d_vox_feat && P > 0)
        hipLaunchKernelGGL(lidf_rows_vox_backward_kernel, dim3((unsigned)((P * 128 + 255) / 256)),
                           dim3(256), 0, st, d_rows, D, pair_vox, P, d_vox_feat);
    return hipGetLastError();
}

// d_feat[b,c,y,x] += 1/half^2 * sum over the 4 bins of the parked gradients of the rays whose bin
// covers the pixel: bin ph = 0 is covered by rays at rows y+1 .. y+half, ph = 1 by rows
// y-half+1 .. y (columns alike) — the adjoint of the forward's box-sum shortcut, no atomics.
// lo: only source pixels in [lo, W-1-lo] x [lo, H-1-lo] (with the clamped boxes parked too, lo = half
// leaves them to the ring kernel).
__global__ void __launch_bounds__(256) lidf_rayfeat_gather_kernel(const float* __restrict__ gimg,
                                                                  int B, int H, int W, int half, int lo,
                                                                  float* __restrict__ d_feat) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * 32 * H * W;
    if (i >= total) return;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const long long bc = i / ((long long)W * H);
    const long long b = bc / 32;
    const int c = (int)(bc % 32);
    float acc = 0.f;
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
        const int ya = ph ? y - half + 1 : y + 1, yb = ph ? y : y + half;
#pragma unroll
        for (int pw = 0; pw < 2; ++pw) {
            const int xa = pw ? x - half + 1 : x + 1, xb = pw ? x : x + half;
            const float* g = gimg + ((size_t)(b * 128 + c * 4 + ph * 2 + pw) * H) * W;
            for (int yy = max(ya, lo); yy <= min(yb, H - 1 - lo); ++yy)
                for (int xx = max(xa, lo); xx <= min(xb, W - 1 - lo); ++xx) acc += g[(size_t)yy * W + xx];
        }
    }
    d_feat[i] += acc / (float)(half * half);
}

// gimg: optional scratch [B,128,H,W]; with it the unclamped boxes take the gather path.
// aux: optional scratch of B*H*W ints behind it (the rays-per-pixel table: pixels named by one ray
// are stored, not added); with it the clamped boxes are parked as well and go through the ring kernel.
extern "C" hipError_t lidf_launch_rayfeat_backward(const float* d_rayfeat, int ld_rf,
                                                   const int* ray_pix, const int* ray_bid,
                                                   long long R, int half, int B, int H, int W,
                                                   float* d_feat, float* gimg, int* aux,
                                                   hipStream_t st) {
    if (R <= 0) return hipSuccess;
    if (half <= 0) gimg = nullptr;
    // the ring kernel wants tiles of 16 pixels to reach the clamped boxes of one image side only
    // (also when the last, partial tile row / column is narrower than the ring)
    const int tx_n = (W + 15) / 16, ty_n = (H + 15) / 16;
    if (!gimg || half > 16 || W < 2 * half + 1 || H < 2 * half + 1 || W - 16 * (tx_n - 1) < half ||
        H - 16 * (ty_n - 1) < half)
        aux = nullptr;
    if (gimg) {
        hipError_t e = hipMemsetAsync(gimg, 0, (size_t)B * 128 * H * W * 4, st);
        if (e != hipSuccess) return e;
    }
    if (aux) {
        hipError_t e = hipMemsetAsync(aux, 0, (size_t)B * H * W * 4, st);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(lidf_rayfeat_count_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st,
                           ray_pix, ray_bid, R, H, W, aux);
    }
    hipLaunchKernelGGL(lidf_rayfeat_backward_kernel, dim3((unsigned)((R + 63) / 64)), dim3(256), 0,
                       st, d_rayfeat, ld_rf, ray_pix, ray_bid, R, half, H, W, d_feat, gimg, aux,
                       aux ? 1 : 0);
    if (gimg) {
        const long long total = (long long)B * 32 * H * W;
        hipLaunchKernelGGL(lidf_rayfeat_gather_kernel, dim3((unsigned)((total + 255) / 256)),
                           dim3(256), 0, st, gimg, B, H, W, half, aux ? half : 0, d_feat);
    }
    if (aux) {
        const int side = ty_n > 2 ? (tx_n > 1 ? 2 : 1) * (ty_n - 2) : 0;
        const int tiles = (ty_n > 1 ? 2 * tx_n : tx_n) + side;
        const int PW = 18 + 2 * half;
        hipLaunchKernelGGL(lidf_rayfeat_backward_ring_kernel, dim3(tiles, B * 32), dim3(256),
                           (size_t)PW * PW * 4, st, gimg, half, H, W, d_feat);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Factorised training path of the query (the layer-1 rewrite of the inference kernel, DESIGN §2,
// carried through the backward): layer 1 = W1[:, enter|leave] PE(p) + voxpart[voxel] + raypart[ray]
// (+ u*off), so only the positional encodings are per-pair rows.
// ------------------------------------------------------------------------------------------------
// pe[p] = [ embed(enter) | embed(leave) ]  (2*(3+6L) columns)
// Thread (pair, end, axis): ONE coordinate value — its index -> direction loads, its product and its revolution
// reduction happen once — and all of its 1 + 2 L row entries (x, sin / cos at every octave). Round 6: a thread per
// (pair, column) repeated the loads and the reduction for each of the 102 columns and threw one of every
// sin / cos pair away (128 threads per pair, 120 us per 614,400 pairs; now 8 threads per pair, 6 of them at work).
// sin/cos as in the inference kernel (revolutions, v_sin_f32 / v_cos_f32, |err| <= 4.2e-7 at every octave —
// lidf_device.h): the values are bit-identical to the per-column form's.
__global__ void __launch_bounds__(256) lidf_pe_rows_kernel(
    const int* __restrict__ pair_ray, const int* __restrict__ pair_vox, const float* __restrict__ pair_t,
    const float* __restrict__ ray_dir, const float* __restrict__ vox_center, int pos_rel, int L,
    long long P, float* __restrict__ pe) {
    // the 32 rows of a workgroup are one contiguous run of memory: built in LDS (a lane's entries lie 3 floats apart),
    // written out as whole 16-byte pieces by all 256 threads (the direct 4-byte stores ran at 2.2 TB/s)
    __shared__ __attribute__((aligned(16))) float s_rows[32 * 2 * (3 + 6 * 16)];
    const int E = 3 + 6 * L, W = 2 * E;
    const long long p0 = (long long)blockIdx.x * 32;
    const long long p = p0 + (threadIdx.x >> 3);
    const int q = threadIdx.x & 7;          // 0..2: enter x y z, 3..5: leave x y z, 6..7: idle
    if (p < P && q < 6) {
        const int end = q >= 3 ? 1 : 0, c = q - 3 * end;
        float x = __fmul_rn(ray_dir[3 * (size_t)pair_ray[p] + c], pair_t[2 * p + end]);
        if (pos_rel) x -= vox_center[3 * (size_t)pair_vox[p] + c];
        float* row = s_rows + (threadIdx.x >> 3) * W + end * E;
        row[c] = x;
        const Rev r = to_rev(x);
        for (int o = 0; o < L; ++o) {
            float sn, cs;
            rev_sincos(r, (float)(1 << o), sn, cs);
            row[3 + 6 * o + c] = sn;
            row[6 + 6 * o + c] = cs;
        }
    }
    __syncthreads();
    const long long rows = P - p0 < 32 ? P - p0 : 32;
    const int total = (int)rows * W;                 // floats of this workgroup's run; the run starts 16-byte aligned
    float* dst = pe + (size_t)p0 * W;                // (32 W floats per workgroup, W even: 32 * W * 4 bytes = 0 mod 16)
    for (int i = 4 * threadIdx.x; i + 3 < total; i += 1024) *(f32x4*)(dst + i) = *(const f32x4*)(s_rows + i);
    if (threadIdx.x < (total & 3)) dst[(total & ~3) + threadIdx.x] = s_rows[(total & ~3) + threadIdx.x];
}

// out[r, :] = sum over the ray's contiguous pairs of S[p, :]   (one wavefront per ray, no atomics)
__global__ void __launch_bounds__(256) lidf_seg_sum_ray_kernel(const float* __restrict__ S, int F,
                                                               const int* __restrict__ pair_off,
                                                               long long R, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int beg = pair_off[r], end = pair_off[r + 1];
    for (int j = lane * 4; j < F; j += 256) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int p = beg; p < end; ++p) {
            const f32x4 v = *(const f32x4*)(S + (size_t)p * F + j);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] += v[i];
        }
        *(f32x4*)(out + (size_t)r * F + j) = acc;
    }
}

// out[idx[p], :] += S[p, :]: a wavefront takes 64 consecutive pairs; each lane owns 4 of the F = 256
// columns and walks the pairs, flushing its running sum with atomics whenever the index changes
// (ray-major pairs change voxel every few pairs, so few flushes; summation order not fixed).
__global__ void __launch_bounds__(256) lidf_seg_sum_idx_kernel(const float* __restrict__ S,
                                                               const int* __restrict__ idx,
                                                               long long P, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long p0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (p0 >= P) return;
    const long long p1 = p0 + 64 < P ? p0 + 64 : P;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int cur = idx[p0];
    for (long long p = p0; p < p1; ++p) {
        const int v = idx[p];
        if (v != cur) {
#pragma unroll
            for (int i = 0; i < 4; ++i) atomicAdd(out + (size_t)cur * 256 + 4 * lane + i, acc[i]);
            acc = f32x4{0.f, 0.f, 0.f, 0.f};
            cur = v;
        }
        const f32x4 s = *(const f32x4*)(S + (size_t)p * 256 + 4 * lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += s[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) atomicAdd(out + (size_t)cur * 256 + 4 * lane + i, acc[i]);
}

extern "C" hipError_t lidf_launch_pe_rows(const int* pair_ray, const int* pair_vox,
                                          const float* pair_t, const float* ray_dir,
                                          const float* vox_center, int pos_rel, int L, long long P,
                                          float* pe, hipStream_t st) {
    if (P <= 0) return hipSuccess;
    if (L < 0 || L > 16) return hipErrorInvalidValue;
    hipLaunchKernelGGL(lidf_pe_rows_kernel, dim3((unsigned)((P + 31) / 32)), dim3(256), 0, st,
                       pair_ray, pair_vox, pair_t, ray_dir, vox_center, pos_rel, L, P, pe);
    return hipGetLastError();
}
extern "C" hipError_t lidf_launch_seg_sum_ray(const float* S, int F, const int* pair_off,
                                              long long R, float* out, hipStream_t st) {
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_seg_sum_ray_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, S, F,
                       pair_off, R, out);
    return hipGetLastError();
}
// The deterministic form used when the table has at most 16,384 rows: a stable counting sort of the
// pairs by their index (per-block histograms -> one exclusive scan over [index][block] -> a
// single-wavefront placement pass per block that keeps the pair order inside every index), then
// every index's rows are summed in chunks of SEG_CH gathered rows (a row = 1 KiB contiguous: the
// column-sliced LDS-atomic form this replaces read 64-byte pieces 1 KiB apart and ran at 0.7 TB/s)
// and the chunk sums are added up per index in order. out[v,:] is written (zero for an index
// without pairs), no atomics, no pre-zeroing.
#define SEG_CH 128
extern "C" hipError_t lidf_launch_scan(const int*, long long, int*, int*, hipStream_t);

struct SegPlan {
    long long nblk, per_blk, nscan, max_chunks;
    size_t hist, scanned, sums, perm, cnt, first, sums2, partial, total;
};
static SegPlan seg_plan(long long P, long long V) {
    SegPlan s;
    s.nblk = (P + 2047) / 2048;
    if (s.nblk > 1024) s.nblk = 1024;
    if (s.nblk < 1) s.nblk = 1;
    s.per_blk = ((P + s.nblk - 1) / s.nblk + 63) / 64 * 64;
    if (s.per_blk < 64) s.per_blk = 64;
    s.nblk = (P + s.per_blk - 1) / s.per_blk;
    if (s.nblk < 1) s.nblk = 1;
    s.nscan = V * s.nblk;
    s.max_chunks = P / SEG_CH + V + 1;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    size_t o = 0;
    s.hist = o;    o += up((size_t)s.nscan * 4);
    s.scanned = o; o += up((size_t)(s.nscan + 1) * 4);
    s.sums = o;    o += up((size_t)(s.nscan / 1024 + 2) * 4);
    s.perm = o;    o += up((size_t)(P > 0 ? P : 1) * 4);
    s.cnt = o;     o += up((size_t)(V + 1) * 4);
    s.first = o;   o += up((size_t)(V + 2) * 4);
    s.sums2 = o;   o += up((size_t)(V / 1024 + 2) * 4);
    s.partial = o; o += up((size_t)s.max_chunks * 256 * 4);
    s.total = o;
    return s;
}
extern "C" size_t lidf_seg_sum_idx_ws_bytes(long long P, long long V) {
    if (V > 16384 || V <= 0) return 0;
    return seg_plan(P, V).total;
}

__global__ void __launch_bounds__(256) lidf_seg_hist_kernel(const int* __restrict__ idx, long long P,
                                                            int V, long long per_blk, int nblk,
                                                            int* __restrict__ histT,
                                                            const int* __restrict__ n_dev) {
    extern __shared__ int s_cnt[];
    if (n_dev) P = *n_dev;   // device-side length (P = the capacity the blocks were cut for)
    for (int i = threadIdx.x; i < V; i += 256) s_cnt[i] = 0;
    __syncthreads();
    const long long p0 = blockIdx.x * per_blk;
    const long long p1 = p0 + per_blk < P ? p0 + per_blk : P;
    for (long long p = p0 + threadIdx.x; p < p1; p += 256) {
        const int v = idx[p];
        if (v >= 0 && v < V) atomicAdd(s_cnt + v, 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < V; i += 256) histT[(size_t)i * nblk + blockIdx.x] = s_cnt[i];
}

// one wavefront per block of pairs: pairs in order, 64 at a time; lanes with the same index find
// each other through `nbits` ballots, rank = earlier lanes of the group, the group's first lane
// advances the index's cursor
__global__ void __launch_bounds__(64) lidf_seg_place_kernel(const int* __restrict__ idx, long long P,
                                                            int V, int nbits, long long per_blk,
                                                            int nblk, const int* __restrict__ scanned,
                                                            int* __restrict__ perm,
                                                            const int* __restrict__ n_dev) {
    extern __shared__ int s_cur[];
    if (n_dev) P = *n_dev;
    for (int i = threadIdx.x; i < V; i += 64) s_cur[i] = scanned[(size_t)i * nblk + blockIdx.x];
    __syncthreads();
    const long long p0 = blockIdx.x * per_blk;
    const long long p1 = p0 + per_blk < P ? p0 + per_blk : P;
    const int lane = threadIdx.x;
    for (long long b = p0; b < p1; b += 64) {
        const long long p = b + lane;
        int v = p < p1 ? idx[p] : -1;
        const bool ok = v >= 0 && v < V;
        if (!ok) v = 0;
        unsigned long long m = __ballot(ok);
        for (int bit = 0; bit < nbits; ++bit) {
            const bool one = (v >> bit) & 1;
            const unsigned long long bb = __ballot(ok && one);
            m &= one ? bb : ~bb;
        }
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        int base = 0;
        if (ok) base = s_cur[v];
        __syncthreads();
        if (ok) {
            perm[base + rank] = (int)p;
            if (rank == 0) s_cur[v] = base + __popcll(m);
        }
        __syncthreads();
    }
}

__global__ void lidf_seg_chunks_kernel(const int* __restrict__ scanned, int V, int nblk, long long P,
                                       int* __restrict__ cnt) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const long long beg = scanned[(size_t)v * nblk];
    const long long end = scanned[(size_t)(v + 1) * nblk];   // v = V-1: the total
    cnt[v] = (int)((end - beg + SEG_CH - 1) / SEG_CH);
}

__global__ void __launch_bounds__(256) lidf_seg_chunk_sum_kernel(
    const float* __restrict__ S, const int* __restrict__ perm, const int* __restrict__ scanned,
    const int* __restrict__ first, int V, int nblk, float* __restrict__ partial) {
    __shared__ f32x4 red[4][64];
    const int c = blockIdx.x;
    if (c >= first[V]) return;
    // last index with first[v] <= c (indices without pairs share their successor's value)
    int lo = 0, hi = V - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (first[mid] <= c) lo = mid; else hi = mid - 1;
    }
    const int v = lo;
    const long long beg = (long long)scanned[(size_t)v * nblk] + (long long)(c - first[v]) * SEG_CH;
    long long end = scanned[(size_t)(v + 1) * nblk];
    if (end > beg + SEG_CH) end = beg + SEG_CH;
    const int rl = threadIdx.x >> 6, c4 = threadIdx.x & 63;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long long i = beg + rl; i < end; i += 16) {
        f32x4 x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long ii = i + 4 * j;
            x[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ii < end) x[j] = *(const f32x4*)(S + (size_t)perm[ii] * 256 + 4 * c4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += x[j];
    }
    red[rl][c4] = acc;
    __syncthreads();
    if (rl == 0) {
        f32x4 s = red[0][c4];
#pragma unroll
        for (int k = 1; k < 4; ++k) s += red[k][c4];
        *(f32x4*)(partial + (size_t)c * 256 + 4 * c4) = s;
    }
}

__global__ void __launch_bounds__(64) lidf_seg_final_kernel(const float* __restrict__ partial,
                                                            const int* __restrict__ first,
                                                            float* __restrict__ out) {
    const int v = blockIdx.x, c4 = threadIdx.x;
    const int c0 = first[v], c1 = first[v + 1];
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int c = c0;
    for (; c + 4 <= c1; c += 4) {
        f32x4 x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = *(const f32x4*)(partial + (size_t)(c + j) * 256 + 4 * c4);
#pragma unroll
        for (int j = 0; j < 4; ++j) s += x[j];
    }
    for (; c < c1; ++c) s += *(const f32x4*)(partial + (size_t)c * 256 + 4 * c4);
    *(f32x4*)(out + (size_t)v * 256 + 4 * c4) = s;
}

// out[v, :] = sum of S[p, :] over the pairs with idx[p] == v (F = 256). With a workspace of
// lidf_seg_sum_idx_ws_bytes(P, V) bytes: the deterministic sorted form; without (V > 16,384): `out`
// is zeroed and the run-walking kernel adds with atomics.
extern "C" hipError_t lidf_launch_seg_sum_idx(const float* S, const int* idx, long long P, long long V,
                                              float* out, void* ws, size_t ws_bytes, hipStream_t st) {
    if (V <= 0) return hipSuccess;
    const size_t need = lidf_seg_sum_idx_ws_bytes(P, V);
    if (P <= 0 || need == 0 || !ws || ws_bytes < need) {
        hipError_t e = hipMemsetAsync(out, 0, (size_t)V * 256 * 4, st);
        if (e != hipSuccess || P <= 0) return e;
        hipLaunchKernelGGL(lidf_seg_sum_idx_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0,
                           st, S, idx, P, out);
        return hipGetLastError();
    }
    const SegPlan s = seg_plan(P, V);
    char* w = (char*)ws;
    int* hist = (int*)(w + s.hist);
    int* scanned = (int*)(w + s.scanned);
    int* perm = (int*)(w + s.perm);
    int* cnt = (int*)(w + s.cnt);
    int* first = (int*)(w + s.first);
    float* partial = (float*)(w + s.partial);
    int nbits = 0;
    while ((1LL << nbits) < V) ++nbits;
    hipLaunchKernelGGL(lidf_seg_hist_kernel, dim3((unsigned)s.nblk), dim3(256), (size_t)V * 4, st, idx,
                       P, (int)V, s.per_blk, (int)s.nblk, hist, (const int*)nullptr);
    hipError_t e = lidf_launch_scan(hist, s.nscan, scanned, (int*)(w + s.sums), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(lidf_seg_place_kernel, dim3((unsigned)s.nblk), dim3(64), (size_t)V * 4, st, idx,
                       P, (int)V, nbits, s.per_blk, (int)s.nblk, scanned, perm, (const int*)nullptr);
    hipLaunchKernelGGL(lidf_seg_chunks_kernel, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st,
                       scanned, (int)V, (int)s.nblk, P, cnt);
    e = lidf_launch_scan(cnt, V, first, (int*)(w + s.sums2), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(lidf_seg_chunk_sum_kernel, dim3((unsigned)s.max_chunks), dim3(256), 0, st, S,
                       perm, scanned, first, (int)V, (int)s.nblk, partial);
    hipLaunchKernelGGL(lidf_seg_final_kernel, dim3((unsigned)V), dim3(64), 0, st, partial, first, out);
    return hipGetLastError();
}

// The counting sort above on its own: perm[0 .. n_perm) = the items whose index lies in [0, V), grouped by
// index, order kept inside a group (the PointNet's voxel-sorted walk over large tables, lidf_pointnet.hip).
// No global atomics: per-block LDS histograms, a scan over [index][block], a placement pass.
// V <= 16,384 (the histogram of a block lives in LDS). n_dev (optional): the item count on the device, P
// then bounds the launches. ws: lidf_sort_idx_ws_bytes(P, V) bytes; perm / n_perm point into it.
extern "C" size_t lidf_sort_idx_ws_bytes(long long P, long long V) {
    if (V > 16384 || V <= 0) return 0;
    const SegPlan s = seg_plan(P, V);
    return s.cnt;   // hist | scanned | sums | perm
}
// where lidf_launch_sort_idx leaves its scan inside `ws`: scanned[v * nblk] = first sorted position of index v
// (the PointNet's training chains turn it into per-voxel row ranges of their sorted buffers)
extern "C" void lidf_sort_idx_layout(long long P, long long V, size_t* scanned_off, int* nblk, size_t* perm_off) {
    const SegPlan s = seg_plan(P, V);
    *scanned_off = s.scanned;
    *nblk = (int)s.nblk;
    *perm_off = s.perm;
}
extern "C" hipError_t lidf_launch_sort_idx(const int* idx, long long P, const int* n_dev, long long V,
                                           void* ws, const int** perm_out, const int** n_perm_out,
                                           hipStream_t st) {
    if (V <= 0 || V > 16384 || P <= 0 || !ws) return hipErrorInvalidValue;
    const SegPlan s = seg_plan(P, V);
    char* w = (char*)ws;
    int* hist = (int*)(w + s.hist);
    int* scanned = (int*)(w + s.scanned);
    int* perm = (int*)(w + s.perm);
    int nbits = 0;
    while ((1LL << nbits) < V) ++nbits;
    hipLaunchKernelGGL(lidf_seg_hist_kernel, dim3((unsigned)s.nblk), dim3(256), (size_t)V * 4, st, idx,
                       P, (int)V, s.per_blk, (int)s.nblk, hist, n_dev);
    hipError_t e = lidf_launch_scan(hist, s.nscan, scanned, (int*)(w + s.sums), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(lidf_seg_place_kernel, dim3((unsigned)s.nblk), dim3(64), (size_t)V * 4, st, idx,
                       P, (int)V, nbits, s.per_blk, (int)s.nblk, scanned, perm, n_dev);
    *perm_out = perm;
    *n_perm_out = scanned + s.nscan;   // the scan's grand total = number of placed items
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Per-pair tail of LIDF.get_pred and its adjoint (models/pipeline.py:437-439, :452-454):
//   s = (off (r1-r0) + r0) sqrt(3) part_size ;  pair_pred_pos = dir t_enter + s dir
//   pred_pos[r] = pair_pred_pos[max_pair_id[r]]   (dummy row (0,0,0) for a ray without pairs)
// The per-ray softmax / arg-max is not differentiated (the reference detaches the logits, :442).
// Backward: d off[p] = k dir[ray(p)] . (g_pair_pred_pos[p] + [p == max_pair_id[ray(p)]] g_pred_pos[ray(p)]),
// k = (r1-r0) sqrt(3) part_size — one thread per pair, no atomics (a pair knows whether it is its
// ray's selected one).
// ------------------------------------------------------------------------------------------------
__global__ void lidf_pair_pos_kernel(const float* __restrict__ off, const int* __restrict__ pair_ray,
                                     const float* __restrict__ pair_t,
                                     const float* __restrict__ ray_dir, long long P, float r0,
                                     float rs, float sqrt3, float part, float* __restrict__ out) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const size_t r = (size_t)pair_ray[p];
    const float dx = ray_dir[3 * r], dy = ray_dir[3 * r + 1], dz = ray_dir[3 * r + 2];
    const float te = pair_t[2 * p];
    float s = __fadd_rn(__fmul_rn(off[p], rs), r0);
    s = __fmul_rn(__fmul_rn(s, sqrt3), part);
    out[3 * p + 0] = __fadd_rn(__fmul_rn(dx, te), __fmul_rn(s, dx));
    out[3 * p + 1] = __fadd_rn(__fmul_rn(dy, te), __fmul_rn(s, dy));
    out[3 * p + 2] = __fadd_rn(__fmul_rn(dz, te), __fmul_rn(s, dz));
}

__global__ void lidf_ray_select_kernel(const float* __restrict__ pos, const long long* __restrict__ id,
                                       long long R, long long P, float* __restrict__ pred_pos) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const long long m = id[r];
    const bool ok = m >= 0 && m < P;
    pred_pos[3 * r + 0] = ok ? pos[3 * m + 0] : 0.f;
    pred_pos[3 * r + 1] = ok ? pos[3 * m + 1] : 0.f;
    pred_pos[3 * r + 2] = ok ? pos[3 * m + 2] : 0.f;
}

__global__ void lidf_pair_pos_backward_kernel(const float* __restrict__ g_pos,
                                              const float* __restrict__ g_pred,
                                              const long long* __restrict__ id,
                                              const int* __restrict__ pair_ray,
                                              const float* __restrict__ ray_dir, long long P,
                                              float k, float* __restrict__ d_off) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const size_t r = (size_t)pair_ray[p];
    float gx = 0.f, gy = 0.f, gz = 0.f;
    if (g_pos) {
        gx = g_pos[3 * p];
        gy = g_pos[3 * p + 1];
        gz = g_pos[3 * p + 2];
    }
    if (g_pred && id[r] == p) {
        gx += g_pred[3 * r];
        gy += g_pred[3 * r + 1];
        gz += g_pred[3 * r + 2];
    }
    d_off[p] = k * (ray_dir[3 * r] * gx + ray_dir[3 * r + 1] * gy + ray_dir[3 * r + 2] * gz);
}

extern "C" hipError_t lidf_launch_pair_pos(const float* off, const int* pair_ray, const float* pair_t,
                                           const float* ray_dir, long long P, float r0, float rs,
                                           float sqrt3, float part, float* out, hipStream_t st) {
    if (P <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_pair_pos_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, off,
                       pair_ray, pair_t, ray_dir, P, r0, rs, sqrt3, part, out);
    return hipGetLastError();
}
extern "C" hipError_t lidf_launch_ray_select(const float* pos, const long long* id, long long R,
                                             long long P, float* pred_pos, hipStream_t st) {
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_ray_select_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, pos,
                       id, R, P, pred_pos);
    return hipGetLastError();
}
extern "C" hipError_t lidf_launch_pair_pos_backward(const float* g_pos, const float* g_pred,
                                                    const long long* id, const int* pair_ray,
                                                    const float* ray_dir, long long P, float k,
                                                    float* d_off, hipStream_t st) {
    if (P <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_pair_pos_backward_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0,
                       st, g_pos, g_pred, id, pair_ray, ray_dir, P, k, d_off);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// The offset decoder's backward from the selected pairs only (lidf_query_decoder_backward_rows_f32). The
// reference's losses reach offset_dec through pred_pos = pair_pred_pos[max_pair_id] alone
// (models/pipeline.py:437-454, 468-476): dL/d pred_offset is k (ray_dir[r] . g_pred_pos[r]) at the selected pair
// of ray r and exactly zero at every other pair, and a row with a zero output gradient adds exactly zero to
// every sum of the backward. This kernel forms the one-pair-per-ray problem: row r = the kept activations
// (every plane of every pass), the pre-activation, the voxel and the position-embedding row of pair rows[r],
// and its output gradient; a ray without a pair (rows[r] >= P) takes pair 0's rows with gradient 0.
// 128 threads per row: float4 c of the 112 of H1 | H2 | H3, then offset-in, sign words (2 + 1 float4).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) lidf_gather_sel_rows_kernel(
    const float* __restrict__ act, long long P, int npass, const long long* __restrict__ rows, long long R,
    const int* __restrict__ pair_vox, const float* __restrict__ pe, int E2, const float* __restrict__ g_pred,
    const float* __restrict__ g_extra, const float* __restrict__ ray_dir, float k, float* __restrict__ act_dst, int* __restrict__ pvox,
    float* __restrict__ pe_dst, float* __restrict__ g_dst, int* __restrict__ poff) {
    const long long r = (long long)blockIdx.x * 2 + (threadIdx.x >> 7);
    const int c = threadIdx.x & 127;
    if (r >= R) return;
    const long long id = rows[r];
    const bool has = id >= 0 && id < P;
    const size_t p = has ? (size_t)id : 0;
    for (int ps = 0; act && ps < npass; ++ps) {   // (act == NULL: the activations are those of the list already)
        const float* src = act + (size_t)ps * P * LIDF_ACT_ROW_FLOATS;
        float* dst = act_dst + (size_t)ps * R * LIDF_ACT_ROW_FLOATS;
        if (c < 64) {
            *(f32x4u*)(dst + (size_t)r * LIDF_H1 + 4 * c) = *(const f32x4u*)(src + p * LIDF_H1 + 4 * c);
        } else if (c < 96) {
            const int j = c - 64;
            *(f32x4u*)(dst + (size_t)R * LIDF_H1 + (size_t)r * LIDF_H2 + 4 * j) =
                *(const f32x4u*)(src + (size_t)P * LIDF_H1 + p * LIDF_H2 + 4 * j);
        } else if (c < 112) {
            const int j = c - 96;
            *(f32x4u*)(dst + (size_t)R * (LIDF_H1 + LIDF_H2) + (size_t)r * LIDF_H3 + 4 * j) =
                *(const f32x4u*)(src + (size_t)P * (LIDF_H1 + LIDF_H2) + p * LIDF_H3 + 4 * j);
        } else if (c == 112) {
            dst[(size_t)R * LIDF_ACT_OIN + r] = src[(size_t)P * LIDF_ACT_OIN + p];
        } else if (c < 115) {
            const int j = c - 113;
            *(f32x4u*)(dst + (size_t)R * LIDF_ACT_M1 + (size_t)r * 8 + 4 * j) =
                *(const f32x4u*)(src + (size_t)P * LIDF_ACT_M1 + p * 8 + 4 * j);
        } else if (c == 115) {
            *(f32x4u*)(dst + (size_t)R * LIDF_ACT_M2 + (size_t)r * 4) = *(const f32x4u*)(src + (size_t)P * LIDF_ACT_M2 + p * 4);
        }
    }
    // the pre-activation of the last pass sits behind the passes
    if (c == 116 && act)
        act_dst[(size_t)npass * R * LIDF_ACT_ROW_FLOATS + r] = act[(size_t)npass * P * LIDF_ACT_ROW_FLOATS + p];
    if (c == 117) {
        pvox[r] = pair_vox[p];
        poff[r] = (int)r;
        if (r == R - 1) poff[R] = (int)R;
        // (the adjoint of pred_pos = enter + k off dir, as lidf_pair_pos_backward_kernel forms it)
        float gv = 0.f;
        if (has && g_pred)
            gv = k * (ray_dir[3 * r] * g_pred[3 * r] + ray_dir[3 * r + 1] * g_pred[3 * r + 1] +
                      ray_dir[3 * r + 2] * g_pred[3 * r + 2]);
        if (has && g_extra) gv += g_extra[r];
        g_dst[r] = gv;
    }
    for (int j = c; j < E2; j += 128) pe_dst[(size_t)r * E2 + j] = pe[p * E2 + j];
}
extern "C" hipError_t lidf_launch_gather_sel_rows(const float* act, long long P, int npass, const long long* rows,
                                                  long long R, const int* pair_vox, const float* pe, int E2,
                                                  const float* g_pred, const float* g_extra, const float* ray_dir,
                                                  float k, float* act_dst, int* pvox, float* pe_dst, float* g_dst,
                                                  int* poff, hipStream_t st) {
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_gather_sel_rows_kernel, dim3((unsigned)((R + 1) / 2)), dim3(256), 0, st, act, P, npass,
                       rows, R, pair_vox, pe, E2, g_pred, g_extra, ray_dir, k, act_dst, pvox, pe_dst, g_dst, poff);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// PointNet2Stage training pieces (models/pointnet.py:22-38 under autograd):
//   relu mask, arg of the per-voxel max (torch_scatter's scatter-max sends the gradient of a pooled
//   entry to ONE source row: here the lowest row index attaining the maximum), its backward, and
//   the per-voxel sum of rows (adjoint of the gather g1[vox]).
// ------------------------------------------------------------------------------------------------
__global__ void lidf_relu_mask_kernel(const float* __restrict__ g, const float* __restrict__ src,
                                      long long n, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[i] > 0.f ? g[i] : 0.f;
}

__global__ void lidf_segmax_arg_kernel(const float* __restrict__ f, const int* __restrict__ vox,
                                       const float* __restrict__ pool, long long N, int F,
                                       int* __restrict__ arg) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * F) return;
    const long long n = i / F;
    const int c = (int)(i % F), v = vox[n];
    if (v < 0) return;
    if (f[i] == pool[(size_t)v * F + c]) atomicMin(arg + (size_t)v * F + c, (int)n);
}

// out[n, c] (+)= dp[vox[n], c] if row n is the arg of (vox[n], c) and the pooled value is positive
// (a pooled 0 came from the zero fill or from a relu output at its kink: no gradient either way)
__global__ void lidf_segmax_backward_kernel(const float* __restrict__ dp, const int* __restrict__ arg,
                                            const int* __restrict__ vox,
                                            const float* __restrict__ pool, long long N, int F,
                                            int accumulate, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * F) return;
    const long long n = i / F;
    const int c = (int)(i % F), v = vox[n];
    float g = 0.f;
    if (v >= 0) {
        const size_t k = (size_t)v * F + c;
        if (arg[k] == (int)n && pool[k] > 0.f) g = dp[k];
    }
    out[i] = accumulate ? out[i] + g : g;
}

__global__ void lidf_seg_sum_rows_kernel(const float* __restrict__ S, const int* __restrict__ idx,
                                         long long N, int F, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * (F / 4)) return;
    const long long n = i / (F / 4);
    const int c = (int)(i % (F / 4)) * 4, v = idx[n];
    if (v < 0) return;
    const f32x4 s = *(const f32x4*)(S + (size_t)n * F + c);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (s[k] != 0.f) atomicAdd(out + (size_t)v * F + c + k, s[k]);
}

extern "C" hipError_t lidf_launch_relu_mask(const float* g, const float* src, long long n, float* out,
                                            hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_relu_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, src,
                       n, out);
    return hipGetLastError();
}
extern "C" hipError_t lidf_launch_segmax_arg(const float* f, const int* vox, const float* pool,
                                             long long N, int F, int* arg, hipStream_t st) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_segmax_arg_kernel, dim3((unsigned)((N * F + 255) / 256)), dim3(256), 0, st, f,
                       vox, pool, N, F, arg);
    return hipGetLastError();
}
extern "C" hipError_t lidf_launch_segmax_backward(const float* dp, const int* arg, const int* vox,
                                                  const float* pool, long long N, int F,
                                                  int accumulate, float* out, hipStream_t st) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_segmax_backward_kernel, dim3((unsigned)((N * F + 255) / 256)), dim3(256), 0,
                       st, dp, arg, vox, pool, N, F, accumulate, out);
    return hipGetLastError();
}
extern "C" hipError_t lidf_launch_seg_sum_rows(const float* S, const int* idx, long long N, int F,
                                               float* out, hipStream_t st) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_seg_sum_rows_kernel, dim3((unsigned)((N * (F / 4) + 255) / 256)), dim3(256),
                       0, st, S, idx, N, F, out);
    return hipGetLastError();
}

// Positional-encoding backward (models/implicit_net.py:9-39 under autograd):
//   d x[i,c] = g[i,c] + sum_o 2^o ( cos(2^o x) g_sin[o,c] - sin(2^o x) g_cos[o,c] )
__global__ void lidf_embed_backward_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                           long long n, int L, float* __restrict__ dx) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 3) return;
    const long long r = i / 3;
    const int c = (int)(i % 3);
    const float v = x[i];
    const float* gr = g + (size_t)r * (3 + 6 * L);
    float acc = gr[c];
    float f = 1.f;
    for (int o = 0; o < L; ++o) {
        const float a = v * f;
        acc += f * (cosf(a) * gr[3 + 6 * o + c] - sinf(a) * gr[3 + 6 * o + 3 + c]);
        f *= 2.f;
    }
    dx[i] = acc;
}
extern "C" hipError_t lidf_launch_embed_backward(const float* x, const float* g, long long n, int L,
                                                 float* dx, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_embed_backward_kernel, dim3((unsigned)((n * 3 + 255) / 256)), dim3(256), 0,
                       st, x, g, n, L, dx);
    return hipGetLastError();
}
