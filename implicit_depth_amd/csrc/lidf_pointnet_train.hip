// lidf_pointnet_train.hip — PointNet2Stage under autograd as register chains (gfx950 / CDNA4).
//
// models/pointnet.py:22-38 with torch_scatter's max pooling, forward with what the backward needs kept, and
// the backward itself. Rounds 2-4 ran the training forward layer by layer through lidf_linear_kernel (every
// [n,32] / [n,64] / [n,128] intermediate written and read back, the arg row of every pooled entry found by a
// second pass of one thread per (point, feature)) and the backward as seven transposed layer launches with
// torch-free but separate relu-mask / arg-scatter / row-sum launches in between: 3.9 ms of the 7.3 ms stage-2
// training step. Here:
//
//  * the points are walked GROUPED BY VOXEL (the stable counting sort of lidf_train.hip, once per forward) and
//    every per-point buffer the backward reads is stored in that sorted order: chains read and write
//    contiguous rows, a voxel's rows are one contiguous run (per-voxel row sums without atomics, in a fixed
//    order), and "the arg row of a pooled entry" is a sorted row index.
//  * forward = the two inference chains (lidf_pointnet.hip: same weight stream, same instruction sequence per
//    value) that also store f1 | f2 (stage 1) and f4 (stage 2) tiles as they complete and pool (value, row)
//    PAIRS: a 64-bit integer maximum of (value bits << 32 | ~row) is the maximum value and, among equal
//    values, the lowest row — torch_scatter's scatter-max semantics as rounds 2-4 pinned them — in one atomic,
//    through a windowed LDS table per workgroup flushed with one global 64-bit maximum per touched entry.
//  * backward phase A: dz5 is never stored — it is non-zero only at the arg rows, so a lane builds its
//    tile from the voxel's (dp2, arg2) rows; dz4 = (W4^T dz5) * relu'(f4) and df2 = W3[:, 64:]^T dz4 leave the
//    accumulators as the next product's B operands. dW4 / db4 are sums of at most V x 128 outer products
//    (lidf_pnet_dw4_kernel) instead of a [n,128]^T [n,128] product over all points.
//  * backward phase B: df2 + the pool1 arg scatter, relu masks, dz1 = W2^T dz2, d inp = W1^T dz1.
//  * per-voxel sums of dz4 over the voxel's contiguous rows, chunked (a voxel can hold thousands of points).
// The dense weight gradients (dW3[:, 64:], dW2, dW1 over the points; the per-voxel layers over V rows) stay
// launches of lidf_wgrad2_kernel: their contraction runs over the points, which the chains hold in the lane
// dimension.
#include "lidf_device.h"

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define LDQ(rs, voff, soff) \
    __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128((rs), (voff), (soff), 0))

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

#define PNT_WINDOW 32   // rows of the workgroup's pooling window (sorted points: ascending voxels)

static_assert(PNB_A_QUADS % LIDF_RING == 0 && PNB_B_QUADS % LIDF_RING == 0, "ring phase");

struct PnTrainFwdArgs {
    const float* stream;   // the inference chains' stream (PN_S2_QUADS KiB)
    const float* inp;      // [n,6]
    const int* vox;        // [n]
    const int* perm;       // sorted position -> point
    const int* n_perm;     // device: number of sorted points
    const float* gpart;    // stage 2: [V,128] = W3[:, :64] g1 + b3
    float* inps;           // stage 1 writes [n,6], [n], [n,32], [n,64]; stage 2 writes f4s [n,128]
    int* voxs;
    float *f1s, *f2s, *f4s;
    u64* pool64;           // [V, F] (zeroed by the caller)
    int V;
    long long n;           // capacity of the launch
};

__device__ __forceinline__ void pnt_relu(f32x16& v) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = fmaxf(v[i], 0.f);
}

// one accumulator tile as 4 x 16 bytes of the row-major row `row` (width F): features 32 T + 8 g + 4 h + {0..3}
__device__ __forceinline__ void pnt_store_tile(const f32x16& a, float* base, long long row, int F, int T, int h,
                                               bool ok) {
    if (!ok) return;
    float* p = base + (size_t)row * F + 32 * T + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = {a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
        *(f32x4*)(p + 8 * g) = v;
    }
}
__device__ __forceinline__ void pnt_load_tile(f32x4 (&v)[4], const float* base, long long row, int F, int T, int h) {
    const float* p = base + (size_t)row * F + 32 * T + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) v[g] = *(const f32x4*)(p + 8 * g);
}

__device__ __forceinline__ void pnt_pool64(const f32x16& acc, int T, u64* row, int h, unsigned key, bool ok) {
    if (!ok) return;
    u64* p = row + 32 * T + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = acc[4 * g + i];
            if (v > 0.f) atomicMax(p + 8 * g + i, ((u64)__float_as_uint(v) << 32) | key);
        }
    }
}

template <int STAGE>
__global__ void __launch_bounds__(256, 2) lidf_pnet_train_fwd_kernel(PnTrainFwdArgs a) {
    constexpr int F = STAGE == 1 ? 64 : 128;
    constexpr int FP = F + 1;   // (row stride of the table in 8-byte words: rows start in different banks)
    constexpr int NQ = STAGE == 1 ? PN_S1_QUADS : PN_S2_QUADS;
    __shared__ u64 tab[PNT_WINDOW * FP];
    const long long AN = *a.n_perm;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, col = lane & 31;
    const float one_b = h ? 0.f : 1.f;
    const __amdgpu_buffer_rsrc_t srs =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.stream, 0, PN_S2_QUADS * 1024, 0x00020000);
    const int vq = lane * 16;
    // (tiles over the launch's capacity: rows [AN, n) of the sorted buffers — points left out of the pooling —
    // are written as zeros, so that the weight-gradient products over all n rows meet neither stale nor
    // non-finite memory)
    const long long ntile = (a.n + 127) / 128;
    const long long per = ntile / gridDim.x, rem = ntile % gridDim.x, bx = blockIdx.x;
    const long long tb = bx * per + (bx < rem ? bx : rem), te = tb + per + (bx < rem ? 1 : 0);
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (tb >= te) return;
    for (int i = threadIdx.x; i < PNT_WINDOW * FP; i += 256) tab[i] = 0ull;
    __syncthreads();
    f32x4 ring[LIDF_RING];
#pragma unroll
    for (int i = 0; i < LIDF_RING; ++i) ring[i] = LDQ(srs, vq, i * 1024);
    const int vbase = tb * 128 < AN ? a.vox[a.perm[tb * 128]] : 0;

    auto fetch = [&](long long tile, int& vox_o, float (&b)[4]) {
        const long long i = tile * 128 + wave * 32 + col;
        const long long pc = a.perm[i < AN ? i : AN - 1];
        vox_o = a.vox[pc];
        const float* x = a.inp + (size_t)pc * 6;
        if (h == 0) {
            b[0] = x[0]; b[1] = x[1]; b[2] = x[2]; b[3] = x[3];
        } else {
            b[0] = x[4]; b[1] = x[5]; b[2] = 1.f; b[3] = 0.f;
        }
    };
    int vox_n = 0;
    float b1n[4] = {0.f, 0.f, 0.f, 0.f};
    if (tb * 128 + wave * 32 < AN) fetch(tb, vox_n, b1n);

    for (long long tile = tb; tile < te; ++tile) {
        const long long i = tile * 128 + wave * 32 + col;   // sorted row of this lane's point
        if (tile * 128 + wave * 32 >= AN) {   // wave-uniform: no point left for this wavefront, zero rows only
            if (i < a.n) {
                if (STAGE == 1) {
                    float* xs = a.inps + (size_t)i * 6;
                    if (h == 0) { *(f32x2*)xs = f32x2{0.f, 0.f}; *(f32x2*)(xs + 2) = f32x2{0.f, 0.f}; a.voxs[i] = 0; }
                    else *(f32x2*)(xs + 4) = f32x2{0.f, 0.f};
                    pnt_store_tile(zero16, a.f1s, i, 32, 0, h, true);
                    pnt_store_tile(zero16, a.f2s, i, 64, 0, h, true);
                    pnt_store_tile(zero16, a.f2s, i, 64, 1, h, true);
                } else {
#pragma unroll
                    for (int T = 0; T < 4; ++T) pnt_store_tile(zero16, a.f4s, i, 128, T, h, true);
                }
            }
            continue;
        }
        const bool valid = i < AN;
        const bool inb = i < a.n;
        const int vox = vox_n;
        const float b1[4] = {b1n[0], b1n[1], b1n[2], b1n[3]};
        const int vrow = (vox >= vbase && vox - vbase < PNT_WINDOW) ? vox - vbase : -1;
        const unsigned key = 0xffffffffu - (unsigned)i;
        u64* prow = vrow >= 0 ? tab + vrow * FP : a.pool64 + (size_t)vox * F;   // (beyond the window: the global table)
        f32x16 F4[STAGE == 2 ? 4 : 1];
        if (STAGE == 2) {
            const float* gp = a.gpart + (size_t)vox * 128 + 4 * h;
#pragma unroll
            for (int T = 0; T < 4; ++T) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = *(const f32x4*)(gp + 32 * T + 8 * g);
#pragma unroll
                    for (int k = 0; k < 4; ++k) F4[STAGE == 2 ? T : 0][4 * g + k] = v[k];
                }
            }
        }
        const float live = valid ? 1.f : 0.f;   // (a dead lane of a live wavefront ran a clamped point: zero rows)
        if (STAGE == 1 && inb) {   // the sorted copies of the point's row and voxel
            float* xs = a.inps + (size_t)i * 6;
            if (h == 0) {
                *(f32x2*)xs = f32x2{b1[0] * live, b1[1] * live};
                *(f32x2*)(xs + 2) = f32x2{b1[2] * live, b1[3] * live};
                a.voxs[i] = valid ? vox : 0;
            } else {
                *(f32x2*)(xs + 4) = f32x2{b1[0] * live, b1[1] * live};
            }
        }
        if (tile + 1 < te && (tile + 1) * 128 + wave * 32 < AN) fetch(tile + 1, vox_n, b1n);
        SCHED_FENCE();

        f32x16 F1, F2[2], acc;
#pragma unroll
        for (int s = 0; s < NQ; ++s) {
            const f32x4 aq = ring[s % LIDF_RING];
            {
                const int nx = s + LIDF_RING;
                const int rel = nx < NQ ? nx : nx - NQ;
                ring[s % LIDF_RING] = LDQ(srs, vq + (rel & 3) * 1024, (rel >> 2) * 4096);
            }
            if (s < PN_P1) {
                F1 = MFMA(aq[0], b1[0], zero16);
                F1 = MFMA(aq[1], b1[1], F1);
                F1 = MFMA(aq[2], b1[2], F1);
                F1 = MFMA(aq[3], b1[3], F1);
                pnt_relu(F1);
                if (STAGE == 1) pnt_store_tile(valid ? F1 : zero16, a.f1s, i, 32, 0, h, inb);
            } else if (s < PN_P1 + PN_P2) {
                const int q = s - PN_P1, kq = q / 2, t = q % 2;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int k = 4 * kq + jj;
                    if (k < 16) F2[t] = MFMA(aq[jj], F1[k], k == 0 ? zero16 : F2[t]);
                    else if (k == 16) F2[t] = MFMA(aq[jj], one_b, F2[t]);
                }
                if (kq == 4) {
                    pnt_relu(F2[t]);
                    if (STAGE == 1) {
                        pnt_store_tile(valid ? F2[t] : zero16, a.f2s, i, 64, t, h, inb);
                        pnt_pool64(F2[t], t, prow, h, key, valid);
                    }
                }
            } else if (STAGE == 2 && s < PN_P1 + PN_P2 + PN_P3) {
                const int q = s - PN_P1 - PN_P2, T = q / 8, kq = q % 8;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int k = 4 * kq + jj;
                    F4[STAGE == 2 ? T : 0] = MFMA(aq[jj], F2[k / 16][k % 16], F4[STAGE == 2 ? T : 0]);
                }
                if (kq == 7) {
                    pnt_relu(F4[STAGE == 2 ? T : 0]);
                    pnt_store_tile(valid ? F4[STAGE == 2 ? T : 0] : zero16, a.f4s, i, 128, T, h, inb);
                }
            } else if (STAGE == 2 && s < PN_P1 + PN_P2 + PN_P3 + PN_P4) {
                const int q = s - PN_P1 - PN_P2 - PN_P3, T = q / 17, kq = q % 17;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int k = 4 * kq + jj;
                    if (k < 64) acc = MFMA(aq[jj], F4[STAGE == 2 ? k / 16 : 0][k % 16], k == 0 ? zero16 : acc);
                    else if (k == 64) acc = MFMA(aq[jj], one_b, acc);
                }
                if (kq == 16) {
                    pnt_relu(acc);
                    pnt_pool64(acc, T, prow, h, key, valid);
                }
            }
            SCHED_FENCE();
        }
    }
    // the window's rows into the global table: one 64-bit maximum per touched entry and workgroup
    __syncthreads();
    for (int e = threadIdx.x; e < PNT_WINDOW * F; e += 256) {
        const u64 v = tab[(e / F) * FP + e % F];
        const int row = vbase + e / F;
        if (v != 0ull && row < a.V) atomicMax(a.pool64 + (size_t)row * F + e % F, v);
    }
}

// pool64 -> the pooled table (floats) and the arg row of every entry (-1: no point raised it — the voxel has no
// point, or every value was 0, whose gradient the ReLU in front of the pooling stops anyway)
__global__ void lidf_pnet_unpack_kernel(const u64* __restrict__ p64, long long count, float* __restrict__ pool,
                                        int* __restrict__ arg) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    const u64 p = p64[e];
    pool[e] = __uint_as_float((unsigned)(p >> 32));
    arg[e] = p ? (int)(0xffffffffu - (unsigned)p) : -1;
}

// vstart[v] = first sorted row of voxel v (vstart[V] = number of sorted rows), first[v] = first chunk of voxel v
// in the list of CH-row chunks (first[V] = number of chunks). One workgroup.
__global__ void __launch_bounds__(1024) lidf_pnet_tables_kernel(const int* __restrict__ scanned, int nblk, int V,
                                                                const int* __restrict__ n_perm, int ch,
                                                                int* __restrict__ vstart, int* __restrict__ first) {
    __shared__ int s_w[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int total = *n_perm;
    int carry = 0;
    for (int b = 0; b < V; b += 1024) {
        const int v = b + threadIdx.x;
        int c = 0;
        if (v < V) {
            const int beg = scanned[(size_t)v * nblk];
            const int end = v + 1 < V ? scanned[(size_t)(v + 1) * nblk] : total;
            vstart[v] = beg;
            c = (end - beg + ch - 1) / ch;
        }
        int inc = c;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const int o = __shfl_up(inc, sft);
            if (lane >= sft) inc += o;
        }
        __syncthreads();
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        int wpre = 0, tot = 0;
        for (int w = 0; w < 16; ++w) {
            const int t = s_w[w];
            wpre += w < wave ? t : 0;
            tot += t;
        }
        if (v < V) first[v] = carry + wpre + inc - c;
        carry += tot;
    }
    if (threadIdx.x == 0) {
        vstart[V] = total;
        first[V] = carry;
    }
}

// ---- backward, phase A --------------------------------------------------------------------------------
struct PnBwdAArgs {
    const float* stream;       // PNB_A_QUADS KiB: W4^T (4 x 16 quads), W3[:, 64:]^T (2 x 16 quads)
    const int* voxs;           // [n] voxel of the sorted rows
    const int* n_perm;
    const float* dp2;          // [V,128] dL/d pool2
    const int* arg2;           // [V,128]
    const float* f4s;          // [n,128]
    float *dz4s, *df2s;        // [n,128], [n,64]
    long long n;
};

__global__ void __launch_bounds__(256, 2) lidf_pnet_bwd_a_kernel(PnBwdAArgs a) {
    const long long AN = *a.n_perm;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, col = lane & 31;
    const __amdgpu_buffer_rsrc_t srs =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.stream, 0, PNB_A_QUADS * 1024, 0x00020000);
    const int vq = lane * 16;
    const long long ntile = (a.n + 127) / 128;   // (over the capacity: rows [AN, n) are written as zeros)
    const long long per = ntile / gridDim.x, rem = ntile % gridDim.x, bx = blockIdx.x;
    const long long tb = bx * per + (bx < rem ? bx : rem), te = tb + per + (bx < rem ? 1 : 0);
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (tb >= te) return;
    f32x4 ring[LIDF_RING];
#pragma unroll
    for (int i = 0; i < LIDF_RING; ++i) ring[i] = LDQ(srs, vq, i * 1024);
    for (long long tile = tb; tile < te; ++tile) {
        const long long i = tile * 128 + wave * 32 + col;
        const bool inb = i < a.n;
        if (tile * 128 + wave * 32 >= AN) {   // wave-uniform: zero rows only
#pragma unroll
            for (int T = 0; T < 4; ++T) pnt_store_tile(zero16, a.dz4s, i, 128, T, h, inb);
            pnt_store_tile(zero16, a.df2s, i, 64, 0, h, inb);
            pnt_store_tile(zero16, a.df2s, i, 64, 1, h, inb);
            continue;
        }
        const bool valid = i < AN;
        const long long ic = valid ? i : AN - 1;
        const int vox = a.voxs[ic];
        // dz5 of this lane's point: dp2[v, f] where the point is the arg row of (v, f)
        f32x16 dz5[4];
        {
            const int* ap = a.arg2 + (size_t)vox * 128 + 4 * h;
            const float* dp = a.dp2 + (size_t)vox * 128 + 4 * h;
#pragma unroll
            for (int T = 0; T < 4; ++T) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const i32x4 ar = *(const i32x4*)(ap + 32 * T + 8 * g);
                    const f32x4 d = *(const f32x4*)(dp + 32 * T + 8 * g);
#pragma unroll
                    for (int k = 0; k < 4; ++k) dz5[T][4 * g + k] = (valid && ar[k] == (int)i) ? d[k] : 0.f;
                }
            }
        }
        SCHED_FENCE();
        f32x16 dz4[4], acc;
        f32x4 mk[4];
#pragma unroll
        for (int s = 0; s < PNB_A_QUADS; ++s) {
            const f32x4 aq = ring[s % LIDF_RING];
            {
                const int nx = s + LIDF_RING;
                const int rel = nx < PNB_A_QUADS ? nx : nx - PNB_A_QUADS;
                ring[s % LIDF_RING] = LDQ(srs, vq + (rel & 3) * 1024, (rel >> 2) * 4096);
            }
            if (s < 64) {   // dz4 = (W4^T dz5) * relu'(f4): output tile T, 16 k-quads over the 128 features of f5
                const int T = s / 16, kq = s % 16;
                if (kq == 0) pnt_load_tile(mk, a.f4s, ic, 128, T, h);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int k = 4 * kq + jj;
                    acc = MFMA(aq[jj], dz5[k / 16][k % 16], k == 0 ? zero16 : acc);
                }
                if (kq == 15) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) dz4[T][r] = mk[r / 4][r % 4] > 0.f ? acc[r] : 0.f;
                    pnt_store_tile(dz4[T], a.dz4s, i, 128, T, h, inb);   // (a dead lane's dz5 is zero: so is its row)
                }
            } else {        // df2 (through the concat) = W3[:, 64:]^T dz4: output tile t, 16 k-quads over f4
                const int q = s - 64, t = q / 16, kq = q % 16;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int k = 4 * kq + jj;
                    acc = MFMA(aq[jj], dz4[k / 16][k % 16], k == 0 ? zero16 : acc);
                }
                if (kq == 15) pnt_store_tile(acc, a.df2s, i, 64, t, h, inb);
            }
            SCHED_FENCE();
        }
    }
}

// ---- backward, phase B --------------------------------------------------------------------------------
struct PnBwdBArgs {
    const float* stream;       // PNB_B_QUADS KiB: W2^T (8 quads), W1^T padded to 32 rows (4 quads), padding
    const int* voxs;
    const int* perm;
    const int* n_perm;
    const float* df2s;         // [n,64] gradient f2 receives through the concat
    const float* dp1;          // [V,64] dL/d pool1
    const int* arg1;           // [V,64]
    const float *f2s, *f1s;    // [n,64], [n,32]
    float *dz2s, *dz1s;        // [n,64], [n,32]
    float* d_inp;              // optional [n,6] in the ORIGINAL row order
    long long n;
};

__global__ void __launch_bounds__(256, 2) lidf_pnet_bwd_b_kernel(PnBwdBArgs a) {
    const long long AN = *a.n_perm;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, col = lane & 31;
    const __amdgpu_buffer_rsrc_t srs =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.stream, 0, PNB_B_QUADS * 1024, 0x00020000);
    const int vq = lane * 16;
    const long long ntile = (a.n + 127) / 128;   // (over the capacity: rows [AN, n) are written as zeros)
    const long long per = ntile / gridDim.x, rem = ntile % gridDim.x, bx = blockIdx.x;
    const long long tb = bx * per + (bx < rem ? bx : rem), te = tb + per + (bx < rem ? 1 : 0);
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (tb >= te) return;
    f32x4 ring[LIDF_RING];
#pragma unroll
    for (int i = 0; i < LIDF_RING; ++i) ring[i] = LDQ(srs, vq, i * 1024);
    for (long long tile = tb; tile < te; ++tile) {
        const long long i = tile * 128 + wave * 32 + col;
        const bool inb = i < a.n;
        if (tile * 128 + wave * 32 >= AN) {   // wave-uniform: zero rows only
            pnt_store_tile(zero16, a.dz2s, i, 64, 0, h, inb);
            pnt_store_tile(zero16, a.dz2s, i, 64, 1, h, inb);
            pnt_store_tile(zero16, a.dz1s, i, 32, 0, h, inb);
            continue;
        }
        const bool valid = i < AN;
        const long long ic = valid ? i : AN - 1;
        const int vox = a.voxs[ic];
        // dz2 = (df2 + the pooled gradient where this point is the arg row) * relu'(f2)
        f32x16 dz2[2];
        {
            const int* ap = a.arg1 + (size_t)vox * 64 + 4 * h;
            const float* dp = a.dp1 + (size_t)vox * 64 + 4 * h;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 dfv[4], fv[4];
                pnt_load_tile(dfv, a.df2s, ic, 64, t, h);
                pnt_load_tile(fv, a.f2s, ic, 64, t, h);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const i32x4 ar = *(const i32x4*)(ap + 32 * t + 8 * g);
                    const f32x4 d = *(const f32x4*)(dp + 32 * t + 8 * g);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float gsum = dfv[g][k] + ((ar[k] == (int)i) ? d[k] : 0.f);
                        dz2[t][4 * g + k] = (valid && fv[g][k] > 0.f) ? gsum : 0.f;
                    }
                }
                pnt_store_tile(dz2[t], a.dz2s, i, 64, t, h, inb);
            }
        }
        f32x4 m1[4];
        pnt_load_tile(m1, a.f1s, ic, 32, 0, h);
        const int prow = a.d_inp ? a.perm[ic] : 0;
        SCHED_FENCE();
        f32x16 dz1, acc;
#pragma unroll
        for (int s = 0; s < PNB_B_QUADS; ++s) {
            const f32x4 aq = ring[s % LIDF_RING];
            {
                const int nx = s + LIDF_RING;
                const int rel = nx < PNB_B_QUADS ? nx : nx - PNB_B_QUADS;
                ring[s % LIDF_RING] = LDQ(srs, vq + (rel & 3) * 1024, (rel >> 2) * 4096);
            }
            if (s < 8) {          // dz1 = (W2^T dz2) * relu'(f1): 8 k-quads over the 64 features of f2
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int k = 4 * s + jj;
                    acc = MFMA(aq[jj], dz2[k / 16][k % 16], k == 0 ? zero16 : acc);
                }
                if (s == 7) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) dz1[r] = m1[r / 4][r % 4] > 0.f ? acc[r] : 0.f;
                    pnt_store_tile(dz1, a.dz1s, i, 32, 0, h, inb);
                }
            } else if (s < 12) {  // d inp = W1^T dz1 (6 useful rows of a 32-row tile): 4 k-quads over f1
                if (a.d_inp) {    // (uniform)
                    const int kq = s - 8;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int k = 4 * kq + jj;
                        acc = MFMA(aq[jj], dz1[k], k == 0 ? zero16 : acc);
                    }
                    if (kq == 3 && valid) {   // rows 0..3 in the low half's registers 0..3, rows 4, 5 in the high half's 0, 1
                        float* o = a.d_inp + (size_t)prow * 6;
                        if (h == 0) {
                            *(f32x2*)o = f32x2{acc[0], acc[1]};
                            *(f32x2*)(o + 2) = f32x2{acc[2], acc[3]};
                        } else {
                            *(f32x2*)(o + 4) = f32x2{acc[0], acc[1]};
                        }
                    }
                }
            }
            SCHED_FENCE();
        }
    }
}

// ---- dW4 += dz5^T f4, db4 += column sums of dz5, from the at most V x 128 non-zero entries of dz5 ----------------
// dz5[arg2[v, f], f] = dp2[v, f]  =>  dW4[f, :] += sum_v dp2[v, f] f4[arg2[v, f], :]. One workgroup per output row
// f: four groups of 64 lanes take every fourth voxel (each lane two columns of the gathered row, four rows in
// flight), the groups' sums are added in a fixed order.
__global__ void __launch_bounds__(256) lidf_pnet_dw4_kernel(const float* __restrict__ dp2, const int* __restrict__ arg2,
                                                            const float* __restrict__ f4s, int V,
                                                            float* __restrict__ dW4, float* __restrict__ db4) {
    __shared__ f32x2 red[4][64];
    __shared__ float redb[4];
    const int f = blockIdx.x, grp = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x2 acc = {0.f, 0.f};
    float bs = 0.f;
    for (int v0 = grp; v0 < V; v0 += 16) {
        float d[4];
        int ar[4];
        f32x2 row[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int v = v0 + 4 * j;
            ar[j] = v < V ? arg2[(size_t)v * 128 + f] : -1;
            d[j] = v < V ? dp2[(size_t)v * 128 + f] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            row[j] = ar[j] >= 0 ? *(const f32x2*)(f4s + (size_t)ar[j] * 128 + 2 * lane) : f32x2{0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (ar[j] >= 0) {
                acc[0] = fmaf(d[j], row[j][0], acc[0]);
                acc[1] = fmaf(d[j], row[j][1], acc[1]);
                bs += d[j];
            }
        }
    }
    red[grp][lane] = acc;
    if (lane == 0) redb[grp] = bs;
    __syncthreads();
    if (grp == 0) {
        f32x2 s = red[0][lane];
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            s[0] += red[k][lane][0];
            s[1] += red[k][lane][1];
        }
        float* o = dW4 + (size_t)f * 128 + 2 * lane;
        o[0] += s[0];
        o[1] += s[1];
        if (lane == 0) db4[f] += redb[0] + redb[1] + redb[2] + redb[3];
    }
}

// ---- per-voxel sums of [n,128] rows stored in sorted order (a voxel = a contiguous run), in chunks of
// PNT_CH rows: partial[c, :] = sum of the chunk's rows; out[v, :] = the voxel's chunk sums added in order ---------
#define PNT_CH 256
__global__ void __launch_bounds__(256) lidf_pnet_chunk_sum_kernel(const float* __restrict__ rows,
                                                                  const int* __restrict__ vstart,
                                                                  const int* __restrict__ first, int V,
                                                                  float* __restrict__ partial) {
    __shared__ f32x4 red[8][32];
    const int c = blockIdx.x;
    if (c >= first[V]) return;
    int lo = 0, hi = V - 1;   // last voxel with first[v] <= c (voxels without rows share their successor's value)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (first[mid] <= c) lo = mid; else hi = mid - 1;
    }
    const int v = lo;
    const long long beg = (long long)vstart[v] + (long long)(c - first[v]) * PNT_CH;
    long long end = vstart[v + 1];
    if (end > beg + PNT_CH) end = beg + PNT_CH;
    const int rl = threadIdx.x >> 5, c4 = threadIdx.x & 31;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long long i = beg + rl; i < end; i += 32) {
        f32x4 x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long ii = i + 8 * j;
            x[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ii < end) x[j] = *(const f32x4*)(rows + (size_t)ii * 128 + 4 * c4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += x[j];
    }
    red[rl][c4] = acc;
    __syncthreads();
    if (rl == 0) {
        f32x4 s = red[0][c4];
#pragma unroll
        for (int k = 1; k < 8; ++k) s += red[k][c4];
        *(f32x4*)(partial + (size_t)c * 128 + 4 * c4) = s;
    }
}
__global__ void __launch_bounds__(64) lidf_pnet_chunk_final_kernel(const float* __restrict__ partial,
                                                                   const int* __restrict__ first, int V,
                                                                   float* __restrict__ out) {
    const int v = blockIdx.x * 2 + (threadIdx.x >> 5), c4 = threadIdx.x & 31;
    if (v >= V) return;
    const int c0 = first[v], c1 = first[v + 1];
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int c = c0; c < c1; ++c) s += *(const f32x4*)(partial + (size_t)c * 128 + 4 * c4);
    *(f32x4*)(out + (size_t)v * 128 + 4 * c4) = s;
}

// The same chunked sums for rows that are GATHERED through the sort: out[v, 0:256] = sum of S[perm[i] - row0, :] over
// the sorted rows i of voxel v with perm[i] >= row0 (stage 2's training step: S = dL/d(layer-1 pre-activation) of the
// decoder, one row per ray; the rays' predicted points are rows row0.. of the PointNet's input, so the sort of
// the PointNet's points already groups the rays by their end voxel — no second sort, fixed summation order).
__global__ void __launch_bounds__(256) lidf_pnet_gather_chunk_sum_kernel(const float* __restrict__ S,
                                                                         const int* __restrict__ perm, int row0,
                                                                         const int* __restrict__ vstart,
                                                                         const int* __restrict__ first, int V,
                                                                         float* __restrict__ partial) {
    __shared__ f32x4 red[4][64];
    const int c = blockIdx.x;
    if (c >= first[V]) return;
    int lo = 0, hi = V - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (first[mid] <= c) lo = mid; else hi = mid - 1;
    }
    const int v = lo;
    const long long beg = (long long)vstart[v] + (long long)(c - first[v]) * PNT_CH;
    long long end = vstart[v + 1];
    if (end > beg + PNT_CH) end = beg + PNT_CH;
    const int rl = threadIdx.x >> 6, c4 = threadIdx.x & 63;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long long i = beg + rl; i < end; i += 16) {
        int p[4];
        f32x4 x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long ii = i + 4 * j;
            p[j] = ii < end ? perm[ii] - row0 : -1;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            x[j] = p[j] >= 0 ? *(const f32x4*)(S + (size_t)p[j] * 256 + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
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
__global__ void __launch_bounds__(64) lidf_pnet_chunk_final256_kernel(const float* __restrict__ partial,
                                                                      const int* __restrict__ first,
                                                                      float* __restrict__ out) {
    const int v = blockIdx.x, c4 = threadIdx.x;
    const int c0 = first[v], c1 = first[v + 1];
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int c = c0; c < c1; ++c) s += *(const f32x4*)(partial + (size_t)c * 256 + 4 * c4);
    *(f32x4*)(out + (size_t)v * 256 + 4 * c4) = s;
}

// ---- launchers ------------------------------------------------------------------------------------------
static long long pnt_grid(long long n, int cus) {
    const long long ntile = (n + 127) / 128;
    const long long g = ntile < 2LL * cus ? ntile : 2LL * cus;
    return g < 1 ? 1 : g;
}

extern "C" hipError_t lidf_launch_pnet_train_fwd(int stage, const float* stream, const float* inp, const int* vox,
                                                 const int* perm, const int* n_perm, const float* gpart,
                                                 float* inps, int* voxs, float* f1s, float* f2s, float* f4s,
                                                 void* pool64, long long V, long long n, int cus, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    PnTrainFwdArgs a;
    a.stream = stream; a.inp = inp; a.vox = vox; a.perm = perm; a.n_perm = n_perm; a.gpart = gpart;
    a.inps = inps; a.voxs = voxs; a.f1s = f1s; a.f2s = f2s; a.f4s = f4s; a.pool64 = (u64*)pool64;
    a.V = (int)V; a.n = n;
    const long long g = pnt_grid(n, cus);
    if (stage == 1)
        hipLaunchKernelGGL(lidf_pnet_train_fwd_kernel<1>, dim3((unsigned)g), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL(lidf_pnet_train_fwd_kernel<2>, dim3((unsigned)g), dim3(256), 0, st, a);
    return hipGetLastError();
}

extern "C" hipError_t lidf_launch_pnet_unpack(const void* pool64, long long count, float* pool, int* arg,
                                              hipStream_t st) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_pnet_unpack_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st,
                       (const u64*)pool64, count, pool, arg);
    return hipGetLastError();
}

extern "C" int lidf_pnet_chunk_rows(void) { return PNT_CH; }
extern "C" hipError_t lidf_launch_pnet_tables(const int* scanned, int nblk, long long V, const int* n_perm,
                                              int* vstart, int* first, hipStream_t st) {
    if (V <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_pnet_tables_kernel, dim3(1), dim3(1024), 0, st, scanned, nblk, (int)V, n_perm, PNT_CH,
                       vstart, first);
    return hipGetLastError();
}

extern "C" hipError_t lidf_launch_pnet_bwd_a(const float* stream, const int* voxs, const int* n_perm,
                                             const float* dp2, const int* arg2, const float* f4s, float* dz4s,
                                             float* df2s, long long n, int cus, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    PnBwdAArgs a;
    a.stream = stream; a.voxs = voxs; a.n_perm = n_perm; a.dp2 = dp2; a.arg2 = arg2; a.f4s = f4s;
    a.dz4s = dz4s; a.df2s = df2s; a.n = n;
    hipLaunchKernelGGL(lidf_pnet_bwd_a_kernel, dim3((unsigned)pnt_grid(n, cus)), dim3(256), 0, st, a);
    return hipGetLastError();
}

extern "C" hipError_t lidf_launch_pnet_bwd_b(const float* stream, const int* voxs, const int* perm,
                                             const int* n_perm, const float* df2s, const float* dp1,
                                             const int* arg1, const float* f2s, const float* f1s, float* dz2s,
                                             float* dz1s, float* d_inp, long long n, int cus, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    PnBwdBArgs a;
    a.stream = stream; a.voxs = voxs; a.perm = perm; a.n_perm = n_perm; a.df2s = df2s; a.dp1 = dp1; a.arg1 = arg1;
    a.f2s = f2s; a.f1s = f1s; a.dz2s = dz2s; a.dz1s = dz1s; a.d_inp = d_inp; a.n = n;
    hipLaunchKernelGGL(lidf_pnet_bwd_b_kernel, dim3((unsigned)pnt_grid(n, cus)), dim3(256), 0, st, a);
    return hipGetLastError();
}

extern "C" hipError_t lidf_launch_pnet_dw4(const float* dp2, const int* arg2, const float* f4s, long long V,
                                           float* dW4, float* db4, hipStream_t st) {
    if (V <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_pnet_dw4_kernel, dim3(128), dim3(256), 0, st, dp2, arg2, f4s, (int)V, dW4, db4);
    return hipGetLastError();
}

// s[v, :] = sum of rows[vstart[v] .. vstart[v + 1]) ([n,128], sorted order). partial: (n / PNT_CH + V + 1) x 128
// floats.
extern "C" hipError_t lidf_launch_pnet_segsum(const float* rows, const int* vstart, const int* first, long long V,
                                              long long n, float* partial, float* out, hipStream_t st) {
    if (V <= 0) return hipSuccess;
    const long long maxc = n / PNT_CH + V + 1;
    hipLaunchKernelGGL(lidf_pnet_chunk_sum_kernel, dim3((unsigned)maxc), dim3(256), 0, st, rows, vstart, first, (int)V,
                       partial);
    hipLaunchKernelGGL(lidf_pnet_chunk_final_kernel, dim3((unsigned)((V + 1) / 2)), dim3(64), 0, st, partial, first,
                       (int)V, out);
    return hipGetLastError();
}

// out[v, 0:256] = sum over the sorted rows i of voxel v with perm[i] >= row0 of S[perm[i] - row0, 0:256];
// partial: (n / PNT_CH + V + 1) x 256 floats
extern "C" hipError_t lidf_launch_pnet_gather_segsum(const float* S, const int* perm, long long row0,
                                                     const int* vstart, const int* first, long long V, long long n,
                                                     float* partial, float* out, hipStream_t st) {
    if (V <= 0) return hipSuccess;
    const long long maxc = n / PNT_CH + V + 1;
    hipLaunchKernelGGL(lidf_pnet_gather_chunk_sum_kernel, dim3((unsigned)maxc), dim3(256), 0, st, S, perm, (int)row0,
                       vstart, first, (int)V, partial);
    hipLaunchKernelGGL(lidf_pnet_chunk_final256_kernel, dim3((unsigned)V), dim3(64), 0, st, partial, first, out);
    return hipGetLastError();
}
