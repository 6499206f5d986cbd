// lidf_pointnet.hip — the per-point chains of PointNet2Stage (gfx950 / CDNA4), inference.
//
// models/pointnet.py:22-38 (called at models/pipeline.py:149-160 and, twice per frame, :1009-1016):
//   f1 = relu(point_lin1(inp))            6 -> 32
//   f2 = relu(point_lin2(f1))            32 -> 64      pool1[v] = max over the voxel's points of f2
//   g1 = relu(vox_lin1(pool1))           per voxel
//   f4 = relu(point_lin3(cat(g1[vox], f2)))   128 -> 128
//   f5 = relu(point_lin4(f4))           128 -> 128     pool2[v] = max over the voxel's points of f5
//   out = relu(vox_lin2(pool2))          per voxel
// The per-voxel layers are launches over V rows (lidf_linear_kernel). The per-point layers were one
// launch per layer with every intermediate ([n,32], [n,64], [n,128]) written and read back; here
// they are two register chains in the accumulator layout of the decoder kernel (lidf_points.hip:
// a layer's 32 x 32 output tile is the next layer's B operand):
//   stage 1:  inp -> f1 -> f2 -> pool1
//   stage 2:  inp -> f1 -> f2 (recomputed: 38 matrix instructions) -> f4 -> f5 -> pool2
// with W3[:, :64] g1[v] + b3 as a gathered per-voxel row (gpart). No per-point intermediate touches
// memory; per point 24 B in (+ the voxel index), nothing out but the pooled maxima.
//
// Stream (lidf_pack_pointnet_kernel), 1 KiB quads consumed in order through an 8-deep ring:
//   P1  1 quad            K = 6 inputs + bias (operand columns 4h + {0..3}: 6 = 1.0, 7 = 0)
//   P2  2 x 5 quads       K = 32 + bias, quad = 2 kq + t
//   P3  4 x 8 quads       K = 64 (second half of W3's columns), quad = 8 T + kq   [stage 2]
//   P4  4 x 17 quads      K = 128 + bias, quad = 17 T + kq                         [stage 2]
//   (+ 1 padding quad)
// Max-pool (values are post-ReLU, >= 0: their bit patterns order like integers, 0 is the identity):
//  * voxel table small enough for LDS (V x F floats <= 144 KiB — a frame has 50-150 occupied
//    voxels): every lane takes the integer maximum of its values into the workgroup's LDS table,
//    the workgroup writes the table to its slab of a scratch array, lidf_pointnet_poolmax_kernel
//    takes the maximum over the slabs. No global atomics: with ~2,000 points per voxel they all
//    land on the same few hundred addresses and were 80 % of the PointNet's time.
//  * otherwise as in lidf_linear.hip: the wave reduces per distinct voxel of its 32 rows, one lane
//    issues global integer atomic maxima where a plain read does not already prove them unnecessary.
#include "lidf_device.h"

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define LDQ(rs, voff, soff) \
    __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128((rs), (voff), (soff), 0))

static_assert(PN_P1 + PN_P2 <= PN_S1_QUADS && PN_P1 + PN_P2 + PN_P3 + PN_P4 <= PN_S2_QUADS, "stream");
static_assert(PN_S1_QUADS % LIDF_RING == 0 && PN_S2_QUADS % LIDF_RING == 0, "ring phase");

struct PnetChainArgs {
    const float* stream;   // PN_S2_QUADS KiB
    const float* inp;      // [n,6]
    const int* vox;        // [n] (negative: the row is left out)
    const float* gpart;    // [V,128] = W3[:, :64] g1 + b3 (stage 2)
    float* pool;           // stage 1: [V,64], stage 2: [V,128]  (zeroed by the caller; takes integer maxima)
    float* part;           // global-atomic path with copies: [copies, V * F] (zeroed)
    int V, copies;         // V: rows of `pool`
    int v_tab;             // LDS instantiation: rows of the workgroup's full table (rows [0, v_tab) of the voxels)
    long long n;
    // sync-free frame path: device-side counts. n_dev overrides n (n = capacity of the launch).
    // V_dev selects the walk on the device inside ONE launch of the LDS instantiation: *V_dev <= v_tab
    // -> every voxel has a row of the workgroup's table; else, with perm, the voxel-sorted walk over a
    // windowed table; else the full table for the first v_tab voxels and lane-level global maxima for
    // the rest (correct, slow: a single frame with more occupied voxels than the table holds).
    const int* n_dev;
    const int* V_dev;
    // voxel-sorted walk (large tables): point p of the launch is perm[p], the points are grouped by
    // voxel (lidf_launch_pointnet_sort), points left out of the pooling are not in perm; the number of
    // points is *n_perm. The global-atomic pooling then meets one or two voxels per wavefront.
    const int* perm;
    const int* n_perm;
    // window > 0 (with perm; the LDS instantiation): the workgroup's table holds `window` rows starting
    // at the voxel of its first point — its contiguous run of sorted points spans a handful of voxels;
    // a point beyond the table raises `pool` directly. Either table is flushed into `pool` with one
    // atomic maximum per touched entry at the end of the workgroup (round 4; rounds 2-3 wrote the full
    // table to a slab per workgroup and reduced the slabs in a second launch).
    int window;
};

__global__ void lidf_pack_pointnet_kernel(PnetW w, float* __restrict__ stream,
                                          const LidfPackGuardState* guard) {
    if (guard && guard->dirty == 0) return;   // guarded packing: fingerprint unchanged
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= PN_S2_QUADS * 256) return;
    stream[e] = pn_stream_value(w, e);
}

__device__ __forceinline__ void pn_relu(f32x16& v) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = fmaxf(v[i], 0.f);
}

// max-pool the NT tiles of `acc` (row = lane & 31 of this wavefront) into pool[vox, 32 T0 ...]
template <int NT>
__device__ __forceinline__ void pn_pool(const f32x16 (&acc)[NT], const int T0, const int vox,
                                        const bool valid, float* pool, const int ld, const int h,
                                        const int col) {
    unsigned todo = (unsigned)__ballot(valid && h == 0 && vox >= 0);
    int rounds = 0;
    while (todo && rounds < 4) {
        const int lead = __builtin_ctz(todo);
        const int vv = __builtin_amdgcn_readlane(vox, lead);
        const unsigned mem = (unsigned)__ballot(vox == vv) & todo;
        todo &= ~mem;
        ++rounds;
        const bool mine = vox == vv;
        int* pp = (int*)pool + (size_t)vv * ld + 32 * T0 + 4 * h;
        // what the table holds now (plain, possibly stale reads — entries only grow), requested in
        // one batch before the shuffles so that the round waits for memory once
        f32x4 seen[NT][4];
        if (col == lead) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int g = 0; g < 4; ++g) seen[t][g] = *(const f32x4*)(pp + t * 32 + 8 * g);
            }
        }
        f32x4 m[NT][4];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float x = mine ? acc[t][4 * g + i] : 0.f;
#pragma unroll
                    for (int sft = 16; sft >= 1; sft >>= 1) x = fmaxf(x, __shfl_xor(x, sft));
                    m[t][g][i] = x;
                }
            }
        }
        if (col == lead) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (m[t][g][i] > seen[t][g][i])
                            atomicMax(pp + t * 32 + 8 * g + i, __float_as_int(m[t][g][i]));
                }
            }
        }
    }
    if (todo && valid && ((todo >> col) & 1u)) {   // more than four distinct voxels: per lane
        int* pp = (int*)pool + (size_t)vox * ld + 32 * T0 + 4 * h;
        f32x4 seen[NT][4];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int g = 0; g < 4; ++g) seen[t][g] = *(const f32x4*)(pp + t * 32 + 8 * g);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (acc[t][4 * g + i] > seen[t][g][i])
                        atomicMax(pp + t * 32 + 8 * g + i, __float_as_int(acc[t][4 * g + i]));
            }
        }
    }
}

// lane-level atomic maxima of one accumulator tile straight into the global table (rare path)
__device__ __forceinline__ void pn_pool_lane(const f32x16& acc, const int T, const int vox, float* pool,
                                             const int F, const int h) {
    int* row = (int*)pool + (size_t)vox * F + 32 * T + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = acc[4 * g + i];
            if (v > 0.f) atomicMax(row + 8 * g + i, __float_as_int(v));
        }
    }
}

// lane-level integer maxima of one accumulator tile into the LDS table row of the lane's voxel
__device__ __forceinline__ void pn_pool_lds(const f32x16& acc, const int T, const int vox,
                                            const bool valid, int* tab, const int F, const int h) {
    if (!valid || vox < 0) return;
    int* row = tab + vox * F + 32 * T + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = acc[4 * g + i];
            if (v > 0.f) atomicMax(row + 8 * g + i, __float_as_int(v));
        }
    }
}

template <int STAGE, bool LDSPOOL>
__global__ void __launch_bounds__(256, 2) lidf_pointnet_chain_kernel(PnetChainArgs a) {
    extern __shared__ int pn_tab[];
    constexpr int F = STAGE == 1 ? 64 : 128;
    // rows of the LDS table are F + 1 words apart: with a stride of 64 / 128 words every row starts in the same
    // bank, and the 32 points of a wavefront — mostly different voxels, the same feature index — hit ONE
    // bank with every atomic (SQ_LDS_BANK_CONFLICT 92 % of the LDS cycles of the stage-2 chain)
    constexpr int FP = F + 1;
    bool windowed = LDSPOOL && a.window > 0 && a.perm;
    if (LDSPOOL && a.V_dev) windowed = windowed && *a.V_dev > a.v_tab;   // device-side choice of the walk
    const bool use_perm = a.perm && (!LDSPOOL || windowed);
    const long long AN = use_perm ? (long long)*a.n_perm : (a.n_dev ? (long long)*a.n_dev : a.n);
    // rows of the workgroup's table in play: the window, or one row per voxel up to the table's capacity
    // (a frame has 50-150 occupied voxels: zeroing and flushing all 288 rows the launch reserves LDS for
    // was a third of the stage-1 chain's time)
    const int v_now = a.V_dev ? *a.V_dev : a.V;
    const int tab_rows = windowed ? a.window : (v_now < a.v_tab ? v_now : a.v_tab);
    constexpr int NQ = STAGE == 1 ? PN_S1_QUADS : PN_S2_QUADS;
    // global-atomic path: the workgroup's copy of the table (thousands of wavefronts raising the
    // same few rows serialise on their addresses; `copies` tables divide that, a reduce follows)
    float* const gpool = (!LDSPOOL && a.copies > 0)
                             ? a.part + (size_t)(blockIdx.x % a.copies) * a.V * F : a.pool;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int h = lane >> 5;
    const int col = lane & 31;
    const float one_b = h ? 0.f : 1.f;
    const __amdgpu_buffer_rsrc_t srs =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.stream, 0, PN_S2_QUADS * 1024, 0x00020000);
    const int vq = lane * 16;
    const long long ntile = (AN + 127) / 128;
    const long long per = ntile / gridDim.x, rem = ntile % gridDim.x;
    const long long bx = blockIdx.x;
    const long long tb = bx * per + (bx < rem ? bx : rem);
    const long long te = tb + per + (bx < rem ? 1 : 0);
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    if (tb >= te || tb * 128 >= AN) return;   // (workgroup-uniform: nothing to pool, nothing to flush)
    if (LDSPOOL) {
        for (int i = threadIdx.x; i < tab_rows * FP; i += 256) pn_tab[i] = 0;
        __syncthreads();
    }
    f32x4 ring[LIDF_RING];
#pragma unroll
    for (int i = 0; i < LIDF_RING; ++i) ring[i] = LDQ(srs, vq, i * 1024);
    // windowed table: rows [vbase, vbase + window) of the voxel table (sorted points: ascending voxels)
    int vbase = 0;
    if (windowed) vbase = a.vox[a.perm[tb * 128]];

    // a tile's point index, voxel and operand columns (4h + {0..3} of [x0..x5, 1, 0]) are requested while
    // the tile before it runs: the chain of a tile is ~1 us of matrix instructions in stage 1, and the
    // dependent index -> voxel / input round trips at its head were most of a tile's time
    auto fetch = [&](long long tile, int& vox_o, float (&b)[4]) {
        const long long p = tile * 128 + wave * 32 + col;
        long long pc = p < AN ? p : AN - 1;
        if (use_perm) pc = a.perm[pc];   // voxel-sorted walk
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
        if (tile * 128 + wave * 32 >= AN) break;   // wave-uniform
        const long long p = tile * 128 + wave * 32 + col;
        const bool valid = p < AN;
        const int vox = vox_n;
        const float b1[4] = {b1n[0], b1n[1], b1n[2], b1n[3]};
        // row of the LDS table; a point beyond the table goes to the global table (vrow = -1)
        const int vrow = (vox >= vbase && vox - vbase < tab_rows) ? vox - vbase : -1;
        const bool spill = LDSPOOL && valid && vox >= 0 && vrow < 0;
        // stage 2: the gathered per-voxel row the layer-3 accumulators start from
        f32x16 F4[STAGE == 2 ? 4 : 1];
        if (STAGE == 2) {
            const float* gp = a.gpart + (size_t)(vox >= 0 ? vox : 0) * 128 + 4 * h;
#pragma unroll
            for (int T = 0; T < 4; ++T) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = *(const f32x4*)(gp + 32 * T + 8 * g);
#pragma unroll
                    for (int i = 0; i < 4; ++i) F4[STAGE == 2 ? T : 0][4 * g + i] = v[i];
                }
            }
        }
        if (tile + 1 < te && (tile + 1) * 128 + wave * 32 < AN) fetch(tile + 1, vox_n, b1n);
        SCHED_FENCE();

        f32x16 F1, F2[2], acc;
        f32x16 F5[(STAGE == 2 && !LDSPOOL) ? 4 : 1];   // global-atomic path: pooled once per tile row
#pragma unroll
        for (int s = 0; s < NQ; ++s) {
            const f32x4 aq = ring[s % LIDF_RING];
            {
                const int nx = s + LIDF_RING;
                const int rel = nx < NQ ? nx : nx - NQ;   // wraps: the same stream for the next tile
                ring[s % LIDF_RING] = LDQ(srs, vq + (rel & 3) * 1024, (rel >> 2) * 4096);
            }
            if (s < PN_P1) {
                F1 = MFMA(aq[0], b1[0], zero16);
                F1 = MFMA(aq[1], b1[1], F1);
                F1 = MFMA(aq[2], b1[2], F1);
                F1 = MFMA(aq[3], b1[3], F1);
                pn_relu(F1);
            } else if (s < PN_P1 + PN_P2) {
                const int q = s - PN_P1, kq = q / 2, t = q % 2;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int k = 4 * kq + jj;
                    if (k < 16) F2[t] = MFMA(aq[jj], F1[k], k == 0 ? zero16 : F2[t]);
                    else if (k == 16) F2[t] = MFMA(aq[jj], one_b, F2[t]);
                }
                if (kq == 4) pn_relu(F2[t]);
            } else if (STAGE == 2 && s < PN_P1 + PN_P2 + PN_P3) {
                const int q = s - PN_P1 - PN_P2, T = q / 8, kq = q % 8;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int k = 4 * kq + jj;
                    F4[STAGE == 2 ? T : 0] = MFMA(aq[jj], F2[k / 16][k % 16], F4[STAGE == 2 ? T : 0]);
                }
                if (kq == 7) pn_relu(F4[STAGE == 2 ? T : 0]);
            } else if (STAGE == 2 && s < PN_P1 + PN_P2 + PN_P3 + PN_P4) {
                const int q = s - PN_P1 - PN_P2 - PN_P3, T = q / 17, kq = q % 17;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int k = 4 * kq + jj;
                    if (k < 64) acc = MFMA(aq[jj], F4[STAGE == 2 ? k / 16 : 0][k % 16], k == 0 ? zero16 : acc);
                    else if (k == 64) acc = MFMA(aq[jj], one_b, acc);
                }
                if (kq == 16) {
                    pn_relu(acc);
                    if (LDSPOOL) {
                        pn_pool_lds(acc, T, vrow, valid, pn_tab, 129, h);
                        if (spill) pn_pool_lane(acc, T, vox, a.pool, 128, h);
                    } else {
                        F5[(STAGE == 2 && !LDSPOOL) ? T : 0] = acc;
                    }
                }
            }
            SCHED_FENCE();
        }
        if (STAGE == 2 && !LDSPOOL) {
            // (a zero-length array type is avoided: the template instantiates F5[1] elsewhere)
            pn_pool<(STAGE == 2 && !LDSPOOL) ? 4 : 1>(F5, 0, vox, valid, gpool, 128, h, col);
        }
        if (STAGE == 1) {
            if (LDSPOOL) {
                pn_pool_lds(F2[0], 0, vrow, valid, pn_tab, 65, h);
                pn_pool_lds(F2[1], 1, vrow, valid, pn_tab, 65, h);
                if (spill) {
                    pn_pool_lane(F2[0], 0, vox, a.pool, 64, h);
                    pn_pool_lane(F2[1], 1, vox, a.pool, 64, h);
                }
            } else {
                pn_pool<2>(F2, 0, vox, valid, gpool, 64, h, col);
            }
        }
    }
    if (LDSPOOL) {
        // the table's rows into the global table: one atomic maximum per touched entry and workgroup
        __syncthreads();
        for (int i = threadIdx.x; i < tab_rows * F; i += 256) {
            const int v = pn_tab[(i / F) * FP + i % F];
            const int row = vbase + i / F;
            if (v > 0 && row < a.V) atomicMax((int*)a.pool + (size_t)row * F + i % F, v);
        }
    }
}

// pool[e] = max over the G slabs (entries are >= 0): a workgroup owns 64 entries (16 float4), thread
// (k, c) takes every 16th slab of its float4 with eight loads in flight, the sixteen partial maxima
// are combined through LDS.
__global__ void __launch_bounds__(256) lidf_pointnet_poolmax_kernel(const float* __restrict__ part, int G,
                                                                    long long count,
                                                                    float* __restrict__ pool) {
    __shared__ f32x4 red[16][16];
    const int k = threadIdx.x >> 4, c = threadIdx.x & 15;
    const long long e = ((long long)blockIdx.x * 16 + c) * 4;
    f32x4 m = {0.f, 0.f, 0.f, 0.f};
    if (e < count) {
        int g = k;
        for (; g + 112 < G; g += 128) {
            f32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *(const f32x4*)(part + (size_t)(g + 16 * j) * count + e);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int i = 0; i < 4; ++i) m[i] = fmaxf(m[i], v[j][i]);
            }
        }
        for (; g < G; g += 16) {
            const f32x4 v = *(const f32x4*)(part + (size_t)g * count + e);
#pragma unroll
            for (int i = 0; i < 4; ++i) m[i] = fmaxf(m[i], v[i]);
        }
    }
    red[k][c] = m;
    __syncthreads();
    if (k == 0 && e < count) {
#pragma unroll
        for (int j = 1; j < 16; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) m[i] = fmaxf(m[i], red[j][c][i]);
        }
        *(f32x4*)(pool + e) = m;
    }
}

extern "C" hipError_t lidf_launch_pack_pointnet(const float* w_p1, const float* b_p1,
                                                const float* w_p2, const float* b_p2,
                                                const float* w_p3, const float* w_p4,
                                                const float* b_p4, float* stream,
                                                const LidfPackGuardState* guard, hipStream_t st) {
    PnetW w = {w_p1, b_p1, w_p2, b_p2, w_p3, w_p4, b_p4};
    hipLaunchKernelGGL(lidf_pack_pointnet_kernel, dim3(PN_S2_QUADS), dim3(256), 0, st, w, stream,
                       guard);
    return hipGetLastError();
}

extern "C" size_t lidf_pointnet_chain_stream_bytes(void) { return (size_t)PN_S2_QUADS * 1024; }

#define PN_WINDOW 32     // rows of the windowed table of the voxel-sorted walk (16 KiB at 128 features)
#define PN_MAX_WGS 512
#define PN_LDS_LIMIT (288 * 129 * 4)   // 288 rows of 128 + 1 words
#define PN_COPIES 16
static int pn_copies(long long V) {
    long long c = (32LL << 20) / (V * 512);
    return (int)(c > PN_COPIES ? PN_COPIES : (c < 1 ? 1 : c));
}
// Scratch of the pooling: none when the table fits LDS (the workgroups flush their tables into `pool`),
// else up to PN_COPIES copies of the table for the unsorted global-atomic path (at most 32 MiB).
extern "C" size_t lidf_pointnet_pool_scratch_bytes(long long V) {
    if (V <= 0) return 0;
    if ((size_t)V * 129 * 4 > PN_LDS_LIMIT) return (size_t)pn_copies(V) * V * 128 * 4;
    return 0;
}

template <int STAGE>
static hipError_t launch_chain_lds(const PnetChainArgs& a, long long g, size_t lds, hipStream_t st) {
    static bool configured[64];
    hipError_t e = lidf_max_lds_once(configured, (const void*)lidf_pointnet_chain_kernel<STAGE, true>, PN_LDS_LIMIT);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((lidf_pointnet_chain_kernel<STAGE, true>), dim3((unsigned)g), dim3(256), lds, st, a);
    return hipGetLastError();
}

// stage 1: pool = pool1 [V,64]; stage 2: gpart [V,128], pool = pool2 [V,128]. `pool` must be zeroed by the
// caller and takes integer maxima: through per-workgroup LDS tables when V x 128 floats fit, else (with
// `part`, lidf_pointnet_pool_scratch_bytes(V) bytes) through a few zeroed copies of the table that take
// global atomic maxima and a reduce, else straight into `pool`.
extern "C" hipError_t lidf_launch_pointnet_chain(int stage, const float* stream, const float* inp,
                                                 const int* vox, const float* gpart, float* pool,
                                                 float* part, long long V, long long n, int cus,
                                                 hipStream_t st) {
    if (n <= 0) return hipSuccess;
    PnetChainArgs a;
    a.stream = stream; a.inp = inp; a.vox = vox; a.gpart = gpart; a.pool = pool; a.n = n;
    a.part = part; a.V = (int)V; a.copies = 0; a.v_tab = (int)V;
    a.n_dev = nullptr; a.V_dev = nullptr; a.perm = nullptr; a.n_perm = nullptr; a.window = 0;
    const int F = stage == 1 ? 64 : 128;
    const long long count = V * F;
    const long long ntile = (n + 127) / 128;
    const size_t lds = (size_t)V * (F + 1) * 4;
    if ((size_t)V * 129 * 4 <= PN_LDS_LIMIT) {
        // two workgroups per CU while two tables fit, else one
        long long g = lds <= 80 * 1024 ? 2LL * cus : cus;
        if (g > PN_MAX_WGS) g = PN_MAX_WGS;
        if (g > ntile) g = ntile;
        a.part = nullptr;
        return stage == 1 ? launch_chain_lds<1>(a, g, lds, st) : launch_chain_lds<2>(a, g, lds, st);
    }
    if (part) {
        a.copies = pn_copies(V);
        hipError_t e = hipMemsetAsync(part, 0, (size_t)a.copies * count * 4, st);
        if (e != hipSuccess) return e;
    }
    const long long g = ntile < 2LL * cus ? ntile : 2LL * cus;
    if (stage == 1)
        hipLaunchKernelGGL((lidf_pointnet_chain_kernel<1, false>), dim3((unsigned)g), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((lidf_pointnet_chain_kernel<2, false>), dim3((unsigned)g), dim3(256), 0, st, a);
    if (part)
        hipLaunchKernelGGL(lidf_pointnet_poolmax_kernel, dim3((unsigned)((count + 63) / 64)), dim3(256), 0,
                           st, part, a.copies, count, pool);
    return hipGetLastError();
}

// The same stage with device-side counts (the sync-free frame path) in ONE launch: n_cap / V_cap bound
// it, *n_dev / *V_dev are the sizes. Per-workgroup LDS tables of v_lds rows (the caller's bound for
// typical frames, <= 288) serve *V_dev <= v_lds; beyond it the voxel-sorted walk over a windowed table
// when `perm` is given (several frames per batch), else the first v_lds voxels through the table and the
// others through lane-level global maxima. `pool` ([V_cap, F]) is zeroed by the caller.
extern "C" hipError_t lidf_launch_pointnet_chain_dev(int stage, const float* stream, const float* inp,
                                                     const int* vox, const float* gpart, float* pool,
                                                     long long V_cap, int v_lds,
                                                     long long n_cap, const int* n_dev,
                                                     const int* V_dev, const int* perm,
                                                     const int* n_perm, int cus, hipStream_t st) {
    if (n_cap <= 0) return hipSuccess;
    if (v_lds <= 0 || (size_t)v_lds * 129 * 4 > PN_LDS_LIMIT || !V_dev) return hipErrorInvalidValue;
    PnetChainArgs a;
    a.stream = stream; a.inp = inp; a.vox = vox; a.gpart = gpart; a.pool = pool; a.n = n_cap;
    a.part = nullptr; a.V = (int)V_cap; a.copies = 0; a.n_dev = n_dev; a.V_dev = V_dev; a.v_tab = v_lds;
    a.perm = perm; a.n_perm = n_perm; a.window = perm ? PN_WINDOW : 0;
    const int F = stage == 1 ? 64 : 128;
    const long long ntile = (n_cap + 127) / 128;
    size_t lds = (size_t)v_lds * (F + 1) * 4;
    if (perm && lds < (size_t)PN_WINDOW * (F + 1) * 4) lds = (size_t)PN_WINDOW * (F + 1) * 4;
    long long g = lds <= 80 * 1024 ? 2LL * cus : cus;   // two tables per CU while they fit
    if (g > PN_MAX_WGS) g = PN_MAX_WGS;
    if (g > ntile) g = ntile;
    return stage == 1 ? launch_chain_lds<1>(a, g, lds, st) : launch_chain_lds<2>(a, g, lds, st);
}


// ------------------------------------------------------------------------------------------------
// Large voxel tables (more rows than the LDS pooling holds: several frames per batch, or a densely
// occupied grid): the points are grouped by voxel first, so that a wavefront of the chains meets one or
// two voxels instead of ~28 and its pre-reduced maxima cost a few global atomics per tile — with the
// points in input order every point raised its own 64 / 128 table entries (86,800 points on 729 voxels:
// 0.24 ms per PointNet pass, all of it in the pooling). Counting sort, order inside a voxel irrelevant
// (the pooling is a maximum): count per voxel (global atomics on a [V] table), one-workgroup exclusive
// scan, placement through per-voxel cursors. Points with a negative voxel are left out.
// scratch: lidf_pointnet_sort_bytes(n, V) bytes.
// ------------------------------------------------------------------------------------------------
extern "C" size_t lidf_sort_idx_ws_bytes(long long P, long long V);
extern "C" hipError_t lidf_launch_sort_idx(const int* idx, long long P, const int* n_dev, long long V,
                                           void* ws, const int** perm_out, const int** n_perm_out,
                                           hipStream_t st);
extern "C" int lidf_pointnet_lds_max_voxels(void) { return PN_LDS_LIMIT / 516; }
// scratch of the sort (0: the table is too large for it — the unsorted global-atomic path is taken)
extern "C" size_t lidf_pointnet_sort_bytes(long long n, long long V) { return lidf_sort_idx_ws_bytes(n, V); }

// ------------------------------------------------------------------------------------------------
// The same grouping without an order inside a voxel (the pooling is a maximum: the pooled table does not
// depend on it) — the frame path's batches: per-voxel counts by wave-aggregated atomics (neighbouring
// pixels share a voxel: one atomic per distinct voxel of a wavefront), a one-workgroup exclusive scan that
// leaves the table of counts zeroed for the next call, placement through per-voxel cursors (aggregated the
// same way). Three light launches (the stable sort above: per-block histograms, a three-launch scan over
// blocks x voxels and a single-wavefront placement per block: 67 us per pass at 4 frames).
// ws: [V] counts (ZERO on first use; left zero) | [V] cursors | [1] total | [n_cap] perm.
// ------------------------------------------------------------------------------------------------
extern "C" size_t lidf_group_idx_bytes(long long n_cap, long long V) {
    return (size_t)(2 * V + 64 + n_cap) * 4;
}
__device__ __forceinline__ int pn_group_claim(int* table, const int v, const bool live) {
    // wave-aggregated claim of one slot per live lane in table[v]: the lanes of a voxel are ranked first (no
    // memory operation inside the loop over the wavefront's distinct voxels), then the leaders of all groups
    // claim their groups' slots with ONE atomic instruction and hand the base to their lanes
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(live);
    int leader = lane, rank = 0, cnt = 0;
    while (todo) {
        const int lead = __builtin_ctzll(todo);
        const int vv = __builtin_amdgcn_readlane(v, lead);
        const unsigned long long m = __ballot(live && v == vv);
        if (live && v == vv) {
            leader = lead;
            rank = (int)__popcll(m & ((1ull << lane) - 1ull));
            cnt = (int)__popcll(m);
        }
        todo &= ~m;
    }
    int base = 0;
    if (live && lane == leader) base = atomicAdd(table + v, cnt);
    base = __shfl(base, leader);
    return base + rank;
}
__global__ void __launch_bounds__(256) lidf_group_count_kernel(const int* __restrict__ vox, long long n,
                                                               const int* __restrict__ n_dev,
                                                               int* __restrict__ counts) {
    if (n_dev) n = *n_dev;
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if ((long long)blockIdx.x * 256 >= n) return;
    const int v = p < n ? vox[p] : -1;
    (void)pn_group_claim(counts, v, v >= 0);
}
__global__ void __launch_bounds__(1024) lidf_group_scan_kernel(int* __restrict__ counts, int V,
                                                               int* __restrict__ cursor) {
    __shared__ int s_w[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int carry = 0;
    for (int b = 0; b < V; b += 1024) {
        const int k = b + threadIdx.x;
        const int c = k < V ? counts[k] : 0;
        int inc = c;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const int o = __shfl_up(inc, sft);
            if (lane >= sft) inc += o;
        }
        __syncthreads();
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        int wpre = 0, total = 0;
        for (int w = 0; w < 16; ++w) {
            const int t = s_w[w];
            wpre += w < wave ? t : 0;
            total += t;
        }
        if (k < V) {
            cursor[k] = carry + wpre + inc - c;
            counts[k] = 0;   // (ready for the next call)
        }
        carry += total;
    }
    if (threadIdx.x == 0) cursor[V] = carry;   // number of placed points
}
__global__ void __launch_bounds__(256) lidf_group_place_kernel(const int* __restrict__ vox, long long n,
                                                               const int* __restrict__ n_dev,
                                                               int* __restrict__ cursor, int* __restrict__ perm) {
    if (n_dev) n = *n_dev;
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if ((long long)blockIdx.x * 256 >= n) return;
    const int v = p < n ? vox[p] : -1;
    const int slot = pn_group_claim(cursor, v, v >= 0);
    if (v >= 0) perm[slot] = (int)p;
}
extern "C" hipError_t lidf_launch_group_idx(const int* vox, long long n_cap, const int* n_dev, long long V,
                                            void* ws, const int** perm_out, const int** n_perm_out,
                                            hipStream_t st) {
    if (V <= 0 || n_cap <= 0 || !ws) return hipErrorInvalidValue;
    int* counts = (int*)ws;
    int* cursor = counts + V;          // [V] + the total at cursor[V] (= start of the 64-word pad)
    int* perm = counts + 2 * V + 64;
    const unsigned g = (unsigned)((n_cap + 255) / 256);
    hipLaunchKernelGGL(lidf_group_count_kernel, dim3(g), dim3(256), 0, st, vox, n_cap, n_dev, counts);
    hipLaunchKernelGGL(lidf_group_scan_kernel, dim3(1), dim3(1024), 0, st, counts, (int)V, cursor);
    // (the total is read by the chains — n_perm — AFTER the placement advanced the cursors of the voxels, not
    // cursor[V]: nothing claims slots of index V)
    hipLaunchKernelGGL(lidf_group_place_kernel, dim3(g), dim3(256), 0, st, vox, n_cap, n_dev, cursor, perm);
    *perm_out = perm;
    *n_perm_out = cursor + V;
    return hipGetLastError();
}

// One chain stage over the voxel-sorted points: global atomic maxima straight into `pool` ([V, F],
// zeroed by the caller).
extern "C" hipError_t lidf_launch_pointnet_chain_sorted(int stage, const float* stream, const float* inp,
                                                        const int* vox, const float* gpart, float* pool,
                                                        long long V, long long n_cap, const int* perm,
                                                        const int* n_perm, int cus, hipStream_t st) {
    if (n_cap <= 0) return hipSuccess;
    PnetChainArgs a;
    a.stream = stream; a.inp = inp; a.vox = vox; a.gpart = gpart; a.pool = pool; a.n = n_cap;
    a.part = nullptr; a.V = (int)V; a.copies = 0; a.v_tab = PN_WINDOW;
    a.n_dev = nullptr; a.V_dev = nullptr;
    a.n_perm = n_perm; a.perm = perm; a.window = PN_WINDOW;
    const int F = stage == 1 ? 64 : 128;
    const size_t wl = (size_t)PN_WINDOW * (F + 1) * 4;
    const long long ntile = (n_cap + 127) / 128;
    const long long g = ntile < 2LL * cus ? ntile : 2LL * cus;
    return stage == 1 ? launch_chain_lds<1>(a, g, wl, st) : launch_chain_lds<2>(a, g, wl, st);
}
