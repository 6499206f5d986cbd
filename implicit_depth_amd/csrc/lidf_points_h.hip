// lidf_points_h.hip — the fused per-point query with split-f16 matrix instructions (gfx950 only).
//
// Same computation and same register-chained, transposed formulation as lidf_points.hip (FUSED
// mode), but every f32 product  w*x  of layers 1-3 is evaluated as
//     w*x ~= wh*xh + wh*xl + wl*xh ,   w = wh + wl,  x = xh + xl   (f16 pieces, f32 accumulation)
// on v_mfma_f32_32x32x16_f16: three instructions of 32 cycles per 16 k instead of eight f32
// instructions of 64 cycles. The dropped wl*xl term is <= 2^-22 |w x|; measured end-to-end error
// against the oracle is that of the plain f32 kernel. f16 subnormal operands are
// honoured by the instruction (scripts/mfma_f16_ubench.hip), so small low parts are not lost;
// |activations| must stay below the f16 range (65504).
//
// Operand layout of the instruction: lane l supplies 8 consecutive k (8*(l>>5) .. +7) of row/column
// l&31; the result layout is the 32x32 f32 one. A result tile (16 registers per lane, features
// F(r,h) = (r&3) + 8(r>>2) + 4h) therefore splits into two k-sub-steps: registers 0..7 and 8..15.
//
// Everything is k-outer: a layer-1 output tile (32 features x 32 points) is produced — voxel part
// gathered as the accumulator's initial value, ray part as a rank-1 update, positional encoding
// as NK1 k-steps, IEF term as one more instruction — then activated, split and consumed by all
// layer-2 output tiles at once, so no tile of layer-1 or layer-2 outputs is ever held beyond its
// use and a wavefront needs < 256 registers: two workgroups per CU, whose matrix, vector and memory
// instructions overlap. The second IEF iteration recomputes layer 1 (+168 matrix instructions)
// instead of holding 128 registers across the pass.
//
// Weight stream ("H" layout): one section of 288 quads (1 KiB: 64 lanes x 8 halves) per decoder,
// in exactly the order pass_desc() gives; an IEF decoder's section is consumed once per
// iteration. The four wavefronts of a workgroup consume the stream in lockstep; it is staged once
// per workgroup through LDS in chunks of 16 quads (3 rotating buffers, one s_barrier per chunk,
// global loads issued one chunk ahead), and each wavefront keeps a 4-quad register ring of
// ds_read_b128 in flight.
#include "lidf_device.h"
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

#define MFMAH(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, (a)), __builtin_bit_cast(h8, (b)), (c), 0, 0, 0)
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define LDQ(rs, voff, soff) \
    __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128((rs), (voff), (soff), 0))

// development only: per-phase cycle counters of wavefront 0 of workgroup 0, written to a.out_base
#ifdef LIDF_PROFILE
#define PROF_DECL long long prof_t = clock64(); long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const long long prof_w0 = wall_clock64(), prof_c0 = prof_t;
#define PROF(i) { long long t_ = clock64(); prof_acc[i] += t_ - prof_t; prof_t = t_; }
#define PROF_DUMP if (threadIdx.x == 0 && a.out_base) { long long* o_ = (long long*)a.out_base + 16 + 4 * blockIdx.x; o_[0] = prof_w0; o_[1] = wall_clock64(); o_[2] = __builtin_amdgcn_s_getreg(63492); o_[3] = __builtin_amdgcn_s_getreg(63508); } \
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.out_base) { prof_acc[6] = wall_clock64() - prof_w0; prof_acc[7] = clock64() - prof_c0; for (int i_ = 0; i_ < 8; ++i_) ((long long*)a.out_base)[i_] = prof_acc[i_]; }
#else
#define PROF_DECL
#define PROF(i)
#define PROF_DUMP
#endif


#define CH_QUADS 16
#define CH_ELEMS (CH_QUADS * 64)  // f32x4 elements per chunk buffer
#define NBUF 3
#ifndef RING_D
#define RING_D 4  // quads of the stream in flight between LDS and the matrix instructions
#endif
#define NK1 LIDF_H_NK1            // layer-1 k-steps of 16: 6 of sin/cos (8 octaves) + raw x,y,z
#define PASS_QUADS LIDF_HPASS_QUADS
#define LDS_STREAM_ELEMS (NBUF * CH_ELEMS)
// + the tile-grab slot + 7 floats per lane of per-tile geometry parked outside the register file
// (28 B/lane: with it two workgroups use 163,632 of the CU's 163,840 B)
#define LDS_STASH_FLOATS 7
#define LDS_BYTES ((LDS_STREAM_ELEMS + 4 * NK1 * 64 + 1) * 16 + 4 * LDS_STASH_FLOATS * 64 * 4)
#define LIDF_H_GRAB 2

// ------------------------------------------------------------------------------------------------
// Order of a decoder's section (shared by the packer and the kernel)
// ------------------------------------------------------------------------------------------------
enum { K_B2 = 0, K_L1, K_XA, K_XB, K_L2, K_B3, K_L3, K_PAD };
struct QD {
    int kind;
    int t;    // output tile
    int T;    // layer 1 / u: the H1 tile produced; layers 2, 3: the input tile consumed
    int sub;  // layer 1: k-step; layers 2, 3: k-sub-step inside the input tile
    int lo;   // 0: high pieces of the weights, 1: low pieces
    int j;    // ordinal of the (hi, lo) pair inside its segment
};
__host__ __device__ constexpr QD l1u_desc(int T, int r) {
    if (r < 2 * NK1) return {K_L1, 0, T, r >> 1, r & 1, r >> 1};
    return {r == 2 * NK1 ? K_XA : K_XB, 0, T, 0, 0, 0};
}
__host__ __device__ constexpr QD pass_desc(int s) {
    constexpr int L1U = 2 * NK1 + 2;
    if (s < 4) return {K_B2, s, 0, 0, 0, 0};
    s -= 4;
    if (s < L1U) return l1u_desc(0, s);
    s -= L1U;
    if (s < 7 * (L1U + 16)) {
        const int T = s / (L1U + 16);
        int r = s % (L1U + 16);
        if (r < L1U) return l1u_desc(T + 1, r);  // layer 1 runs one tile ahead of layer 2
        r -= L1U;
        const int lo = r & 1, j = r >> 1;
        return {K_L2, j & 3, T, j >> 2, lo, j};
    }
    s -= 7 * (L1U + 16);
    if (s < 16) {
        const int lo = s & 1, j = s >> 1;
        return {K_L2, j >> 1, 7, j & 1, lo, j};  // last segment: tile-major, tiles finish one by one
    }
    s -= 16;
    if (s < 2) return {K_B3, s, 0, 0, 0, 0};
    s -= 2;
    if (s < 32) {
        const int T = s / 8, r = s % 8, lo = r & 1, j = r >> 1;
        if (T < 3) return {K_L3, j & 1, T, j >> 1, lo, j};
        return {K_L3, j >> 1, T, j & 1, lo, j};
    }
    return {K_PAD, 0, 0, 0, 0, 0};
}
static_assert(4 + 8 * (2 * NK1 + 2) + 8 * 16 + 2 + 32 <= PASS_QUADS, "section size");
static_assert(PASS_QUADS % CH_QUADS == 0, "sections are whole chunks");
static_assert(PASS_QUADS % RING_D == 0 && CH_QUADS - RING_D > 6, "ring slots are static; ring reads of the next chunk start after the barrier");

__host__ __device__ constexpr int tile_feature(int r, int half) {
    return (r & 3) + 8 * (r >> 2) + 4 * half;
}

// ------------------------------------------------------------------------------------------------
// Packer: one thread per f16 element.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ _Float16 hpiece(float w, int part) {
    const _Float16 hi = (_Float16)w;
    if (part == 0) return hi;
    const float r = w - (float)hi;
    const _Float16 lo = (_Float16)r;
    if (part == 1) return lo;
    return (_Float16)(r - (float)lo);
}

__device__ _Float16 stream_value_h(const StreamLayout& lay, const NetW* nets, const L1Map& m,
                                   long long e) {
    const int per_net = lay.net_quads * 512;
    const int net = (int)(e / per_net);
    e %= per_net;
    const NetW& n = nets[net];
    const int quad = (int)(e / 512);
    const int lane = (int)(e % 512) / 8, i = (int)(e & 7);
    const int half = lane >> 5, o = lane & 31;
    const QD d = pass_desc(quad);
    switch (d.kind) {
        case K_L1: {
            // k-step ks: 4 (octave, coordinate) combos 4ks..4ks+3, element 2c' = sin, 2c'+1 = cos;
            // lanes 0..31 weight the enter position's embedding, lanes 32..63 the leave position's
            const int c = 4 * d.sub + (i >> 1), oct = c / 3, dd = c % 3;
            if (oct >= m.L) return (_Float16)0.f;
            const int col = (half ? m.leave_c0 : m.enter_c0) + 3 + 6 * oct + 3 * (i & 1) + dd;
            return hpiece(n.w1[(size_t)(32 * d.T + o) * n.ld1 + col], d.lo);
        }
        case K_XA: {
            // raw x, y, z: high weight pieces twice, against (xh, yh, zh, xl, yl, zl, 0, 0)
            if (i >= 6) return (_Float16)0.f;
            const int col = (half ? m.leave_c0 : m.enter_c0) + i % 3;
            return hpiece(n.w1[(size_t)(32 * d.T + o) * n.ld1 + col], 0);
        }
        case K_XB: {
            // (wlx, wly, wlz, uh, ray_hi, ray_lo, uh, ul) against (xh, yh, zh, vh, m, m, vl, vh):
            // low weight pieces of x, y, z; the IEF term u*val (lanes 0..31 only); elements 4, 5
            // are filled in by the kernel with the raypart row of this half's ray
            if (i < 3) {
                const int col = (half ? m.leave_c0 : m.enter_c0) + i;
                return hpiece(n.w1[(size_t)(32 * d.T + o) * n.ld1 + col], 1);
            }
            if (i == 4 || i == 5 || half != 0 || !n.is_ief) return (_Float16)0.f;
            const int out = 32 * d.T + o;
            float u = 0.f;
            for (int j = 0; j < 16; ++j) u += n.w1[(size_t)out * n.ld1 + n.dcore + j] * n.wenc[j];
            return hpiece(u, i == 7 ? 1 : 0);
        }
        case K_B2:
            return half == 0 && i < 3 ? hpiece(n.b2[32 * d.t + o], i) : (_Float16)0.f;
        case K_B3:
            return half == 0 && i < 3 ? hpiece(n.b3[32 * d.t + o], i) : (_Float16)0.f;
        case K_L2:
            return hpiece(n.w2[(size_t)(32 * d.t + o) * LIDF_H1 + 32 * d.T +
                               tile_feature(8 * d.sub + i, half)], d.lo);
        case K_L3:
            return hpiece(n.w3[(size_t)(32 * d.t + o) * LIDF_H2 + 32 * d.T +
                               tile_feature(8 * d.sub + i, half)], d.lo);
        default:
            return (_Float16)0.f;
    }
}

__global__ void lidf_pack_h_kernel(StreamLayout lay, NetW net0, NetW net1, L1Map m,
                                   _Float16* stream, float* aux) {
    if (lay.guard && lay.guard->dirty == 0) return;   // guarded packing: fingerprint unchanged
    NetW nets[2] = {net0, net1};
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (long long)lay.total * 2) stream[e] = stream_value_h(lay, nets, m, e);
    if (e < lay.nets * LIDF_AUX_FLOATS) {
        const int sec = (int)e / LIDF_AUX_FLOATS, i = (int)e % LIDF_AUX_FLOATS;
        const NetW& n = nets[sec];
        float v = 0.f;
        if (i < 64) {
            const int half = i / 32, s = i % 32;
            v = n.w4[32 * (s >> 4) + tile_feature(s & 15, half)];
        } else if (i == 64) {
            v = n.b4[0];
        }
        aux[e] = v;
    }
    if (e == 0) ((int*)aux)[lay.nets * LIDF_AUX_FLOATS] = 0;  // tile counter of lidf_points_h_kernel
}

extern "C" hipError_t lidf_launch_pack_h(const StreamLayout& lay, const NetW& n0, const NetW& n1,
                                         const L1Map& m, float* stream, float* aux,
                                         hipStream_t st) {
    const long long total = (long long)lay.total * 2;
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(lidf_pack_h_kernel, dim3(blocks), dim3(256), 0, st, lay, n0, n1, m,
                       (_Float16*)stream, aux);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Kernel helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float pack2(const _Float16 a, const _Float16 b) {
    h2 v;
    v[0] = a;
    v[1] = b;
    return __builtin_bit_cast(float, v);
}

// (x0, x1) -> packed f16 high pieces (round to nearest) and packed f16 residuals. x - hi is exact
// in f32; v_fma_mix_f32 reads the f16 half directly (1 instruction per residual).
__device__ __forceinline__ void split2(const float x0, const float x1, float& hi, float& lo) {
    h2 hh;
    hh[0] = (_Float16)x0;
    hh[1] = (_Float16)x1;
    const float hw = __builtin_bit_cast(float, hh);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hw), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=v"(r1)
        : "v"(hw), "v"(x1));
    h2 ll;
    ll[0] = (_Float16)r0;
    ll[1] = (_Float16)r1;
    hi = hw;
    lo = __builtin_bit_cast(float, ll);
}

// x -> one word holding (hi piece, lo piece) of x
__device__ __forceinline__ float split1(const float x) {
    const _Float16 hi = (_Float16)x;
    const _Float16 lo = (_Float16)(x - (float)hi);
    return pack2(hi, lo);
}

__device__ __forceinline__ float lrelu1(const float x) {
    const float t = x * 0.02f;
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(t));
    return r;
}

// pairs [p0, p1) of a result tile: leaky-relu, split, store into the two k-sub-step fragments
// (p0, p1 are compile-time constants after unrolling)
__device__ __forceinline__ void prep_pairs(const int p0, const int p1, const f32x16& pre,
                                           f32x4 (&bh)[2], f32x4 (&bl)[2]) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        if (p >= p0 && p < p1) {
            float hi, lo;
            split2(lrelu1(pre[2 * p]), lrelu1(pre[2 * p + 1]), hi, lo);
            bh[p >> 2][p & 3] = hi;
            bl[p >> 2][p & 3] = lo;
        }
    }
}

__device__ __forceinline__ float out_act_h(float y, int use_sigmoid) {
    // implicit_net.py:93-96 / :148-151
    if (use_sigmoid) return 1.f / (1.f + expf(-y));
    return fmaxf(fminf(y, y * 0.01f + 0.99f), y * 0.01f);
}

// sin/cos of x*2^o in revolutions, see lidf_points.hip
struct RevH {
    float hi, lo;
};
__device__ __forceinline__ RevH to_rev_h(float x) {
    const float C_HI = 0.15915493667125702f;
    const float C_LO = 6.4206383e-09f;
    RevH r;
    r.hi = x * C_HI;
    r.lo = fmaf(x, C_HI, -r.hi) + x * C_LO;
    return r;
}
__device__ __forceinline__ void rev_sincos_h(const RevH& r, float sc, float& s, float& c) {
    const float t = __builtin_amdgcn_fractf(r.hi * sc) + r.lo * sc;
    s = __builtin_amdgcn_sinf(t);
    c = __builtin_amdgcn_cosf(t);
}

// The stream feed of one wavefront.
struct Feed {
    f32x4 ring[RING_D];  // next RING_D quads
    f32x4 stage[4];  // this wavefront's quarter of the chunk after the next one, in flight
    int cur, nxt;    // element index (f32x4 units) of this lane in the current / next LDS buffer
    int nb;          // index of the next buffer
    int s_net, s_pass, s_idx;  // sequencer: which chunk the next global load fetches
};

struct FeedCfg {
    __amdgpu_buffer_rsrc_t srs;
    int vq;         // lane * 16
    int wave, lane;
    int net_bytes, nets;
    int npass0, npass1;
};

__device__ __forceinline__ void feed_issue(Feed& f, const FeedCfg& c) {
    const int off = f.s_net * c.net_bytes + f.s_idx * (CH_QUADS * 1024) + c.wave * 4096;
#pragma unroll
    for (int j = 0; j < 4; ++j) f.stage[j] = LDQ(c.srs, c.vq + j * 1024, off);
    // advance the sequencer: a decoder's section is consumed once per pass
    if (++f.s_idx == PASS_QUADS / CH_QUADS) {
        f.s_idx = 0;
        const int np = f.s_net ? c.npass1 : c.npass0;
        if (++f.s_pass == np) {
            f.s_pass = 0;
            if (++f.s_net == c.nets) f.s_net = 0;
        }
    }
}

// position Q (0..15) of the current chunk: returns the quad, refills the ring four quads ahead,
// and at mid-chunk publishes the staged chunk to LDS and starts the next global fetch
__device__ __forceinline__ f32x4 feed_take(Feed& f, const FeedCfg& c, f32x4* sb, const int Q,
                                            const int slot) {
    // slot = (position in the section) % RING_D; sections are multiples of RING_D quads
    const f32x4 a = f.ring[slot];
    if (Q + RING_D < CH_QUADS)
        f.ring[slot] = sb[f.cur + (Q + RING_D) * 64];
    else
        f.ring[slot] = sb[f.nxt + (Q + RING_D - CH_QUADS) * 64];
    if (Q == 4) {
        // The buffer written here last held the chunk before the previous one: every wavefront
        // finished reading it before it passed the previous barrier.
#pragma unroll
        for (int j = 0; j < 4; ++j) sb[f.nb * CH_ELEMS + (4 * c.wave + j) * 64 + c.lane] = f.stage[j];
        feed_issue(f, c);
    }
    if (Q == 6) {
        // LDS operations complete in order and at least two (the ring reads of positions 5 and 6)
        // were issued after the four writes of position 4: once at most two are outstanding the
        // writes have landed, and the ring need not drain. The first read of the next buffer
        // comes at position CH_QUADS - RING_D > 6.
        asm volatile("s_waitcnt lgkmcnt(2)\n\ts_barrier" ::: "memory");
    }
    if (Q == CH_QUADS - 1) {
        f.cur = f.nxt;
        f.nb = f.nb == NBUF - 1 ? 0 : f.nb + 1;
        f.nxt = f.nb * CH_ELEMS + c.lane;
    }
    return a;
}

// What one pass needs to know about the points of this wavefront.
struct TileCtx {
    const float* vp;     // voxpart row of this lane's point (+ net, + 4*half)
    const float* rp;     // raypart row of this half's ray of round 0 (+ net, + lane&31)
    float rayB;          // packed (1,1) if this lane's point belongs to this half's ray
    int ray;             // this lane's ray (for the extra rounds)
    unsigned todo;       // points not covered by round 0 (0 in the common case)
    const float* rbase;  // raypart + net*256 + lane&31 (extra rounds)
    int ray_stride;      // nets*256
    int pl;              // LDS element index of this lane's low pieces of the embedding operands
};

// One decoder pass on the 32 points of this wavefront:
//   H1 = lrelu(W1 x + b1 [+ u*val]);  H2 = lrelu(W2 H1 + b2);  H3 = lrelu(W3 H2 + b3);  y = w4.H3 + b4
// with W1 x = voxpart[voxel] + raypart[ray] + W1[:, enter|leave] PE(position).
#ifdef LIDF_PROFILE_CHUNKS
__device__ long long g_chunk_ticks[32];
#define CHUNK_PROF(q, s_) if ((q) == CH_QUADS - 1) { const long long t_ = clock64(); if (blockIdx.x == 0 && threadIdx.x == 0) g_chunk_ticks[(s_) / CH_QUADS] += t_ - chunk_t; chunk_t = t_; }
#else
#define CHUNK_PROF(q, s_)
#endif
__device__ __forceinline__ float decoder_pass_h(Feed& f, const FeedCfg& c, f32x4* sb,
                                                const TileCtx& tc, const f32x4 (&pbh)[NK1],
                                                const f32x4 xaB, const float xyB, const _Float16 zh,
                                                const float val, const int h, const int col,
                                                const float* __restrict__ ax) {
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                           0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const _Float16 one = (_Float16)1.f, hz = (_Float16)0.f;
    f32x4 onesB = zero4, xbB = zero4;
    xbB[0] = xyB;
    if (!h) {
        onesB[0] = pack2(one, one);
        onesB[1] = pack2(one, hz);
    }
    {
        // (xh, yh, zh, vh, m, m, vl, vh): raw position, IEF input value, ray membership
        const _Float16 vh = (_Float16)val;
        const _Float16 vl = (_Float16)(val - (float)vh);
        xbB[1] = pack2(zh, vh);
        xbB[2] = tc.rayB;
        xbB[3] = pack2(vl, vh);
    }

    f32x16 a1[2];               // layer-1 tiles, ping-pong: accumulating / being split
    float rpv[2];               // raypart value of this lane's row, same ping-pong
    f32x16 acc2[4], acc3[2];
    f32x4 bh[2], bl[2];         // split H1 tile, [k-sub-step] (single buffer, see below)
    f32x4 gh[4][2], gl[4][2];   // split H2 tiles
    f32x4 pl_cur = zero4, pl_nxt = zero4;  // low pieces of the embedding operand, from LDS
    f32x4 w4[8];
    float b4 = 0.f;
    float ys[4] = {0.f, 0.f, 0.f, 0.f};

    // voxel part of tile T straight into the accumulator layout; ray part of this lane's row
    auto gather = [&](const int T) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *(const f32x4*)(tc.vp + T * 32 + 8 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i) a1[T & 1][4 * g + i] = v[i];
        }
        rpv[T & 1] = tc.rp[T * 32];
    };
    gather(0);
    gather(1);
    pl_cur = sb[tc.pl];
#ifdef LIDF_PROFILE_CHUNKS
    long long chunk_t = clock64();
#endif

#pragma unroll
    for (int s = 0; s < PASS_QUADS; ++s) {
        const int Q = s % CH_QUADS;
        const QD d = pass_desc(s);
        const f32x4 A = feed_take(f, c, sb, Q, s % RING_D);
        if (d.kind == K_B2) {
            acc2[d.t] = MFMAH(A, onesB, zero16);
        } else if (d.kind == K_L1) {
            f32x16& acc = a1[d.T & 1];
            if (!d.lo) {
                acc = MFMAH(A, pbh[d.sub], acc);
                acc = MFMAH(A, pl_cur, acc);
                // low pieces of the next k-step's operand (wraps to k-step 0 for the next tile)
                pl_nxt = sb[tc.pl + ((d.sub + 1) % NK1) * 64];
                // split the previous tile behind these matrix instructions (8 pairs over the
                // NK1 k-steps and the two mixed steps that follow)
                if (d.T >= 1) prep_pairs(d.sub, d.sub + 1, a1[(d.T - 1) & 1], bh, bl);
            } else {
                acc = MFMAH(A, pbh[d.sub], acc);
                pl_cur = pl_nxt;
            }
        } else if (d.kind == K_XA) {
            a1[d.T & 1] = MFMAH(A, xaB, a1[d.T & 1]);
            if (d.T >= 1) prep_pairs(NK1, NK1 + 1, a1[(d.T - 1) & 1], bh, bl);
        } else if (d.kind == K_XB) {
            // elements 4, 5 of the weights fragment <- (hi, lo) of the raypart row of this half's
            // ray: the ray part of layer 1 is a rank-1 update, two rays per instruction. Tiles that
            // straddle more than two rays (ragged scenes) take extra rounds.
            f32x16& acc = a1[d.T & 1];
            f32x4 Ar = A;
            Ar[2] = split1(rpv[d.T & 1]);
            acc = MFMAH(Ar, xbB, acc);
            if (d.T >= 1) prep_pairs(NK1 + 1, 8, a1[(d.T - 1) & 1], bh, bl);
            if (tc.todo) {
                unsigned todo = tc.todo;
                while (todo) {
                    const int p0 = __builtin_ctz(todo);
                    const int r0 = __builtin_amdgcn_readlane(tc.ray, p0);
                    const unsigned m0 = (unsigned)__ballot(tc.ray == r0) & todo;
                    todo &= ~m0;
                    int r1 = r0;
                    unsigned m1 = 0;
                    if (todo) {
                        const int p1 = __builtin_ctz(todo);
                        r1 = __builtin_amdgcn_readlane(tc.ray, p1);
                        m1 = (unsigned)__ballot(tc.ray == r1) & todo;
                        todo &= ~m1;
                    }
                    const float v = tc.rbase[(size_t)(h ? r1 : r0) * tc.ray_stride + d.T * 32];
                    f32x4 xa = zero4, xb = zero4;
                    xa[2] = split1(v);
                    xb[2] = (((h ? m1 : m0) >> col) & 1u) ? pack2(one, one) : 0.f;
                    acc = MFMAH(xa, xb, acc);
                }
            }
        } else if (d.kind == K_L2) {
            if (!d.lo) {
                // the tile after next can start loading: its accumulator is free once the split
                // of tile T (same parity) is done
                if (d.j == 0 && d.T + 2 < 8) gather(d.T + 2);
                // tile 7 has no layer-1 segment after it to hide its split behind, and a second
                // operand buffer just for it would push the kernel into spilling
                if (d.T == 7 && d.j == 0) prep_pairs(0, 8, a1[1], bh, bl);
                acc2[d.t] = MFMAH(A, bh[d.sub], acc2[d.t]);
                acc2[d.t] = MFMAH(A, bl[d.sub], acc2[d.t]);
                if (d.T == 7) {
                    // last segment (tile-major): output tiles complete one by one
                    if (d.j >= 2 && d.j < 6) prep_pairs(2 * (d.j - 2), 2 * (d.j - 2) + 2, acc2[0], gh[0], gl[0]);
                    if (d.j >= 6) prep_pairs(2 * (d.j - 6), 2 * (d.j - 6) + 2, acc2[1], gh[1], gl[1]);
                }
            } else {
                acc2[d.t] = MFMAH(A, bh[d.sub], acc2[d.t]);
            }
        } else if (d.kind == K_B3) {
            acc3[d.t] = MFMAH(A, onesB, zero16);
            if (d.t == 0) b4 = ax[64];
        } else if (d.kind == K_L3) {
            if (!d.lo) {
                // operands of the tail (layer 4), fetched late and in two halves so that they do not
                // hold 32 registers through layer 3; their latency still hides behind it
                if (d.T == 2 && d.j == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) w4[i] = *(const f32x4*)(ax + h * 32 + 4 * i);
                }
                if (d.T == 3 && d.j == 0) {
#pragma unroll
                    for (int i = 4; i < 8; ++i) w4[i] = *(const f32x4*)(ax + h * 32 + 4 * i);
                }
                acc3[d.t] = MFMAH(A, gh[d.T][d.sub], acc3[d.t]);
                acc3[d.t] = MFMAH(A, gl[d.T][d.sub], acc3[d.t]);
                // pending splits: segment T handles the second half of H2[T+1] (first two pairs)
                // and the first half of H2[T+2] (last two pairs)
                if (d.T < 3 && d.j < 2) prep_pairs(4 + 2 * d.j, 6 + 2 * d.j, acc2[d.T + 1], gh[d.T + 1], gl[d.T + 1]);
                if (d.T < 2 && d.j >= 2) prep_pairs(2 * (d.j - 2), 2 * (d.j - 2) + 2, acc2[d.T + 2], gh[d.T + 2], gl[d.T + 2]);
                if (d.T == 3 && d.j >= 2) {
                    // acc3[0] is complete: its half of layer 4 runs behind acc3[1]'s last steps
#pragma unroll
                    for (int r = 8 * (d.j - 2); r < 8 * (d.j - 2) + 8; ++r)
                        ys[r & 3] = fmaf(w4[r >> 2][r & 3], lrelu1(acc3[0][r]), ys[r & 3]);
                }
            } else {
                acc3[d.t] = MFMAH(A, gh[d.T][d.sub], acc3[d.t]);
            }
        }
        CHUNK_PROF(Q, s)
        SCHED_FENCE();
    }
#ifdef LIDF_PROFILE_CHUNKS
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_chunk_ticks[31] += 1; }
#endif
    // layer 4 (64 -> 1) on the VALU, second tile; halves combined with one cross-half shuffle
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int sidx = 16 + r;
        ys[r & 3] = fmaf(w4[sidx >> 2][sidx & 3], lrelu1(acc3[1][r]), ys[r & 3]);
    }
    float y = (ys[0] + ys[1]) + (ys[2] + ys[3]);
    y += __shfl_xor(y, 32);
    return y + b4;
}

struct GeoH {
    int ray, vid;
    float te, tl, dx, dy, dz;
};

__global__ void __launch_bounds__(256, 2) lidf_points_h_kernel(PointsArgs a) {
    extern __shared__ f32x4 sb[];  // [NBUF chunks of the stream][per wavefront: NK1 operand quads]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;
    const int col = lane & 31;

    FeedCfg c;
    c.srs = __builtin_amdgcn_make_buffer_rsrc((void*)a.stream, 0, a.nets * a.net_quads * 1024,
                                              0x00020000);
    c.vq = lane * 16;
    c.wave = wave;
    c.lane = lane;
    c.net_bytes = a.net_quads * 1024;
    c.nets = a.nets;
    c.npass0 = a.npass[0];
    c.npass1 = a.npass[1];

    // 128-point tiles are handed out dynamically, LIDF_H_GRAB at a time, from a counter the packer
    // zeroes: the two workgroups of a CU do not progress at the same rate (the older one wins the
    // issue arbitration), and with a static split the younger one would run the tail alone.
    // Every wavefront of the workgroup runs the same tiles (the stream is shared); out-of-range
    // points are clamped.
    const long long AN = a.n_dev ? (long long)*a.n_dev : a.n;   // device-side count (the frame path)
    const long long ntile = (AN + 127) / 128;
    int* const grab_slot = (int*)(sb + LDS_STREAM_ELEMS + 4 * NK1 * 64);
    // geometry that is only needed again at the end of the tile (the offset net's output, the next
    // tile's indices) waits in LDS instead of occupying registers through every pass
    float* const stash = (float*)(sb + LDS_STREAM_ELEMS + 4 * NK1 * 64 + 1) +
                         wave * (LDS_STASH_FLOATS * 64) + lane;

    // prologue: chunk 0 into buffer 0, chunk 1 in flight
    Feed f;
    f.s_net = 0;
    f.s_pass = 0;
    f.s_idx = 0;
    feed_issue(f, c);
#pragma unroll
    for (int j = 0; j < 4; ++j) sb[(4 * wave + j) * 64 + lane] = f.stage[j];
    feed_issue(f, c);
    __syncthreads();
    f.cur = lane;
    f.nb = 1;
    f.nxt = CH_ELEMS + lane;
#pragma unroll
    for (int i = 0; i < RING_D; ++i) f.ring[i] = sb[f.cur + i * 64];

    // addresses = wave-uniform base (SGPRs) + a 32-bit lane offset: no 64-bit pointer registers
    auto load_idx = [&](long long tile, GeoH& g) {
        const long long last = AN - 1;
        long long t0 = tile * 128;                  // uniform
        if (t0 > last) t0 = last & ~127LL;          // a prefetch past the end re-reads the last tile
        const long long rem = last - t0;
        const int lo = wave * 32 + col;
        const unsigned off = (unsigned)(lo < rem ? lo : rem);  // out-of-range points are clamped
        g.ray = (a.pair_ray + t0)[off];
        g.vid = (a.pair_vox + t0)[off];
        const f32x2 tt = *(const f32x2*)((const char*)(a.pair_t + 2 * t0) + 8u * off);
        g.te = tt[0];
        g.tl = tt[1];
    };
    auto load_dir = [&](GeoH& g) {
        const char* rd = (const char*)a.ray_dir + 12u * (unsigned)g.ray;
        g.dx = *(const float*)(rd + 0);
        g.dy = *(const float*)(rd + 4);
        g.dz = *(const float*)(rd + 8);
    };
    GeoH cur = {}, nxt = {}, nx2 = {};

    TileCtx tc;
    tc.pl = LDS_STREAM_ELEMS + wave * NK1 * 64 + lane;
    tc.ray_stride = a.nets * 256;

    PROF_DECL
    for (;;) {
    if (threadIdx.x == 0) *grab_slot = atomicAdd(a.tile_counter, LIDF_H_GRAB);
    __syncthreads();
    const long long tb = __builtin_amdgcn_readfirstlane(*grab_slot);
    if (tb >= ntile) break;
    const long long te_ = tb + LIDF_H_GRAB < ntile ? tb + LIDF_H_GRAB : ntile;
    load_idx(tb, cur);
    load_idx(tb + 1, nxt);
    load_dir(cur);
    for (long long tile = tb; tile < te_; ++tile) {
        PROF(0)
        const long long p = tile * 128 + wave * 32 + col;
        const bool valid = p < AN;
        load_dir(nxt);
        load_idx(tile + 2, nx2);
        // this half's embedding input: lanes 0..31 embed the enter position, lanes 32..63 the leave
        // position (pipeline.py:349-360; 'rel' subtracts the voxel centre)
        const float tt = h ? cur.tl : cur.te;
        float px = __fmul_rn(cur.dx, tt);
        float py = __fmul_rn(cur.dy, tt);
        float pz = __fmul_rn(cur.dz, tt);
        if (a.pos_rel) {
            px -= a.vox_center[3 * (size_t)cur.vid + 0];
            py -= a.vox_center[3 * (size_t)cur.vid + 1];
            pz -= a.vox_center[3 * (size_t)cur.vid + 2];
        }
        stash[0 * 64] = cur.te;
        stash[1 * 64] = cur.dx;
        stash[2 * 64] = cur.dy;
        stash[3 * 64] = cur.dz;
        stash[4 * 64] = nxt.te;
        stash[5 * 64] = nxt.tl;
        stash[6 * 64] = __int_as_float(nxt.vid);
        // operands of the layer-1 k-steps, shared by all passes of this tile: high pieces in
        // registers, low pieces in this wavefront's LDS rows
        f32x4 pbh[NK1];
        {
            const RevH rv[3] = {to_rev_h(px), to_rev_h(py), to_rev_h(pz)};
#pragma unroll
            for (int ks = 0; ks < NK1; ++ks) {
                f32x4 lo4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int idx = 4 * ks + i;  // combo: octave idx/3, coordinate idx%3
                    float sv, cv, hi, lo;
                    rev_sincos_h(rv[idx % 3], (float)(1 << (idx / 3)), sv, cv);
                    split2(sv, cv, hi, lo);
                    pbh[ks][i] = hi;
                    lo4[i] = lo;
                }
                sb[tc.pl + ks * 64] = lo4;
            }
        }
        // raw position against the two mixed k-steps: (xh, yh, zh, xl, yl, zl, 0, 0) and the
        // (xh, yh, zh, .) head of the second one. (zh travels as a scalar: extracting it from a
        // packed word made hipcc 7.2 pick the low half of the wrong register.)
        f32x4 xaB = {0.f, 0.f, 0.f, 0.f};
        float xyB;
        const _Float16 zh = (_Float16)pz;
        {
            const _Float16 xh = (_Float16)px, yh = (_Float16)py;
            const _Float16 xl = (_Float16)(px - (float)xh), yl = (_Float16)(py - (float)yh),
                           zl = (_Float16)(pz - (float)zh);
            xaB[0] = pack2(xh, yh);
            xaB[1] = pack2(zh, xl);
            xaB[2] = pack2(yl, zl);
            xyB = pack2(xh, yh);
        }
        // rays of this wavefront's 32 points: round 0 covers the first two distinct rays
        int my_ray;
        {
            unsigned todo = (unsigned)__ballot(h == 0);
            const int p0 = __builtin_ctz(todo);
            const int r0 = __builtin_amdgcn_readlane(cur.ray, p0);
            const unsigned m0 = (unsigned)__ballot(cur.ray == r0) & todo;
            todo &= ~m0;
            int r1 = r0;
            unsigned m1 = 0;
            if (todo) {
                const int p1 = __builtin_ctz(todo);
                r1 = __builtin_amdgcn_readlane(cur.ray, p1);
                m1 = (unsigned)__ballot(cur.ray == r1) & todo;
                todo &= ~m1;
            }
            my_ray = h ? r1 : r0;
            tc.rayB = (((h ? m1 : m0) >> col) & 1u) ? pack2((_Float16)1.f, (_Float16)1.f) : 0.f;
            tc.todo = todo;
            tc.ray = cur.ray;
        }
        PROF(1)

        for (int net = 0; net < a.nets; ++net) {
            tc.vp = a.voxpart + ((size_t)cur.vid * a.nets + net) * 256 + 4 * h;
            tc.rp = a.raypart + ((size_t)my_ray * a.nets + net) * 256 + col;
            tc.rbase = a.raypart + net * 256 + col;
            float val = a.init[net];
            const float* ax = a.aux + net * LIDF_AUX_FLOATS;
            const int npass = a.npass[net];
            for (int pass = 0; pass < npass; ++pass)
                val += decoder_pass_h(f, c, sb, tc, pbh, xaB, xyB, zh, val, h, col, ax);
            PROF(2 + net)

            // ---------------- outputs ----------------
            if (valid && h == 0) {
                const float o = out_act_h(val, a.sigmoid[net]);
                if (a.out[net]) a.out[net][p] = o;
                if (a.is_offset[net]) {
                    // pipeline.py:437-439, same operation order in f32
                    const float te = stash[0 * 64], dx = stash[1 * 64], dy = stash[2 * 64],
                                dz = stash[3 * 64];
                    const float ex = __fmul_rn(dx, te);
                    const float ey = __fmul_rn(dy, te);
                    const float ez = __fmul_rn(dz, te);
                    float s = __fadd_rn(__fmul_rn(o, a.rscale), a.r0);
                    s = __fmul_rn(__fmul_rn(s, a.sqrt3), a.part_size);
                    a.pair_pred_pos[3 * p + 0] = __fadd_rn(ex, __fmul_rn(s, dx));
                    a.pair_pred_pos[3 * p + 1] = __fadd_rn(ey, __fmul_rn(s, dy));
                    a.pair_pred_pos[3 * p + 2] = __fadd_rn(ez, __fmul_rn(s, dz));
                }
            }
        }
        cur.ray = nxt.ray;
        cur.dx = nxt.dx;
        cur.dy = nxt.dy;
        cur.dz = nxt.dz;
        cur.te = stash[4 * 64];
        cur.tl = stash[5 * 64];
        cur.vid = __float_as_int(stash[6 * 64]);
        nxt.ray = nx2.ray;
        nxt.vid = nx2.vid;
        nxt.te = nx2.te;
        nxt.tl = nx2.tl;
    }
    }
    PROF_DUMP
}

extern "C" hipError_t lidf_launch_points_h(const PointsArgs& a, int cus, hipStream_t st) {
    if (a.n <= 0) return hipSuccess;
    // > 64 KiB of dynamic LDS needs the opt-in once per device (one process may drive several)
    static bool configured[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!configured[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)lidf_points_h_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
        configured[dev] = true;
#ifdef LIDF_PROFILE
        for (int lds = 32768; lds <= LDS_BYTES + 8192; lds += 4096) {
            int nb = -1;
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)lidf_points_h_kernel, 256, lds);
            fprintf(stderr, "occupancy with %d B of LDS per workgroup: %d workgroups per CU\n", lds, nb);
        }
#endif
    }
    // two workgroups per CU
    const long long ntile = (a.n + 127) / 128;
    int grid = (int)(ntile < 2LL * cus ? ntile : 2LL * cus);
#ifdef LIDF_PROFILE
    if (getenv("LIDF_H_SOLO")) grid = cus;
#endif
#ifdef LIDF_PROFILE_CHUNKS
    static long long zero_[32] = {0};
    hipMemcpyToSymbolAsync(HIP_SYMBOL(g_chunk_ticks), zero_, sizeof(zero_), 0, hipMemcpyHostToDevice, st);
#endif
    hipLaunchKernelGGL(lidf_points_h_kernel, dim3(grid), dim3(256), LDS_BYTES, st, a);
#ifdef LIDF_PROFILE_CHUNKS
    {
        long long t_[32];
        hipMemcpyFromSymbol(t_, HIP_SYMBOL(g_chunk_ticks), sizeof(t_), 0, hipMemcpyDeviceToHost);
        fprintf(stderr, "passes of workgroup 0: %lld ; ticks per chunk (avg per pass):", t_[31]);
        for (int i = 0; i < PASS_QUADS / CH_QUADS; ++i) fprintf(stderr, " %lld", t_[31] ? t_[i] / t_[31] : 0);
        fprintf(stderr, "\n");
    }
#endif
    return hipGetLastError();
}
