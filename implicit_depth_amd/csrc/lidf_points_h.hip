// lidf_points_h.hip — the fused per-point query with split-f16 matrix instructions (gfx950 only).
//
// Same computation and same register-chained, transposed formulation as lidf_points.hip (FUSED
// mode), but every f32 product  w*x  of layers 1-3 is evaluated as
//     w*x ~= wh*xh + wh*xl + wl*xh ,   w = wh + wl,  x = xh + xl   (f16 pieces, f32 accumulation)
// on v_mfma_f32_32x32x16_f16: three instructions of 32 cycles per 16 k instead of eight f32
// instructions of 64 cycles. The dropped wl*xl term is <= 2^-22 |w x|; measured end-to-end error
// against an f64 evaluation is that of the plain f32 kernel. f16 subnormal
// operands are honoured by the instruction (scripts/mfma_f16_ubench.hip), so small low parts are
// not lost; |activations| must stay below the f16 range (65504).
//
// Operand layout of the instruction: lane l supplies 8 consecutive k (8*(l>>5) .. +7) of row/column
// l&31; the result layout is the 32x32 f32 one. A result tile (16 registers per lane, features
// F(r,h) = (r&3) + 8(r>>2) + 4h) therefore splits into two k-sub-steps: registers 0..7 and 8..15.
//
// Weight stream ("H" layout), per decoder, in quads of 1 KiB (64 lanes x 8 halves):
//   layer 1 : NK1 k-steps x [for tile t<8: hi(t), lo(t)]            (16 quads = one chunk each)
//             k-step ks < NK1-1 holds 4 (octave, coordinate) combos 4ks..4ks+3, element 2c' = sin,
//             2c'+1 = cos; lanes 0..31 the enter position, lanes 32..63 the leave position;
//             the last k-step holds raw x, y, z
//   pass    : 176 quads, order given by pass_desc() below (k-outer: each H1 / H2 tile is consumed
//             by all output tiles as soon as it has been produced, so only one split tile is live)
// The four wavefronts of a workgroup consume the same stream in lockstep; it is staged once per
// workgroup through LDS in chunks of 16 quads (3 rotating buffers, one s_barrier per chunk,
// global loads issued one chunk ahead), and each wavefront keeps a 4-quad register ring of
// ds_read_b128 in flight.
#include "lidf_device.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

#define MFMAH(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, (a)), __builtin_bit_cast(h8, (b)), (c), 0, 0, 0)
#define MFMAF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define LDQ(rs, voff, soff) \
    __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128((rs), (voff), (soff), 0))

// development only: per-phase cycle counters of wavefront 0 of workgroup 0, written to a.out_base
#ifdef LIDF_PROFILE
#define PROF_DECL long long prof_t = clock64(); long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PROF(i) { long long t_ = clock64(); prof_acc[i] += t_ - prof_t; prof_t = t_; }
#define PROF_DUMP if (blockIdx.x == 0 && threadIdx.x == 0 && a.out_base) { for (int i_ = 0; i_ < 8; ++i_) ((long long*)a.out_base)[i_] = prof_acc[i_]; }
#else
#define PROF_DECL
#define PROF(i)
#define PROF_DUMP
#endif

#define CH_QUADS 16
#define CH_ELEMS (CH_QUADS * 64)  // f32x4 elements per chunk buffer
#define NBUF 3

// ------------------------------------------------------------------------------------------------
// Order of the pass section (shared by the packer and the kernel)
// ------------------------------------------------------------------------------------------------
enum { K_B2 = 0, K_U, K_L2, K_B3, K_L3, K_PAD };
struct QD {
    int kind;
    int t;    // output tile
    int T;    // input tile (H1 tile for layer 2 / u, H2 tile for layer 3)
    int sub;  // k-sub-step inside the input tile
    int lo;   // 0: high pieces of the weights, 1: low pieces
    int j;    // ordinal of the (hi, lo) pair inside its 16- or 8-quad segment
};
__host__ __device__ constexpr QD pass_desc(int s) {
    if (s < 4) return {K_B2, s, 0, 0, 0, 0};
    if (s == 4) return {K_U, 0, 0, 0, 0, 0};
    s -= 5;
    if (s < 7 * 17 + 16) {
        const int T = s / 17 < 7 ? s / 17 : 7;
        int r = s - 17 * T;
        if (T < 7) {
            if (r == 0) return {K_U, 0, T + 1, 0, 0, 0};  // u of the NEXT tile, one segment early
            r -= 1;
        }
        const int lo = r & 1, j = r >> 1;
        if (T < 7) return {K_L2, j & 3, T, j >> 2, lo, j};
        return {K_L2, j >> 1, T, j & 1, lo, j};  // last segment: tile-major, tiles finish one by one
    }
    s -= 7 * 17 + 16;
    if (s < 2) return {K_B3, s, 0, 0, 0, 0};
    s -= 2;
    if (s < 32) {
        const int T = s / 8, r = s % 8, lo = r & 1, j = r >> 1;
        if (T < 3) return {K_L3, j & 1, T, j >> 1, lo, j};
        return {K_L3, j >> 1, T, j & 1, lo, j};
    }
    return {K_PAD, 0, 0, 0, 0, 0};
}

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
    int quad = (int)(e / 512);
    const int lane = (int)(e % 512) / 8, i = (int)(e & 7);
    const int half = lane >> 5, o = lane & 31;
    if (quad < lay.l1_quads) {
        const int ks = quad / 16, r = quad % 16, t = r >> 1, part = r & 1;
        const int nk1 = lay.l1_quads / 16;
        int feat = -1;
        if (ks < nk1 - 1) {
            const int c = 4 * ks + (i >> 1), oct = c / 3, d = c % 3;
            if (oct < m.L) feat = 3 + 6 * oct + 3 * (i & 1) + d;
        } else if (i < 3) {
            feat = i;
        }
        if (feat < 0) return (_Float16)0.f;
        const int col = (half ? m.leave_c0 : m.enter_c0) + feat;
        return hpiece(n.w1[(size_t)(32 * t + o) * n.ld1 + col], part);
    }
    quad -= lay.l1_quads;
    const QD d = pass_desc(quad);
    switch (d.kind) {
        case K_B2:
            return half == 0 && i < 3 ? hpiece(n.b2[32 * d.t + o], i) : (_Float16)0.f;
        case K_B3:
            return half == 0 && i < 3 ? hpiece(n.b3[32 * d.t + o], i) : (_Float16)0.f;
        case K_U: {
            if (half != 0 || i >= 3 || !n.is_ief) return (_Float16)0.f;
            const int out = 32 * d.T + o;
            float u = 0.f;
            for (int j = 0; j < 16; ++j) u += n.w1[(size_t)out * n.ld1 + n.dcore + j] * n.wenc[j];
            return hpiece(u, i == 2 ? 1 : 0);  // (uh, uh, ul) against (vh, vl, vh)
        }
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

__device__ __forceinline__ float lrelu1(const float x) {
    const float t = x * 0.02f;
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(t));
    return r;
}

// pairs [P0, P1) of a result tile: leaky-relu, split, store into the two k-sub-step fragments
template <int P0, int P1>
__device__ __forceinline__ void prep_pairs(const f32x16& pre, f32x4 (&bh)[2], f32x4 (&bl)[2]) {
#pragma unroll
    for (int p = P0; p < P1; ++p) {
        float hi, lo;
        split2(lrelu1(pre[2 * p]), lrelu1(pre[2 * p + 1]), hi, lo);
        bh[p >> 2][p & 3] = hi;
        bl[p >> 2][p & 3] = lo;
    }
}
__device__ __forceinline__ void prep_pairs_dyn(const int p0, const int p1, const f32x16& pre,
                                               f32x4 (&bh)[2], f32x4 (&bl)[2]) {
    // p0, p1 are compile-time constants after unrolling
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
    f32x4 ring[4];   // next four quads
    f32x4 stage[4];  // this wavefront's quarter of the chunk after the next one, in flight
    int cur, nxt;    // element index (f32x4 units) of this lane in the current / next LDS buffer
    int nb;          // index of the next buffer
    int s_net, s_pass, s_idx;  // sequencer: which chunk the next global load fetches
};

struct FeedCfg {
    __amdgpu_buffer_rsrc_t srs;
    int vq;         // lane * 16
    int wave, lane;
    int net_bytes, nk1, nets;
    int npass0, npass1;
};

__device__ __forceinline__ void feed_issue(Feed& f, const FeedCfg& c) {
    const int chunk = f.s_pass < 0 ? f.s_idx : c.nk1 + f.s_idx;
    const int off = f.s_net * c.net_bytes + chunk * (CH_QUADS * 1024) + c.wave * 4096;
#pragma unroll
    for (int j = 0; j < 4; ++j) f.stage[j] = LDQ(c.srs, c.vq + j * 1024, off);
    // advance the sequencer
    const int lim = f.s_pass < 0 ? c.nk1 : LIDF_HPASS_QUADS / CH_QUADS;
    if (++f.s_idx == lim) {
        f.s_idx = 0;
        const int np = f.s_net ? c.npass1 : c.npass0;
        if (++f.s_pass == np) {
            f.s_pass = -1;
            if (++f.s_net == c.nets) f.s_net = 0;
        }
    }
}

// position Q (0..15) of the current chunk: returns the quad, refills the ring four quads ahead,
// and at mid-chunk publishes the staged chunk to LDS and starts the next global fetch
__device__ __forceinline__ f32x4 feed_take(Feed& f, const FeedCfg& c, f32x4* sb, const int Q) {
    const f32x4 a = f.ring[Q & 3];
    if (Q + 4 < CH_QUADS)
        f.ring[Q & 3] = sb[f.cur + (Q + 4) * 64];
    else
        f.ring[Q & 3] = sb[f.nxt + (Q + 4 - CH_QUADS) * 64];
    if (Q == 6) {
        // The buffer written here last held the chunk before the previous one: every wavefront
        // finished reading it before it passed the previous barrier.
#pragma unroll
        for (int j = 0; j < 4; ++j) sb[f.nb * CH_ELEMS + (4 * c.wave + j) * 64 + c.lane] = f.stage[j];
        feed_issue(f, c);
    }
    if (Q == 8) {
        // LDS operations complete in order: once at most two are outstanding (the ring reads of
        // positions 7 and 8) the four writes of position 6 have landed; no need to drain the ring.
        asm volatile("s_waitcnt lgkmcnt(2)\n\ts_barrier" ::: "memory");
    }
    if (Q == CH_QUADS - 1) {
        f.cur = f.nxt;
        f.nb = f.nb == NBUF - 1 ? 0 : f.nb + 1;
        f.nxt = f.nb * CH_ELEMS + c.lane;
    }
    return a;
}

// One decoder pass on the 32 points of this wavefront (see lidf_points.hip:decoder_pass):
//   H1 = lrelu(base + u*val);  H2 = lrelu(W2 H1 + b2);  H3 = lrelu(W3 H2 + b3);  y = w4.H3 + b4
__device__ __forceinline__ float decoder_pass_h(Feed& f, const FeedCfg& c, f32x4* sb,
                                                const f32x16 (&base)[8], const float val,
                                                const int h, const float* __restrict__ ax) {
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                           0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const _Float16 one = (_Float16)1.f, hz = (_Float16)0.f;
    f32x4 onesB = zero4, offB = zero4;
    if (!h) {
        onesB[0] = pack2(one, one);
        onesB[1] = pack2(one, hz);
        const _Float16 vh = (_Float16)val;
        const _Float16 vl = (_Float16)(val - (float)vh);
        offB[0] = pack2(vh, vl);
        offB[1] = pack2(vh, hz);
    }
    f32x16 acc2[4], acc3[2], pre;
    f32x4 bh[2][2], bl[2][2];      // split H1 tile, [parity of T][k-sub-step]
    f32x4 gh[4][2], gl[4][2];      // split H2 tiles
    f32x4 w4[8];
    float b4 = 0.f;
    float ys[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < LIDF_HPASS_QUADS; ++s) {
        const int Q = s % CH_QUADS;
        const QD d = pass_desc(s);
        const f32x4 A = feed_take(f, c, sb, Q);
        if (d.kind == K_B2) {
            acc2[d.t] = MFMAH(A, onesB, zero16);
        } else if (d.kind == K_U) {
            pre = MFMAH(A, offB, base[d.T]);
            if (d.T == 0) prep_pairs<0, 8>(pre, bh[0], bl[0]);
        } else if (d.kind == K_L2) {
            const int par = d.T & 1;
            if (!d.lo) {
                acc2[d.t] = MFMAH(A, bh[par][d.sub], acc2[d.t]);
                acc2[d.t] = MFMAH(A, bl[par][d.sub], acc2[d.t]);
                if (d.T < 7) {
                    // split pair j of the next H1 tile behind these matrix instructions
                    prep_pairs_dyn(d.j, d.j + 1, pre, bh[par ^ 1], bl[par ^ 1]);
                } else {
                    // last segment (tile-major): output tiles complete one by one
                    if (d.j >= 2 && d.j < 6) prep_pairs_dyn(2 * (d.j - 2), 2 * (d.j - 2) + 2, acc2[0], gh[0], gl[0]);
                    if (d.j >= 6) prep_pairs_dyn(2 * (d.j - 6), 2 * (d.j - 6) + 2, acc2[1], gh[1], gl[1]);
                }
            } else {
                acc2[d.t] = MFMAH(A, bh[par][d.sub], acc2[d.t]);
            }
        } else if (d.kind == K_B3) {
            acc3[d.t] = MFMAH(A, onesB, zero16);
            if (d.t == 0) {
                // operands of the tail, fetched here so that their latency hides behind layer 3
#pragma unroll
                for (int i = 0; i < 8; ++i) w4[i] = *(const f32x4*)(ax + h * 32 + 4 * i);
                b4 = ax[64];
            }
        } else if (d.kind == K_L3) {
            if (!d.lo) {
                acc3[d.t] = MFMAH(A, gh[d.T][d.sub], acc3[d.t]);
                acc3[d.t] = MFMAH(A, gl[d.T][d.sub], acc3[d.t]);
                // pending splits: segment T handles the second half of H2[T+1] (first two pairs)
                // and the first half of H2[T+2] (last two pairs)
                if (d.T < 3 && d.j < 2) prep_pairs_dyn(4 + 2 * d.j, 6 + 2 * d.j, acc2[d.T + 1], gh[d.T + 1], gl[d.T + 1]);
                if (d.T < 2 && d.j >= 2) prep_pairs_dyn(2 * (d.j - 2), 2 * (d.j - 2) + 2, acc2[d.T + 2], gh[d.T + 2], gl[d.T + 2]);
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
        SCHED_FENCE();
    }
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

__global__ void __launch_bounds__(256) lidf_points_h_kernel(PointsArgs a) {
    __shared__ f32x4 sb[NBUF * CH_ELEMS];
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
    c.nk1 = a.l1_quads / CH_QUADS;
    c.nets = a.nets;
    c.npass0 = a.npass[0];
    c.npass1 = a.npass[1];
    const int G = (a.L + 3) / 4;  // groups of three layer-1 k-steps (four octaves)

    // contiguous range of 128-point tiles per workgroup; every wavefront of the workgroup runs
    // the same number of tiles (the stream is shared), out-of-range points are clamped
    const long long ntile = (a.n + 127) / 128;
    const long long per = ntile / gridDim.x, rem = ntile % gridDim.x;
    const long long tb = blockIdx.x * per + (blockIdx.x < rem ? blockIdx.x : rem);
    const long long te_ = tb + per + (blockIdx.x < rem ? 1 : 0);
    if (tb >= te_) return;

    // prologue: chunk 0 into buffer 0, chunk 1 in flight
    Feed f;
    f.s_net = 0;
    f.s_pass = -1;
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
    for (int i = 0; i < 4; ++i) f.ring[i] = sb[f.cur + i * 64];

    auto load_idx = [&](long long tile, GeoH& g) {
        long long pc = tile * 128 + wave * 32 + col;
        pc = pc < a.n ? pc : a.n - 1;
        g.ray = a.pair_ray[pc];
        g.vid = a.pair_vox[pc];
        const f32x2 tt = *(const f32x2*)(a.pair_t + 2 * pc);
        g.te = tt[0];
        g.tl = tt[1];
    };
    auto load_dir = [&](GeoH& g) {
        g.dx = a.ray_dir[3 * (size_t)g.ray + 0];
        g.dy = a.ray_dir[3 * (size_t)g.ray + 1];
        g.dz = a.ray_dir[3 * (size_t)g.ray + 2];
    };
    GeoH cur = {}, nxt = {}, nx2 = {};
    load_idx(tb, cur);
    load_idx(tb + 1, nxt);
    load_dir(cur);

    PROF_DECL
    for (long long tile = tb; tile < te_; ++tile) {
        PROF(0)
        const long long p = tile * 128 + wave * 32 + col;
        const bool valid = p < a.n;
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
        const RevH rv[3] = {to_rev_h(px), to_rev_h(py), to_rev_h(pz)};
        PROF(1)

        for (int net = 0; net < a.nets; ++net) {
            f32x16 base[8];
            // ---------------- layer 1 ----------------
            // accumulator init = voxpart[vid] (+ layer-1 bias) gathered per lane, + raypart[ray] as
            // rank-1 f32 updates, exactly as lidf_points.hip does
            unsigned todo = (unsigned)__ballot(h == 0);
            float ar[8], bsel;
            auto next_round = [&]() {
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
                bsel = (((h ? m1 : m0) >> col) & 1u) ? 1.f : 0.f;
                const float* rp = a.raypart + ((size_t)(h ? r1 : r0) * a.nets + net) * 256 + col;
#pragma unroll
                for (int t = 0; t < 8; ++t) ar[t] = rp[t * 32];
            };
            next_round();
            const float* vp = a.voxpart + ((size_t)cur.vid * a.nets + net) * 256 + 4 * h;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = *(const f32x4*)(vp + t * 32 + 8 * g);
#pragma unroll
                    for (int i = 0; i < 4; ++i) base[t][4 * g + i] = v[i];
                }
            }
            for (;;) {
#pragma unroll
                for (int t = 0; t < 8; ++t) base[t] = MFMAF(ar[t], bsel, base[t]);
                if (!todo) break;
                next_round();
            }

            PROF(2)
            // positional-encoding k-steps: 4 (octave, coordinate) combos each; the operand of the
            // next k-step is produced behind the matrix instructions of the current one
            f32x4 ph, pl, nh, nl;
            {
                float sv[4], cv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) rev_sincos_h(rv[i % 3], (float)(1 << (i / 3)), sv[i], cv[i]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float hi, lo;
                    split2(sv[i], cv[i], hi, lo);
                    ph[i] = hi;
                    pl[i] = lo;
                }
            }
            float scg = 1.f;  // 2^(4g)
            for (int g = 0; g < G; ++g) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    // next k-step: (g, j+1) or (g+1, 0)
                    const int jn = (j + 1) % 3;
                    const float scn = j == 2 ? scg * 16.f : scg;
                    float sv[4], cv[4];
#pragma unroll
                    for (int q = 0; q < CH_QUADS; ++q) {
                        const int t = q >> 1;
                        const f32x4 A = feed_take(f, c, sb, q);
                        if (!(q & 1)) {
                            base[t] = MFMAH(A, ph, base[t]);
                            base[t] = MFMAH(A, pl, base[t]);
                            if (t < 4) {
                                const int idx = 4 * jn + t;
                                rev_sincos_h(rv[idx % 3], scn * (float)(1 << (idx / 3)), sv[t], cv[t]);
                            } else {
                                float hi, lo;
                                split2(sv[t - 4], cv[t - 4], hi, lo);
                                nh[t - 4] = hi;
                                nl[t - 4] = lo;
                            }
                        } else {
                            base[t] = MFMAH(A, ph, base[t]);
                        }
                        SCHED_FENCE();
                    }
                    ph = nh;
                    pl = nl;
                }
                scg *= 16.f;
            }
            {
                // tail k-step: raw x, y, z
                const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
                ph = zero4;
                pl = zero4;
                float hi, lo;
                split2(px, py, hi, lo);
                ph[0] = hi;
                pl[0] = lo;
                split2(pz, 0.f, hi, lo);
                ph[1] = hi;
                pl[1] = lo;
#pragma unroll
                for (int q = 0; q < CH_QUADS; ++q) {
                    const int t = q >> 1;
                    const f32x4 A = feed_take(f, c, sb, q);
                    if (!(q & 1)) {
                        base[t] = MFMAH(A, ph, base[t]);
                        base[t] = MFMAH(A, pl, base[t]);
                    } else {
                        base[t] = MFMAH(A, ph, base[t]);
                    }
                    SCHED_FENCE();
                }
            }

            PROF(3)
            // ---------------- passes (1 for IMNet, n_iter for IEF) ----------------
            float val = a.init[net];
            const float* ax = a.aux + net * LIDF_AUX_FLOATS;
            const int npass = a.npass[net];
            for (int pass = 0; pass < npass; ++pass) val += decoder_pass_h(f, c, sb, base, val, h, ax);

            PROF(4)
            // ---------------- outputs ----------------
            if (valid && h == 0) {
                const float o = out_act_h(val, a.sigmoid[net]);
                if (a.out[net]) a.out[net][p] = o;
                if (a.is_offset[net]) {
                    // pipeline.py:437-439, same operation order in f32
                    const float ex = __fmul_rn(cur.dx, cur.te);
                    const float ey = __fmul_rn(cur.dy, cur.te);
                    const float ez = __fmul_rn(cur.dz, cur.te);
                    float s = __fadd_rn(__fmul_rn(o, a.rscale), a.r0);
                    s = __fmul_rn(__fmul_rn(s, a.sqrt3), a.part_size);
                    a.pair_pred_pos[3 * p + 0] = __fadd_rn(ex, __fmul_rn(s, cur.dx));
                    a.pair_pred_pos[3 * p + 1] = __fadd_rn(ey, __fmul_rn(s, cur.dy));
                    a.pair_pred_pos[3 * p + 2] = __fadd_rn(ez, __fmul_rn(s, cur.dz));
                }
            }
        }
        cur = nxt;
        nxt.ray = nx2.ray;
        nxt.vid = nx2.vid;
        nxt.te = nx2.te;
        nxt.tl = nx2.tl;
    }
    PROF_DUMP
}

extern "C" hipError_t lidf_launch_points_h(const PointsArgs& a, int grid, hipStream_t st) {
    if (a.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_points_h_kernel, dim3(grid), dim3(256), 0, st, a);
    return hipGetLastError();
}
