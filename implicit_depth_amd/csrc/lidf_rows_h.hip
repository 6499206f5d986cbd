// lidf_rows_h.hip — IMNet / IEF on materialised [n, D] rows with split-f16 matrix instructions
// (gfx950 only): the decoder boundary of the reference (models/pipeline.py:434-435) at the
// arithmetic of lidf_points_h.hip (w*x ~= wh*xh + wh*xl + wl*xh, f32 accumulation).
//
// Layer 1 is k-outer over the D input columns: per k-step of 16 columns every lane splits its 8
// row values into f16 pieces (the rows stream from HBM in double-buffered bursts of 8 k-steps),
// and all 8 output tiles accumulate (the 128-register pre-activation `base` is held for the IEF
// iterations, as in lidf_points.hip). Layers 2-4 are the k-outer chained pass: each H1 / H2 tile
// is consumed by all output tiles as soon as it has been produced. One wavefront per SIMD; the
// four wavefronts of a workgroup consume the same weight stream in lockstep through LDS (chunks of
// 16 quads, 3 buffers, one barrier per chunk, 4-deep ds_read ring).
//
// Stream per decoder: [layer 1: NKS k-steps x (8 tiles x (hi, lo))] [pass: 176 quads, pass_desc()].
#include "lidf_device.h"

namespace rowsh {
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

#define MFMAH(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, (a)), __builtin_bit_cast(h8, (b)), (c), 0, 0, 0)
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define LDQ(rs, voff, soff) \
    __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128((rs), (voff), (soff), 0))

#define RPASS_QUADS 176
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

__device__ _Float16 stream_value_r(const StreamLayout& lay, const NetW* nets, const L1Map& m,
                                   long long e) {
    const int per_net = lay.net_quads * 512;
    const int net = (int)(e / per_net);
    e %= per_net;
    const NetW& n = nets[net];
    int quad = (int)(e / 512);
    const int lane = (int)(e % 512) / 8, i = (int)(e & 7);
    const int half = lane >> 5, o = lane & 31;
    if (quad < lay.l1_quads) {
        // layer 1, k-step ks: operand column 16ks + 8*half + i; column D carries b1 (+ the IEF constant)
        const int ks = quad / 16, r = quad % 16, t = r >> 1, part = r & 1;
        const int x = 16 * ks + 8 * half + i, out = 32 * t + o;
        float w = 0.f;
        if (x < m.D) {
            const int colw = x < m.n0 ? m.c0 + x : m.c1 + (x - m.n0);
            w = n.w1[(size_t)out * n.ld1 + colw];
        } else if (x == m.D && m.add_bias) {
            w = n.b1[out];
            if (n.is_ief)
                for (int j = 0; j < 16; ++j) w += n.w1[(size_t)out * n.ld1 + n.dcore + j] * n.benc[j];
        }
        return hpiece(w, part);
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

__global__ void lidf_pack_r_kernel(StreamLayout lay, NetW net0, NetW net1, L1Map m,
                                   _Float16* stream, float* aux) {
    if (lay.guard && lay.guard->dirty == 0) return;   // guarded packing: fingerprint unchanged
    NetW nets[2] = {net0, net1};
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (long long)lay.total * 2) stream[e] = stream_value_r(lay, nets, m, e);
    if (aux && e < lay.nets * LIDF_AUX_FLOATS) {
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

__device__ __forceinline__ float out_act_r(float y, int use_sigmoid) {
    // implicit_net.py:93-96 / :148-151
    if (use_sigmoid) return 1.f / (1.f + expf(-y));
    return fmaxf(fminf(y, y * 0.01f + 0.99f), y * 0.01f);
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
    const int lim = f.s_pass < 0 ? c.nk1 : RPASS_QUADS / CH_QUADS;
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
    if (Q == 8) {
        // The buffer written here last held the chunk before the previous one: every wavefront
        // finished reading it before it passed the previous barrier.
#pragma unroll
        for (int j = 0; j < 4; ++j) sb[f.nb * CH_ELEMS + (4 * c.wave + j) * 64 + c.lane] = f.stage[j];
        feed_issue(f, c);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
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
    for (int s = 0; s < RPASS_QUADS; ++s) {
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

#define XBURST 4   // k-steps of row operands per burst

__global__ void __launch_bounds__(256) lidf_rows_h_kernel(PointsArgs a) {
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
    const int nks = c.nk1;

    const long long AN = a.n_dev ? (long long)*a.n_dev : a.n;   // device-side count (the frame path)
    const long long ntile = (AN + 127) / 128;
    const long long per = ntile / gridDim.x, rem = ntile % gridDim.x;
    const long long tb = blockIdx.x * per + (blockIdx.x < rem ? blockIdx.x : rem);
    const long long te_ = tb + per + (blockIdx.x < rem ? 1 : 0);
    if (tb >= te_) return;

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

    for (long long tile = tb; tile < te_; ++tile) {
        const long long p = tile * 128 + wave * 32 + col;
        const bool valid = p < AN;
        const long long pc = valid ? p : AN - 1;
        // this lane's operand columns in k-step ks: 16ks + 8h + {0..7}; column D = 1 (bias)
        const float* xrow = a.X + (size_t)pc * a.ldx + 8 * h;
        auto load_x = [&](int ks, f32x4 (&x)[2]) {
            const int x0 = 16 * ks + 8 * h;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (x0 + 4 * q + 3 < a.D) {
                    const f32x4u v = *(const f32x4u*)(xrow + 16 * ks + 4 * q);
                    x[q] = f32x4{v[0], v[1], v[2], v[3]};
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int xc = x0 + 4 * q + i;
                        x[q][i] = xc < a.D ? xrow[16 * ks + 4 * q + i] : (xc == a.D ? 1.f : 0.f);
                    }
                }
            }
        };

        for (int net = 0; net < a.nets; ++net) {
            f32x16 base[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
#pragma unroll
                for (int i = 0; i < 16; ++i) base[t][i] = 0.f;
            }
            // ---------------- layer 1 ----------------
            // The rows stream from HBM and VMEM loads return in order: the operands of XBURST
            // k-steps are fetched in one burst, the next burst while this one multiplies.
            f32x4 xa[XBURST][2], xn[XBURST][2];
#pragma unroll
            for (int i = 0; i < XBURST; ++i) {
                if (i < nks) load_x(i, xa[i]);
                else xa[i][0] = xa[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            for (int k0 = 0; k0 < nks; k0 += XBURST) {
#pragma unroll
                for (int i = 0; i < XBURST; ++i) {
                    if (k0 + XBURST + i < nks) load_x(k0 + XBURST + i, xn[i]);
                    else xn[i][0] = xn[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                SCHED_FENCE();
#pragma unroll
                for (int i = 0; i < XBURST; ++i) {
                    if (k0 + i >= nks) break;
                    f32x4 ph, pl;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        float hi, lo;
                        split2(xa[i][w >> 1][2 * (w & 1)], xa[i][w >> 1][2 * (w & 1) + 1], hi, lo);
                        ph[w] = hi;
                        pl[w] = lo;
                    }
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
#pragma unroll
                for (int i = 0; i < XBURST; ++i) {
                    xa[i][0] = xn[i][0];
                    xa[i][1] = xn[i][1];
                }
            }

            if (a.out_base) {
                // layer-1-only use (the per-ray partial products of the query): [n, nets*256]
                if (valid) {
                    float* ob = a.out_base + ((size_t)p * a.nets + net) * 256 + 4 * h;
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            f32x4 v;
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = base[t][4 * g + i];
                            *(f32x4*)(ob + t * 32 + 8 * g) = v;
                        }
                    }
                }
                continue;
            }
            // ---------------- passes (1 for IMNet, n_iter for IEF) ----------------
            float val = a.init[net];
            const float* ax = a.aux + net * LIDF_AUX_FLOATS;
            const int npass = a.npass[net];
            for (int pass = 0; pass < npass; ++pass) val += decoder_pass_h(f, c, sb, base, val, h, ax);

            if (valid && h == 0 && a.out[net]) a.out[net][p] = out_act_r(val, a.sigmoid[net]);
        }
    }
}

}  // namespace rowsh

// l1only: the stream holds the layer-1 sections only (npass must be 0 for every net)
extern "C" StreamLayout lidf_make_layout_rows_h(int nets, int D, int l1only) {
    StreamLayout s;
    s.nets = nets;
    s.mode = LIDF_MODE_FUSED_H;
    s.l1_quads = 16 * ((D + 1 + 15) / 16);
    s.net_quads = s.l1_quads + (l1only ? 0 : RPASS_QUADS);
    s.total = nets * s.net_quads * 256;
    s.guard = nullptr;
    return s;
}

extern "C" hipError_t lidf_launch_pack_rows_h(const StreamLayout& lay, const NetW& n0, const NetW& n1,
                                              const L1Map& m, float* stream, float* aux,
                                              hipStream_t st) {
    const long long total = (long long)lay.total * 2;
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(rowsh::lidf_pack_r_kernel, dim3(blocks), dim3(256), 0, st, lay, n0, n1, m,
                       (_Float16*)stream, aux);
    return hipGetLastError();
}

extern "C" hipError_t lidf_launch_rows_h(const PointsArgs& a, int grid, hipStream_t st) {
    if (a.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(rowsh::lidf_rows_h_kernel, dim3(grid), dim3(256), 0, st, a);
    return hipGetLastError();
}
