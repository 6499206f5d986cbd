// lidf_points.hip — weight-stream packer and the per-point decoder kernel (gfx950 / CDNA4).
//
// What the kernel computes (reference: models/implicit_net.py:81-98 IMNet.forward, :129-152
// IEF.forward, called at models/pipeline.py:434-435; fused mode additionally restates
// models/pipeline.py:343-365 (positional encoding of enter / leave positions), :410 (voxel feature
// gather), :367-397 (ROI feature, via the per-ray partial) and :437-439 (offset scaling, position)).
//
// Layer-1 algebra used by the fused mode (exact up to f32 re-association):
//   W1 x = W1[:, vox] vox_feat[v]  +  W1[:, rgb|dir] rayfeat[r]  +  W1[:, enter|leave] PE(p)  + b1
//          `----- voxpart[v] ------'  `------ raypart[r] -------'  `--- in this kernel -----'
//   IEF:  W1[:, D:D+16] (wenc*off + benc) = u*off + c   (u, c are [256] vectors; c joins the bias)
// so per point only the 2*(3+6L) position-embedding columns go through MFMA in layer 1.
//
// Execution structure (one wavefront = 32 points, 1 wavefront per SIMD, see lidf_device.h for the
// stream format): the A fragments of every layer / pass / net / tile are consumed strictly in
// stream order through ONE 8-deep register ring that is refilled 8 quads ahead and never drains,
// so no layer ever starts behind an L2 round trip.
//
// Measured fact that shapes everything else (scripts/mfma_valu_ubench.hip): on gfx950 the f32 MFMA
// executes on the vector ALU itself — a VALU instruction issued by the same wavefront does NOT
// overlap it, every one adds its ~4 issue cycles to the 64 cycles of the MFMA. So the kernel is
// bound by (MFMA count x 64 + VALU instruction count x 4) cycles per wavefront, and the VALU side
// is kept minimal: leaky-relu in 2 instructions per register, sin/cos on the transcendental unit
// behind an exact range reduction in revolutions, the per-ray layer-1 partial added by a rank-1
// MFMA instead of 128 v_add per net, no staging through LDS.
#include "lidf_device.h"

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// Development-only phase timers (-DLIDF_PROFILE): wave 0 of block 0 accumulates s_memtime deltas
// per phase into a.out_base[0..7] (as integers); every wavefront leaves its start / end wall clock
// and its shader-clock total at a.out_base[16 + 4 (4 block + wave) ..]. The shipped library is built
// without it.
#ifdef LIDF_PROFILE
#define PROF_DECL long long prof_t = clock64(); long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const long long prof_c0 = prof_t, prof_w0 = wall_clock64();
#define PROF(i) { long long t_ = clock64(); prof_acc[i] += t_ - prof_t; prof_t = t_; }
#define PROF_DUMP if (blockIdx.x == 0 && threadIdx.x == 0 && a.out_base) { for (int i_ = 0; i_ < 8; ++i_) ((long long*)a.out_base)[i_] = prof_acc[i_]; } \
    if ((threadIdx.x & 63) == 0 && a.out_base) { long long* o_ = (long long*)a.out_base + 16 + 4 * (blockIdx.x * 4 + (threadIdx.x >> 6)); o_[0] = prof_w0; o_[1] = wall_clock64(); o_[2] = clock64() - prof_c0; }
#else
#define PROF_DECL
#define PROF(i)
#define PROF_DUMP
#endif

// Weight-stream loads go through a buffer descriptor: the (wave-uniform) stream position lives in
// an SGPR and the lane offset is one constant VGPR, so the unrolled layers need no per-load
// 64-bit address registers (with flat pointers hipcc materialises and spills hundreds of them).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define LDQ(rs, voff, soff) \
    __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128((rs), (voff), (soff), 0))

// ------------------------------------------------------------------------------------------------
// Packer: one thread per stream float.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int k_to_feature(int s, int half) {
    // k-step s of a layer whose input is the previous layer's accumulator registers:
    // tile T = s/16, register r = s%16 -> feature 32T + (r&3) + 8(r>>2) + 4*half
    int T = s >> 4, r = s & 15;
    return 32 * T + (r & 3) + 8 * (r >> 2) + 4 * half;
}

__device__ __forceinline__ float ief_u(const NetW& n, int out) {
    float acc = 0.f;
    for (int j = 0; j < 16; ++j) acc += n.w1[(size_t)out * n.ld1 + n.dcore + j] * n.wenc[j];
    return acc;
}
__device__ __forceinline__ float ief_c(const NetW& n, int out) {
    float acc = 0.f;
    for (int j = 0; j < 16; ++j) acc += n.w1[(size_t)out * n.ld1 + n.dcore + j] * n.benc[j];
    return acc;
}

// (the two nets by reference, selected per element: a NetW[2] copy indexed by a runtime value lived in scratch —
// 208 bytes per lane in every pack launch)
__device__ float stream_value(const StreamLayout& lay, const NetW& net0, const NetW& net1, const L1Map& m, int e) {
    const int net = e / (lay.net_quads * 256);
    e %= lay.net_quads * 256;
    const NetW& n = net ? net1 : net0;
    int quad = e / 256;
    const int lane = (e % 256) / 4, jj = e & 3;
    const int half = lane >> 5, c32 = lane & 31;
    if (quad < lay.l1_quads) {  // ---- layer 1 (8 output tiles)
        if (lay.mode == LIDF_MODE_FUSED) {
            const int Lp = (m.L + 1) / 2;
            int t, feat;  // feat: index inside the (3+6L)-wide embedding, -1 = zero pad
            if (quad < 24 * Lp) {
                const int it = quad / 24, jq = quad % 3;
                t = (quad % 24) / 3;
                const int i = 4 * jq + jj;  // 0..11: sin xyz, cos xyz of octave 2it, then 2it+1
                const int o = 2 * it + i / 6;
                feat = o < m.L ? 3 + 6 * o + i % 6 : -1;
            } else {
                t = quad - 24 * Lp;
                feat = jj < 3 ? jj : -1;
            }
            if (feat < 0) return 0.f;
            const int col = (half ? m.leave_c0 : m.enter_c0) + feat;
            return n.w1[(size_t)(32 * t + c32) * n.ld1 + col];
        }
        const int kq = quad / m.nt, t = quad % m.nt;
        const int s = 4 * kq + jj;
        const int out = (m.xcol && t == m.nt - 1) ? 32 * t : 32 * t + c32;
        if (out >= m.nout) return 0.f;
        const int x = 8 * kq + 4 * half + jj;  // operand column
        (void)s;
        if (x < m.D) {
            const int col = x < m.n0 ? m.c0 + x : m.c1 + (x - m.n0);
            if (m.transposed == 2 && x >= m.n0) return net1.w1[(size_t)col * net1.ld1 + out];
            return m.transposed ? n.w1[(size_t)col * n.ld1 + out] : n.w1[(size_t)out * n.ld1 + col];
        }
        if (x == m.D && m.add_bias) {
            float b = n.b1[out];
            if (n.is_ief) b += ief_c(n, out);
            return b;
        }
        if (x == m.D + 1 && m.add_u && n.is_ief) return ief_u(n, out);
        return 0.f;
    }
    quad -= lay.l1_quads;
    if (quad < LIDF_U_QUADS) {  // ---- u fragments (zero for IMNet)
        if (half != 0 || !n.is_ief) return 0.f;
        return ief_u(n, 32 * (4 * quad + jj) + c32);
    }
    quad -= LIDF_U_QUADS;
    if (quad < 4 * LIDF_L2_QUADS) {  // ---- layer 2, k-quad major: quad = 4 kq + output tile
        const int t = quad % 4, s = 4 * (quad / 4) + jj;
        const int out = 32 * t + c32;
        if (s < LIDF_H1 / 2) return n.w2[(size_t)out * LIDF_H1 + k_to_feature(s, half)];
        if (s == LIDF_H1 / 2 && half == 0) return n.b2[out];
        return 0.f;
    }
    quad -= 4 * LIDF_L2_QUADS;
    {  // ---- layer 3, k-quad major: quad = 2 kq + output tile
        const int t = quad % 2, s = 4 * (quad / 2) + jj;
        const int out = 32 * t + c32;
        if (s < LIDF_H2 / 2) return n.w3[(size_t)out * LIDF_H2 + k_to_feature(s, half)];
        if (s == LIDF_H2 / 2 && half == 0) return n.b3[out];
        return 0.f;
    }
}

// element e of the stage-2 decoder's 16 x 16 x 4 stream (layout: lidf_ief16.hip); m.n0 = E embed(pos)
// columns starting at w1 column m.c0
__device__ float ief16_stream_value(const NetW& n, const L1Map& m, int e) {
    const int KQ = (m.n0 + 15) / 16;
    int quad = e / 256;
    const int lane = (e % 256) / 4, r = e & 3;
    const int j = lane & 15, g = lane >> 4;
    if (quad < KQ * 16) {
        const int kq = quad / 16, To = quad % 16;
        const int x = 16 * kq + 4 * g + r;
        return x < m.n0 ? n.w1[(size_t)(16 * To + j) * n.ld1 + m.c0 + x] : 0.f;
    }
    quad -= KQ * 16;
    if (quad < 2)   // bias quads of layer 2: component r = b2 of output tile 4 quad + r, in group 0
        return g == 0 ? n.b2[16 * (4 * quad + r) + j] : 0.f;
    quad -= 2;
    if (quad < 132) {   // layer 2, k-major: a u quad in front of every fourth input tile
        const int blk = quad / 33, in = quad % 33;
        if (in == 0) return (g == 0 && n.is_ief) ? ief_u(n, 16 * (4 * blk + r) + j) : 0.f;
        const int T = 4 * blk + (in - 1) / 8, To = (in - 1) % 8;
        return n.w2[(size_t)(16 * To + j) * LIDF_H1 + 16 * T + 4 * g + r];
    }
    quad -= 132;
    if (quad < 1) return g == 0 ? n.b3[16 * r + j] : 0.f;   // bias quad of layer 3
    quad -= 1;
    if (quad < 32) {
        const int T = quad / 4, To = quad % 4;
        return n.w3[(size_t)(16 * To + j) * LIDF_H2 + 16 * T + 4 * g + r];
    }
    return 0.f;
}

// element e of the any-width decoder chain's stream (layout: lidf_chain16.hip); G = m.nt = gf / 16, m.n0 = E per-row
// columns starting at w1 column m.c0
__device__ float chain16_stream_value(const NetW& n, const L1Map& m, int e) {
    const int G = m.nt, T1 = 4 * G, T2 = 2 * G, T3 = G;
    const int KQ = (m.n0 + 15) / 16;
    int quad = e / 256;
    const int lane = (e % 256) / 4, r = e & 3;
    const int j = lane & 15, g = lane >> 4;
    if (quad < KQ * T1) {
        const int kq = quad / T1, To = quad % T1;
        const int x = 16 * kq + 4 * g + r;
        return x < m.n0 ? n.w1[(size_t)(16 * To + j) * n.ld1 + m.c0 + x] : 0.f;
    }
    quad -= KQ * T1;
    const int NB2 = (T2 + 3) / 4, NB3 = (T3 + 3) / 4;
    if (quad < NB2) {   // bias quads of layer 2: component r = b2 of output tile 4 quad + r, in group 0
        const int tile = 4 * quad + r;
        return (g == 0 && tile < T2) ? n.b2[16 * tile + j] : 0.f;
    }
    quad -= NB2;
    const int blk_len = 1 + 4 * T2;   // a u quad in front of every fourth input tile, T2 quads per input tile
    if (quad < (T1 / 4) * blk_len) {
        const int blk = quad / blk_len, in = quad % blk_len;
        if (in == 0) return (g == 0 && n.is_ief) ? ief_u(n, 16 * (4 * blk + r) + j) : 0.f;
        const int T = 4 * blk + (in - 1) / T2, To = (in - 1) % T2;
        return n.w2[(size_t)(16 * To + j) * (64 * G) + 16 * T + 4 * g + r];
    }
    quad -= (T1 / 4) * blk_len;
    if (quad < NB3) {
        const int tile = 4 * quad + r;
        return (g == 0 && tile < T3) ? n.b3[16 * tile + j] : 0.f;
    }
    quad -= NB3;
    if (quad < T2 * T3) {
        const int T = quad / T3, To = quad % T3;
        return n.w3[(size_t)(16 * To + j) * (32 * G) + 16 * T + 4 * g + r];
    }
    return 0.f;
}

__device__ __forceinline__ void pack_body(const StreamLayout& lay, const NetW& net0, const NetW& net1,
                                          const L1Map& m, float* stream, float* aux) {
    if (lay.guard && lay.guard->dirty == 0) return;   // guarded packing: fingerprint unchanged
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (lay.mode == LIDF_MODE_PNET_CHAIN) {   // a PointNet2Stage's per-point chain stream riding as a job
        // (net0 carries w_p1, b_p1, w_p2, b_p2, w_p3, w_p4, b_p4 in w1, b1, w2, b2, w3, b3, w4)
        const PnetW w = {net0.w1, net0.b1, net0.w2, net0.b2, net0.w3, net0.b3, net0.w4};
        if (e < lay.total) stream[e] = pn_stream_value(w, e);
        return;
    }
    if (lay.mode == LIDF_MODE_PNET_BWD) {     // its backward chains' stream (same fields of net0)
        const PnetW w = {net0.w1, net0.b1, net0.w2, net0.b2, net0.w3, net0.b3, net0.w4};
        if (e < lay.total) stream[e] = pn_bwd_stream_value(w, e);
        return;
    }
    if (lay.mode == LIDF_MODE_IEF16) {
        if (e < lay.total) stream[e] = ief16_stream_value(net0, m, e);
        if (e < IEF16_AUX_FLOATS)
            aux[e] = e < 64 ? net0.w4[e] : (e == 64 ? net0.b4[0] : 0.f);
        return;
    }
    if (lay.mode == LIDF_MODE_CHAIN16) {
        const int gf = 16 * m.nt;
        if (e < lay.total) stream[e] = chain16_stream_value(net0, m, e);
        if (e < gf + 8) aux[e] = e < gf ? net0.w4[e] : (e == gf ? net0.b4[0] : 0.f);
        return;
    }
    if (e < lay.total) stream[e] = stream_value(lay, net0, net1, m, e);
    if (lay.mode != LIDF_MODE_L1ONLY && lay.mode != LIDF_MODE_LINEAR && e < lay.nets * LIDF_AUX_FLOATS) {
        int sec = e / LIDF_AUX_FLOATS, i = e % LIDF_AUX_FLOATS;
        const NetW& n = sec ? net1 : net0;
        float v = 0.f;
        if (i < 64) {
            int half = i / 32, s = i % 32;
            v = n.w4[k_to_feature(s, half)];
        } else if (i == 64) {
            v = n.b4[0];
        }
        aux[e] = v;
    }
}

__global__ void lidf_pack_kernel(StreamLayout lay, NetW net0, NetW net1, L1Map m, float* stream,
                                 float* aux) {
    pack_body(lay, net0, net1, m, stream, aux);
}

// Several streams of one module packed by ONE launch (grid.y = job): the three streams of the fused
// query, the per-voxel layers of a PointNet, the two streams of the stage-2 IEF — with guarded packing
// these launches return at once on almost every call, so their number is what they cost.
__global__ void lidf_pack_multi_kernel(PackJobs j) {
    const PackJob& k = j.job[blockIdx.y];
    pack_body(k.lay, k.n0, k.n1, k.m, k.stream, k.aux);
}

extern "C" hipError_t lidf_launch_pack_multi(const PackJobs& j, hipStream_t st) {
    if (j.n <= 0) return hipSuccess;
    int blocks = 1;
    for (int i = 0; i < j.n; ++i) {
        const StreamLayout& lay = j.job[i].lay;
        const int total = lay.total > lay.nets * LIDF_AUX_FLOATS ? lay.total : lay.nets * LIDF_AUX_FLOATS;
        const int b = (total + 255) / 256;
        blocks = b > blocks ? b : blocks;
    }
    hipLaunchKernelGGL(lidf_pack_multi_kernel, dim3(blocks, j.n), dim3(256), 0, st, j);
    return hipGetLastError();
}

extern "C" hipError_t lidf_launch_pack(const StreamLayout& lay, const NetW& n0, const NetW& n1,
                                       const L1Map& m, float* stream, float* aux,
                                       hipStream_t st) {
    int total = lay.total > lay.nets * LIDF_AUX_FLOATS ? lay.total : lay.nets * LIDF_AUX_FLOATS;
    int blocks = (total + 255) / 256;
    hipLaunchKernelGGL(lidf_pack_kernel, dim3(blocks), dim3(256), 0, st, lay, n0, n1, m, stream,
                       aux);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Per-point kernel helpers
// ------------------------------------------------------------------------------------------------
// leaky_relu(0.02) on registers [LO, HI) of one accumulator tile
// (fmaxf would add a canonicalising v_max per input under the IEEE mode; one v_mul + one v_max
// is the whole activation, implicit_net.py:83)
template <int LO, int HI>
__device__ __forceinline__ void lrelu_part(f32x16& v) {
    // two values per v_pk_mul_f32, one v_max_f32 each
#pragma unroll
    for (int i = LO; i < HI; i += 2) {
        f32x2 x = {v[i], v[i + 1]};
        f32x2 t;
        const f32x2 slope = {0.02f, 0.02f};
        asm("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(x), "v"(slope));
        float r0, r1;
        asm("v_max_f32 %0, %1, %2" : "=v"(r0) : "v"(x[0]), "v"(t[0]));
        asm("v_max_f32 %0, %1, %2" : "=v"(r1) : "v"(x[1]), "v"(t[1]));
        v[i] = r0;
        v[i + 1] = r1;
    }
}

__device__ __forceinline__ float out_act(float y, int use_sigmoid) {
    // implicit_net.py:93-96 / :148-151
    if (use_sigmoid) return 1.f / (1.f + expf(-y));
    return fmaxf(fminf(y, y * 0.01f + 0.99f), y * 0.01f);
}

// One decoder pass on the 32 points of this wavefront:
//   H1 = lrelu(base + u*val);  H2 = lrelu(W2 H1 + b2);  H3 = lrelu(W3 H2 + b3);  y = w4.H3 + b4
// `ring` holds the next 8 quads of the stream on entry (u0, u1, first six layer-2 quads) and the
// next 8 quads after the pass on exit: the pass's own first quads again when wrap_base ==
// pass_base (another pass of the same net follows) or the first quads of the next block.
//
// Prefetch of the next layer-1 accumulator init: while layer 2 runs, the pass copies the voxpart
// row that the NEXT net / tile starts from (32 x 16 bytes per lane, address vox_rows + vp_off)
// into this wavefront's 32 KiB of LDS with global -> LDS DMA (global_load_lds_dwordx4: no register
// is held, nothing waits); the caller moves it from LDS into `base` once the passes of the net
// are over. Slot j (tile j/4, register group j%4) of lane l sits at lds_wave + j*1024 + l*16.
#define LIDF_PF_S0 36  // first DMA step (every H1 tile exists after step 26); one every 2nd step
#define LIDF_PF_S1 (LIDF_U_QUADS + 4 * LIDF_L2_QUADS + 1)  // first LDS -> base step (layer 3)
typedef __attribute__((address_space(3))) float lds_float;
// Issued as inline assembly on purpose: when the compiler sees LDS-DMA and ordinary buffer loads
// share vmcnt it assumes they may complete out of order and drains the queue (s_waitcnt vmcnt(0))
// every few steps — the weight ring would stall eight times per pass. Hidden from its counter the
// DMA only makes the ring waits slightly stricter (the compiler under-counts what is in flight), so
// it is issued every second step. M0 (LDS base of the transfer) is not used by anything else in
// this kernel. The instruction offset is added to the global AND the LDS address: global row
// offset 32 j bytes, LDS slot offset 1024 j bytes -> M0 = wave base + 992 j.
#define LIDF_DMA_CASE(J)                                                                     \
    case J:                                                                                  \
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3"  \
                     :                                                                       \
                     : "s"(lds_addr + (J) * 992), "v"(vp_off), "s"(vox_rows), "n"(32 * (J))   \
                     : "memory");                                                            \
        break;
__device__ __forceinline__ void dma_slot(const char* vox_rows, const unsigned vp_off,
                                         const unsigned lds_addr, const int j) {
    switch (j) {
        LIDF_DMA_CASE(0) LIDF_DMA_CASE(1) LIDF_DMA_CASE(2) LIDF_DMA_CASE(3) LIDF_DMA_CASE(4)
        LIDF_DMA_CASE(5) LIDF_DMA_CASE(6) LIDF_DMA_CASE(7) LIDF_DMA_CASE(8) LIDF_DMA_CASE(9)
        LIDF_DMA_CASE(10) LIDF_DMA_CASE(11) LIDF_DMA_CASE(12) LIDF_DMA_CASE(13) LIDF_DMA_CASE(14)
        LIDF_DMA_CASE(15) LIDF_DMA_CASE(16) LIDF_DMA_CASE(17) LIDF_DMA_CASE(18) LIDF_DMA_CASE(19)
        LIDF_DMA_CASE(20) LIDF_DMA_CASE(21) LIDF_DMA_CASE(22) LIDF_DMA_CASE(23) LIDF_DMA_CASE(24)
        LIDF_DMA_CASE(25) LIDF_DMA_CASE(26) LIDF_DMA_CASE(27) LIDF_DMA_CASE(28) LIDF_DMA_CASE(29)
        LIDF_DMA_CASE(30) LIDF_DMA_CASE(31)
        default: break;
    }
}
// ST: the activations of the pass are kept (training forward): tile by tile, as each is complete,
// to row pointers that already carry the lane's 4h column offset.
struct TrainRow {
    float* h1;
    float* h2;
    float* h3;
    unsigned* m1;   // sign words of the lane's H1 values (4 words: tiles 2w | 2w+1), see mask_tile
    unsigned* m2;   // and of its H2 values (2 words)
};
// One bit per value, (v > 0): what the backward chain needs of H1 / H2 (lrelu'), 1/32 of the bytes.
// 0 - v has its sign bit set exactly when v > 0 (+-0 give +0); v_alignbit shifts it in from the
// right, so value e of tile T ends at bit 31 - (16 (T & 1) + e) of word T / 2.
__device__ __forceinline__ void mask_tile(unsigned& word, const f32x16& v) {
#pragma unroll
    for (int e = 0; e < 16; ++e)
        word = __builtin_amdgcn_alignbit(word, __builtin_bit_cast(unsigned, 0.f - v[e]), 31);
}
template <int W>
__device__ __forceinline__ void store_tile(float* row, const int t, const f32x16& v) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = v[4 * g + i];
        *(f32x4*)(row + 32 * t + 8 * g) = o;
    }
}
template <bool PF, bool ST = false>
__device__ __forceinline__ float decoder_pass(const __amdgpu_buffer_rsrc_t srs, const int vq,
                                              f32x4 (&ring)[LIDF_RING], const int pass_base,
                                              const int wrap_base, f32x16 (&base)[8],
                                              const float val, const int h, const float one_b,
                                              const float* __restrict__ ax,
                                              const char* __restrict__ vox_rows,
                                              const unsigned vp_off, const unsigned lds_addr,
                                              const lds_float* lds_wave, const int lane,
                                              const TrainRow tr
#ifdef LIDF_PROFILE
                                              , long long& prof_t, long long (&prof_acc)[8]
#endif
                                              ) {
    constexpr int S_L2 = LIDF_U_QUADS;
    constexpr int S_L3 = S_L2 + 4 * LIDF_L2_QUADS;
    // Stream offsets: one SGPR per group of four quads plus the instruction's immediate offset.
    // The bases are made opaque so that the 168 offsets are formed next to their loads (s_add)
    // instead of being hoisted out of the pass loop, where they overflow the SGPR file and come
    // back through v_readlane — a VALU instruction per region.
    int pb = pass_base, wb = wrap_base;
    asm volatile("" : "+s"(pb), "+s"(wb));
    const float ob = h ? 0.f : val;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                           0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 H1[8], H2[4], H3[2];
    f32x4 u0, u1, w4[8];
    unsigned mk1 = 0u, mk2 = 0u;   // ST only: the sign word being filled (stored once its 2 tiles are in)
    f32x4 pend = {0.f, 0.f, 0.f, 0.f};
    float b4 = 0.f;
#pragma unroll
    for (int s = 0; s < LIDF_PASS_QUADS; ++s) {
        const f32x4 a = ring[s % LIDF_RING];
        {
            const int nx = s + LIDF_RING;
            const int rel = nx < LIDF_PASS_QUADS ? nx : nx - LIDF_PASS_QUADS;
            // (the whole stream offset in the scalar operand, ONE vector offset: written as vq + k * 1024 the three
            // lane offsets vq + 1024 / 2048 / 3072 became loop-invariant registers of their own — parked in AGPRs, a
            // v_accvgpr_read in front of every second ring load, and in the activation-keeping variant one of them
            // was spilled: a scratch reload + s_waitcnt vmcnt(0) in front of 37 ring loads per pass set, each
            // draining the ring that is meant never to drain; LIDF_RING_VOFF=1 restores that form for A/B runs)
#if defined(LIDF_RING_VOFF)
            ring[s % LIDF_RING] = LDQ(srs, vq + (rel & 3) * 1024,
                                      (nx < LIDF_PASS_QUADS ? pb : wb) + (rel >> 2) * 4096);
#else
            ring[s % LIDF_RING] = LDQ(srs, vq, (nx < LIDF_PASS_QUADS ? pb : wb) + rel * 1024);
#endif
        }
        if (PF && s >= LIDF_PF_S0 && s < LIDF_PF_S0 + 64 && ((s - LIDF_PF_S0) & 1) == 0)
            dma_slot(vox_rows, vp_off, lds_addr, (s - LIDF_PF_S0) / 2);
        if (PF && s == LIDF_PF_S1) {
            // the last DMA is more than 30 ring loads old and the queue completes in order: with
            // at most 12 operations outstanding every transfer has landed (this never stalls)
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        }
        if (PF && s >= LIDF_PF_S1 && s <= LIDF_PF_S1 + 32) {
            // LDS -> base, one slot per step during layer 3 (the old contents are dead: this is
            // the last pass that reads them, and it did so in its first 27 steps); the value read
            // in step s is moved into the accumulator registers in step s+1, so nothing waits on
            // the LDS latency
            const int j = s - LIDF_PF_S1;
            if (j > 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) base[(j - 1) / 4][4 * ((j - 1) % 4) + i] = pend[i];
            }
            if (j < 32)
                pend = *(const __attribute__((address_space(3))) f32x4*)(lds_wave + j * 256 + 4 * lane);
        }
        if (s == 0) {
            u0 = a;
        } else if (s == 1) {
            u1 = a;
            H1[0] = MFMA(u0[0], ob, base[0]);
            lrelu_part<0, 16>(H1[0]);
            if (ST) {
                store_tile<0>(tr.h1, 0, H1[0]);
                mask_tile(mk1, H1[0]);
            }
        } else if (s < S_L3) {
            // ---- layer 2, k-quad major: all four output tiles advance together, so a layer-1
            // tile is produced right before its four k-quads and is dead after them
            const int kq = (s - S_L2) / 4, t = (s - S_L2) % 4;
            if (t == 3 && (kq & 3) == 3 && kq / 4 + 1 < 8) {
                const int T = kq / 4 + 1;   // needed from the next step on
                H1[T] = MFMA(T < 4 ? u0[T & 3] : u1[T & 3], ob, base[T]);
                lrelu_part<0, 16>(H1[T]);
                if (ST) {
                    store_tile<0>(tr.h1, T, H1[T]);
                    mask_tile(mk1, H1[T]);
                    if (T & 1) tr.m1[T / 2] = mk1;
                }
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int k = 4 * kq + jj;
                if (k == 0)
                    H2[t] = MFMA(a[jj], H1[0][0], zero16);
                else if (k < LIDF_H1 / 2)
                    H2[t] = MFMA(a[jj], H1[k / 16][k % 16], H2[t]);
                else if (k == LIDF_H1 / 2)
                    H2[t] = MFMA(a[jj], one_b, H2[t]);
            }
            if (kq == LIDF_L2_QUADS - 1) {
                lrelu_part<0, 16>(H2[t]);   // tile complete (bias quad)
                if (ST) {
                    store_tile<0>(tr.h2, t, H2[t]);
                    mask_tile(mk2, H2[t]);
                    if (t & 1) tr.m2[t / 2] = mk2;
                }
            }
        } else {
            // ---- layer 3, k-quad major
            const int kq = (s - S_L3) / 2, t = (s - S_L3) % 2;
            if (s == S_L3 + 8) {
                // operands of the tail, fetched here so that their latency hides behind layer 3
#pragma unroll
                for (int i = 0; i < 8; ++i) w4[i] = *(const f32x4*)(ax + h * 32 + 4 * i);
                b4 = ax[64];
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int k = 4 * kq + jj;
                if (k == 0)
                    H3[t] = MFMA(a[jj], H2[0][0], zero16);
                else if (k < LIDF_H2 / 2)
                    H3[t] = MFMA(a[jj], H2[k / 16][k % 16], H3[t]);
                else if (k == LIDF_H2 / 2)
                    H3[t] = MFMA(a[jj], one_b, H3[t]);
            }
            if (kq == LIDF_L3_QUADS - 1 && t == 0) {
                lrelu_part<0, 16>(H3[0]);
                if (ST) store_tile<0>(tr.h3, 0, H3[0]);
            }
        }
        SCHED_FENCE();
#ifdef LIDF_PROFILE
        if (s == 1) PROF(4)
        if (s == S_L3 - 1) PROF(6)
        if (s == LIDF_PASS_QUADS - 1) PROF(7)
#endif
    }
    lrelu_part<0, 16>(H3[1]);
    if (ST) store_tile<0>(tr.h3, 1, H3[1]);
    // layer 4: 64 -> 1 on the VALU (4 partial sums), halves combined with one cross-half shuffle
    float ys[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int s = 4 * i + k;
            ys[k] = fmaf(w4[i][k], H3[s / 16][s % 16], ys[k]);
        }
    }
    float y = (ys[0] + ys[1]) + (ys[2] + ys[3]);
    y += __shfl_xor(y, 32);
    return y + b4;
}

struct Geo {
    int ray, vid;
    float te, tl, dx, dy, dz;
};

// ------------------------------------------------------------------------------------------------
// Fused per-point kernel (LIDF_MODE_FUSED): get_embedding + both decoders + pair_pred_pos.
// Per 32-point wave-tile and per net:
//   base  = voxpart[vid]            prefetched by the previous pass (see decoder_pass<PF>)
//   base += W1[:, enter|leave] PE   51 k-steps x 8 tiles, embedding produced in registers
//   base += raypart[ray]            rank-1 MFMAs, two distinct rays per instruction; the row loads
//                                   are issued before the PE steps and consumed after them, up to
//                                   RB rounds (2*RB rays) per batch
//   passes (1 for the IMNet, n_iter for the IEF)
// ------------------------------------------------------------------------------------------------
#ifndef LIDF_RB
#define LIDF_RB 4   // rank-1 rounds whose row loads are in flight together
#endif
#ifndef LIDF_CHUNK
#define LIDF_CHUNK 2   // consecutive wave-tiles per dynamic hand-out (4: 12.40 ms, 2: 12.36, 1: 12.37)
#endif

// ST: every pass's H1 | H2 | H3 | offset-in and the pre-activation output are kept (training forward
// of both decoders in one launch, a.tr_passes / a.tr_pre per net).
template <bool ST>
__device__ __forceinline__ void points_fused_body(const PointsArgs& a) {
    // row / point count: a device-side count (a.n_dev: the sync-free frame path) overrides the host value
    const long long AN = a.n_dev ? (long long)*a.n_dev : a.n;
    extern __shared__ float lds_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int h = lane >> 5;
    const int col = lane & 31;
    const float one_b = h ? 0.f : 1.f;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.stream, 0, a.nets * a.net_quads * 1024, 0x00020000);
    const int vq = lane * 16;
    const int net_bytes = a.net_quads * 1024;
    const int l1_bytes = a.l1_quads * 1024;
    const int Lp = (a.L + 1) / 2;
    // this wavefront's staging area: 32 slots x 64 lanes x 16 B (wave-uniform address)
    lds_float* lds_wave = (lds_float*)lds_raw + __builtin_amdgcn_readfirstlane(wave) * 8192;
    const unsigned lds_addr = (unsigned)(size_t)lds_wave;

    // Wave-tiles (32 points) are the unit of work; a wavefront walks a sequence of them.
    //  * a.tile_counter == nullptr: the static split — a contiguous range of 128-point tiles per
    //    workgroup, the 4 waves interleaved inside it (wave-tile 4 tile + wave);
    //  * else: dynamic hand-out in chunks of LIDF_CHUNK consecutive wave-tiles per wavefront. The
    //    first chunk of wavefront g is chunk g; further chunks come from an atomic counter (zeroed
    //    by the launch before), requested one chunk ahead so that the round trip is never waited
    //    for. Compute units do not run at one clock (2.36-2.40 GHz on the headline launch) and tiles
    //    of ragged lists do not cost the same (1 to 16 rank-1 rounds per net): the faster take more.
    const long long nwt = (AN + 31) / 32;
    // (short lists keep the static split: with a handful of tiles per wavefront one tile is the
    // grain either way and the first requests would be waited for)
    const bool dyn = a.tile_counter != nullptr &&
                     nwt >= (long long)(a.dyn_min_tiles > 0 ? a.dyn_min_tiles : 32) * gridDim.x * 4;
    const long long ntile = (AN + 127) / 128;
    const long long per = ntile / gridDim.x, rem = ntile % gridDim.x;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const long long nwaves = (long long)gridDim.x * 4;
    // The static split's partial last round, NET BY NET (a.tail_split, two nets; round 5). ntile tiles over G
    // workgroups are `per` full rounds and `rem` tiles more: handed out whole, rem of the G compute units run one
    // more (tile, both nets) while the others idle — 8,736 wave-tiles of a real frame over 1,024 SIMDs are 8.53
    // rounds in the time of 9. The two nets of a tile are independent given its geometry (1,062 and 1,716 matrix
    // instructions: prob_dec one pass, offset_dec two), so the 4 rem wave-tiles of the last round become 8 rem
    // one-net items over all 4 G wavefronts: wavefront g < 4 rem takes the offset net of tail wave-tile g, the
    // others the probability net of tail wave-tiles g - 4 rem and (when there are more items than wavefronts)
    // g - 4 rem + (4 G - 4 rem): the round then lasts max(1716, 2 x 1062) instead of 2,778 instruction times.
    // Beyond two prob items per free wavefront (rem > 2/3 G) nothing is gained and the tiles stay whole. Same
    // instruction sequence per (tile, net) as ever: results are bit-identical.
    bool split = !dyn && a.tail_split && a.nets == 2 && rem > 0;
    long long wt0 = per * gridDim.x * 4;     // first wave-tile of the partial round
    int n_tail = 0, freew = 0;
    if (split) {
        n_tail = (int)(nwt - wt0);
        freew = (int)nwaves - n_tail;
        if (n_tail > 2 * freew) split = false;
    }
    const long long tb = split ? blockIdx.x * per : blockIdx.x * per + (blockIdx.x < rem ? blockIdx.x : rem);
    const long long te_ = split ? tb + per : tb + per + (blockIdx.x < rem ? 1 : 0);
    const int gw = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + uwave));
    const int chunk = a.dyn_chunk > 0 ? a.dyn_chunk : LIDF_CHUNK;   // (a chunk is also the grain of the tail)
    // item q of this wavefront's static sequence: its wave-tile (>= nwt: the sequence has ended) and nets [lo, hi)
    auto seq = [&](int q, long long& w, int& lo, int& hi) {
        lo = 0;
        hi = a.nets;
        w = nwt;
        if (q < (int)(te_ - tb)) {
            w = (tb + q) * 4 + uwave;
            return;
        }
        if (!split) return;
        const int t = q - (int)per;
        if (gw < n_tail) {
            if (t == 0) { w = wt0 + gw; lo = 1; hi = 2; }
            return;
        }
        const int j = gw - n_tail + t * freew;
        if (t < 2 && j < n_tail) { w = wt0 + j; lo = 0; hi = 1; }
    };
    int pos = 0;
    long long w_cur;
    int lo_cur = 0, hi_cur = a.nets, lo_nxt = 0, hi_nxt = a.nets, lo_nx2 = 0, hi_nx2 = a.nets;
    if (dyn) w_cur = ((long long)blockIdx.x * 4 + uwave) * chunk;
    else seq(0, w_cur, lo_cur, hi_cur);
    if (w_cur >= nwt) return;
    int pend_raw = 0;   // lane 0: the counter value of the chunk requested ahead
    int left = chunk;   // wave-tiles left in the chunk the sequence is in (counting its head)
    auto request_chunk = [&]() {
        if (lane == 0) pend_raw = atomicAdd(a.tile_counter, 1);
    };
    // successor of wave-tile w in this wavefront's sequence (>= nwt: the sequence has ended), with its nets
    auto next_wt = [&](long long w, int& lo, int& hi) -> long long {
        if (!dyn) {
            long long s;
            seq(++pos, s, lo, hi);
            return s;
        }
        lo = 0;
        hi = a.nets;
        if (--left > 0) return w + 1;
        left = chunk;
        const long long s = (nwaves + __builtin_amdgcn_readfirstlane(pend_raw)) * chunk;
        request_chunk();
        return s;
    };
    if (dyn) request_chunk();
    long long w_nxt = next_wt(w_cur, lo_nxt, hi_nxt), w_nx2 = 0;

    // the ring: next 8 quads of the stream (of the first item's first net), refilled 8 quads ahead, never drained
    f32x4 ring[LIDF_RING];
#pragma unroll
    for (int i = 0; i < LIDF_RING; ++i) ring[i] = LDQ(srs, vq, lo_cur * net_bytes + i * 1024);

    // addresses = wave-uniform base (SGPRs) + a 32-bit lane offset: no 64-bit pointer registers
    auto load_idx = [&](long long wt, Geo& g) {
        const long long last = AN - 1;
        long long t0 = wt * 32;                     // uniform
        if (t0 > last) t0 = last & ~31LL;           // a prefetch past the end re-reads the last tile
        const long long rem = last - t0;
        const int lo = col;
        const unsigned off = (unsigned)(lo < rem ? lo : rem);  // out-of-range points are clamped
        g.ray = (a.pair_ray + t0)[off];
        g.vid = (a.pair_vox + t0)[off];
        const f32x2 tt = *(const f32x2*)((const char*)(a.pair_t + 2 * t0) + 8u * off);
        g.te = tt[0];
        g.tl = tt[1];
    };
    auto load_dir = [&](Geo& g) {
        const char* rd = (const char*)a.ray_dir + 12u * (unsigned)g.ray;
        g.dx = *(const float*)(rd + 0);
        g.dy = *(const float*)(rd + 4);
        g.dz = *(const float*)(rd + 8);
    };
    // (the tables hold part_ld floats per row; this launch's nets start at column part_off — a launch may run
    // one net of a two-net table)
    const int part_ld = a.part_ld > 0 ? a.part_ld : a.nets * 256, part_off = a.part_ld > 0 ? a.part_off : 0;
    auto vox_row = [&](int vid, int net) {
        return a.voxpart + (size_t)vid * part_ld + part_off + net * 256 + 4 * h;
    };

    // two-stage geometry prefetch: `cur` complete, `nxt` has its indices (directions are fetched
    // one tile ahead, indices two tiles ahead)
    Geo cur = {}, nxt = {}, nx2 = {};
    load_idx(w_cur, cur);
    load_idx(w_nxt, nxt);
    load_dir(cur);

    // layer-1 accumulator of the first (tile, net): fetched here, in the open
    f32x16 base[8];
    {
        const float* vp = vox_row(cur.vid, lo_cur);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = *(const f32x4*)(vp + t * 32 + 8 * g);
#pragma unroll
                for (int i = 0; i < 4; ++i) base[t][4 * g + i] = v[i];
            }
        }
    }

    PROF_DECL
    for (; w_cur < nwt; w_cur = w_nxt, w_nxt = w_nx2, lo_cur = lo_nxt, hi_cur = hi_nxt, lo_nxt = lo_nx2,
                       hi_nxt = hi_nx2) {
        PROF(0)
        const long long p = w_cur * 32 + col;
        const bool valid = p < AN;

        load_dir(nxt);
        w_nx2 = next_wt(w_nxt, lo_nx2, hi_nx2);
        load_idx(w_nx2, nx2);
        // this half's embedding input: lanes 0..31 embed the enter position, lanes 32..63 the
        // leave position (pipeline.py:349-360; 'rel' subtracts the voxel centre)
        const float tt = h ? cur.tl : cur.te;
        float px = __fmul_rn(cur.dx, tt);
        float py = __fmul_rn(cur.dy, tt);
        float pz = __fmul_rn(cur.dz, tt);
        if (a.pos_rel) {
            px -= a.vox_center[3 * (size_t)cur.vid + 0];
            py -= a.vox_center[3 * (size_t)cur.vid + 1];
            pz -= a.vox_center[3 * (size_t)cur.vid + 2];
        }
        const Rev rx = to_rev(px), ry = to_rev(py), rz = to_rev(pz);
        PROF(1)

        for (int net = lo_cur; net < hi_cur; ++net) {
            const int nsb = net * net_bytes;  // byte offset of this net's block
            // (what the ring reads next: the other net of this item, or the first net of the next item)
            const int next_blk = net + 1 < hi_cur ? nsb + net_bytes : lo_nxt * net_bytes;

            // ---------------- layer 1 ----------------
            // raypart[ray] is constant over the points of one ray, so it enters as a rank-1
            // update: A = raypart row (lanes 0..31: one ray, lanes 32..63: another), B = one-hot
            // membership of the point. Any pair order works; ray-major pairs need one round per
            // two distinct rays of the tile. The rows of the first LIDF_RB rounds are requested
            // now and used after the embedding steps.
            unsigned todo = (unsigned)__ballot(h == 0);  // points still to be covered
            float ar[LIDF_RB][8], bsel[LIDF_RB];
            int nr = 0;
            auto issue_round = [&](float (&arow)[8], float& bs) {
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
                bs = (((h ? m1 : m0) >> col) & 1u) ? 1.f : 0.f;
                const float* rp = a.raypart + (size_t)(h ? r1 : r0) * part_ld + part_off + net * 256 + col;
#pragma unroll
                for (int t = 0; t < 8; ++t) arow[t] = rp[t * 32];
            };
#pragma unroll
            for (int j = 0; j < LIDF_RB; ++j) {
                if (todo) {
                    issue_round(ar[j], bsel[j]);
                    nr = j + 1;
                }
            }
            PROF(2)
            // octave pairs: 12 k-steps (sin xyz, cos xyz of octaves 2it, 2it+1) x 8 tiles per
            // iteration; the embedding values are produced right here, in registers
            float sc0 = 1.f;
            for (int it = 0; it < Lp; ++it) {
                const float sc1 = sc0 * 2.f;
                float sb[12];
                rev_sincos(rx, sc0, sb[0], sb[3]);
                rev_sincos(ry, sc0, sb[1], sb[4]);
                rev_sincos(rz, sc0, sb[2], sb[5]);
                rev_sincos(rx, sc1, sb[6], sb[9]);
                rev_sincos(ry, sc1, sb[7], sb[10]);
                rev_sincos(rz, sc1, sb[8], sb[11]);
                const int qb = nsb + (it * 24 + LIDF_RING) * 1024;
#pragma unroll
                for (int s = 0; s < 24; ++s) {
                    const int t = s / 3, jq = s % 3;
                    const f32x4 q = ring[s % LIDF_RING];
#if defined(LIDF_RING_VOFF)
                    ring[s % LIDF_RING] = LDQ(srs, vq + (s & 3) * 1024, qb + (s >> 2) * 4096);
#else
                    ring[s % LIDF_RING] = LDQ(srs, vq, qb + s * 1024);
#endif
                    f32x16 c = base[t];
                    c = MFMA(q[0], sb[4 * jq + 0], c);
                    c = MFMA(q[1], sb[4 * jq + 1], c);
                    c = MFMA(q[2], sb[4 * jq + 2], c);
                    c = MFMA(q[3], sb[4 * jq + 3], c);
                    base[t] = c;
                    SCHED_FENCE();
                }
                sc0 = sc1 * 2.f;
            }
            {
                // tail: raw x, y, z; the refills run on into the u / layer-2 quads of this block
                const int qb = nsb + (Lp * 24 + LIDF_RING) * 1024;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const f32x4 q = ring[t];
                    ring[t] = LDQ(srs, vq, qb + t * 1024);
                    f32x16 c = base[t];
                    c = MFMA(q[0], px, c);
                    c = MFMA(q[1], py, c);
                    c = MFMA(q[2], pz, c);
                    base[t] = c;
                    SCHED_FENCE();
                }
            }
            // the rank-1 rounds (rows requested before the embedding steps); more than
            // 2*LIDF_RB distinct rays in the tile: further batches, in the open
            for (;;) {
#pragma unroll
                for (int j = 0; j < LIDF_RB; ++j) {
                    if (j < nr) {
#pragma unroll
                        for (int t = 0; t < 8; ++t) base[t] = MFMA(ar[j][t], bsel[j], base[t]);
                    }
                }
                if (!todo) break;
                nr = 0;
#pragma unroll
                for (int j = 0; j < LIDF_RB; ++j) {
                    if (todo) {
                        issue_round(ar[j], bsel[j]);
                        nr = j + 1;
                    }
                }
            }
            PROF(3)

            // ---------------- passes (1 for IMNet, n_iter for IEF) ----------------
            // the last pass of the net copies the accumulator init of what runs next — the other net
            // of this tile, or the first net of the next tile — into LDS while its layer 2 runs
            const bool last_net = net + 1 == hi_cur;
            const unsigned vp_off =
                (unsigned)(((last_net ? nxt.vid : cur.vid) * part_ld + part_off + (last_net ? lo_nxt : net + 1) * 256) * 4 +
                           16 * h);
            float val = a.init[net];
            const int pass_base = nsb + l1_bytes;
            const float* ax = a.aux + net * LIDF_AUX_FLOATS;
            const int npass = a.npass[net];
            // (ST: rows beyond n repeat point n-1 — load_idx clamps — so their stores repeat its values)
            const long long pc = valid ? p : AN - 1;
            auto train_row = [&](int pass, float v) {
                TrainRow tr = {};
                if (ST) {
                    float* pk = a.tr_passes[net] + (size_t)pass * a.tr_pass_floats;
                    tr.h1 = pk + (size_t)pc * LIDF_H1 + 4 * h;
                    tr.h2 = pk + (size_t)AN * LIDF_H1 + (size_t)pc * LIDF_H2 + 4 * h;
                    tr.h3 = pk + (size_t)AN * (LIDF_H1 + LIDF_H2) + (size_t)pc * LIDF_H3 + 4 * h;
                    tr.m1 = (unsigned*)(pk + (size_t)AN * LIDF_ACT_M1) + (size_t)pc * 8 + 4 * h;
                    tr.m2 = (unsigned*)(pk + (size_t)AN * LIDF_ACT_M2) + (size_t)pc * 4 + 2 * h;
                    if (valid && h == 0) pk[(size_t)AN * (LIDF_H1 + LIDF_H2 + LIDF_H3) + p] = v;
                }
                return tr;
            };
            for (int pass = 0; pass + 1 < npass; ++pass)
                val += decoder_pass<false, ST>(srs, vq, ring, pass_base, pass_base, base, val, h, one_b,
                                               ax, nullptr, 0u, 0u, nullptr, 0, train_row(pass, val)
#ifdef LIDF_PROFILE
                                               , prof_t, prof_acc
#endif
                                               );
            val += decoder_pass<true, ST>(srs, vq, ring, pass_base, next_blk, base, val, h, one_b, ax,
                                          (const char*)a.voxpart, vp_off, lds_addr, lds_wave, lane,
                                          train_row(npass - 1, val)
#ifdef LIDF_PROFILE
                                          , prof_t, prof_acc
#endif
                                          );
            // `base` now holds the accumulator init of the next (net, tile)
            PROF(5)
            // ---------------- outputs ----------------
            if (valid && h == 0) {
                const float o = out_act(val, a.sigmoid[net]);
                if (a.out[net]) a.out[net][p] = o;
                if (ST) a.tr_pre[net][p] = val;
                if (a.is_offset[net] && a.pair_pred_pos) {
                    // pipeline.py:437-439, same operation order in f32
                    const float ex = __fmul_rn(cur.dx, cur.te);
                    const float ey = __fmul_rn(cur.dy, cur.te);
                    const float ez = __fmul_rn(cur.dz, cur.te);
                    float sc = __fadd_rn(__fmul_rn(o, a.rscale), a.r0);
                    sc = __fmul_rn(__fmul_rn(sc, a.sqrt3), a.part_size);
                    a.pair_pred_pos[3 * p + 0] = __fadd_rn(ex, __fmul_rn(sc, cur.dx));
                    a.pair_pred_pos[3 * p + 1] = __fadd_rn(ey, __fmul_rn(sc, cur.dy));
                    a.pair_pred_pos[3 * p + 2] = __fadd_rn(ez, __fmul_rn(sc, cur.dz));
                }
            }
        }
        PROF(0)
        cur = nxt;
        nxt.ray = nx2.ray;
        nxt.vid = nx2.vid;
        nxt.te = nx2.te;
        nxt.tl = nx2.tl;
    }
    PROF_DUMP
}

__global__ void __launch_bounds__(256) lidf_points_fused_kernel(PointsArgs a) { points_fused_body<false>(a); }
__global__ void __launch_bounds__(256) lidf_points_fused_train_kernel(PointsArgs a) { points_fused_body<true>(a); }

// ------------------------------------------------------------------------------------------------
// Rows modes: the decoders (LIDF_MODE_ROWS) or layer 1 only (LIDF_MODE_L1ONLY: voxpart / raypart
// producers) on materialised input rows.
// ------------------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ void points_rows_body(const PointsArgs& a, const int bx, const int gx,
                                                 const int net_lo, const int net_hi) {
    // row / point count: a device-side count (a.n_dev: the sync-free frame path) overrides the host value
    const long long AN = a.n_dev ? (long long)*a.n_dev : a.n;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int h = lane >> 5;
    const int col = lane & 31;
    const float one_b = h ? 0.f : 1.f;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.stream, 0, a.nets * a.net_quads * 1024, 0x00020000);
    const int vq = lane * 16;
    const int net_bytes = a.net_quads * 1024;
    const int l1_bytes = a.l1_quads * 1024;

    // contiguous range of 128-row tiles per workgroup; the 4 waves interleave inside it
    const long long ntile = (AN + 127) / 128;
    const long long per = ntile / gx, rem = ntile % gx;
    const long long tb = bx * per + (bx < rem ? bx : rem);
    const long long te_ = tb + per + (bx < rem ? 1 : 0);
    if (tb >= te_) return;

    // the ring: next 8 quads of the stream, refilled 8 quads ahead, never drained
    f32x4 ring[LIDF_RING];
#pragma unroll
    for (int i = 0; i < LIDF_RING; ++i) ring[i] = LDQ(srs, vq, net_lo * net_bytes + i * 1024);

    PROF_DECL
    for (long long tile = tb; tile < te_; ++tile) {
        if (tile * 128 + wave * 32 >= AN) break;  // whole wave out of range (wave-uniform)
        const long long p = tile * 128 + wave * 32 + col;
        const bool valid = p < AN;
        const long long pc = valid ? p : AN - 1;

        for (int net = net_lo; net < net_hi; ++net) {
            const int nsb = net * net_bytes;  // byte offset of this net's block
            const int next_blk = net + 1 < net_hi ? nsb + net_bytes : net_lo * net_bytes;
            f32x16 base[8];

            // ---------------- layer 1 ----------------
            // TRAIN with gathered rows: the accumulators start from voxpart[pair_vox] (loaded
            // straight into them) and the raypart[pair_ray] row is requested now, added after the
            // matrix instructions of layer 1 — both latencies hide behind them
            // (ROWS_GATHER: the voxel-part rows only — stage 2's IEF, whose voxel feature of the
            // end voxel is a per-voxel product)
            f32x4 rpv[MODE == LIDF_MODE_TRAIN ? 8 : 1][4];
            const bool gathered = (MODE == LIDF_MODE_TRAIN || MODE == LIDF_MODE_ROWS_GATHER) && a.voxpart;
            if (gathered) {
                const float* vp = a.voxpart + (size_t)a.pair_vox[pc] * 256 + 4 * h;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 v = *(const f32x4*)(vp + 32 * t + 8 * g);
#pragma unroll
                        for (int i = 0; i < 4; ++i) base[t][4 * g + i] = v[i];
                    }
                }
                if constexpr (MODE == LIDF_MODE_ROWS_GATHER) {
                    // stage 2's IEF: the per-ray part of layer 1 (ROI + direction columns, constant over
                    // the refine iterations) is a [n, 256] table too — row = the ray itself
                    if (a.raypart) {
                        const float* rp = a.raypart + (size_t)(a.pair_ray ? a.pair_ray[pc] : pc) * 256 + 4 * h;
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const f32x4 v = *(const f32x4*)(rp + 32 * t + 8 * g);
#pragma unroll
                                for (int i = 0; i < 4; ++i) base[t][4 * g + i] += v[i];
                            }
                        }
                    }
                }
                if constexpr (MODE == LIDF_MODE_TRAIN) {
                    const float* rp = a.raypart + (size_t)a.pair_ray[pc] * 256 + 4 * h;
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) rpv[t][g] = *(const f32x4*)(rp + 32 * t + 8 * g);
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) base[t][i] = 0.f;
                }
            }
            PROF(0)
            // operand columns of this lane in k-quad kq: 8kq + 4h + {0..3}; column D = bias
            const float* xrow = a.X + (size_t)pc * a.ldx + 4 * h;
            auto load_b = [&](int kq, float (&b)[4]) {
                const int x0 = 8 * kq + 4 * h;
                if (x0 + 3 < a.D) {
                    const f32x4u v = *(const f32x4u*)(xrow + 8 * kq);
                    b[0] = v[0]; b[1] = v[1]; b[2] = v[2]; b[3] = v[3];
                } else {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                        b[jj] = x0 + jj < a.D ? xrow[8 * kq + jj]
                                              : ((x0 + jj == a.D && a.has_bias) ? 1.f : 0.f);
                }
            };
            // where the stream continues after this layer-1 section
            const int after_l1 = MODE == LIDF_MODE_L1ONLY ? next_blk : nsb + l1_bytes;
            // The operand rows stream from HBM, and VMEM loads return in order: an HBM-latency
            // load would hold up every weight-ring load queued behind it, once per iteration.
            // So the operands of up to 25 k-quads (100 VGPRs, free during layer 1) are fetched
            // in one burst per chunk and the iterations run with only the L2-resident ring in
            // the queue.
            constexpr int XCH = MODE == LIDF_MODE_ROWS ? 25 : (MODE == LIDF_MODE_TRAIN || MODE == LIDF_MODE_ROWS_GATHER) ? 13 : 8;  // L1ONLY: keep 2 waves/SIMD
            for (int k0 = 0; k0 < a.KQ1; k0 += XCH) {
                float xb[XCH][4];
#pragma unroll
                for (int i = 0; i < XCH; ++i) {
                    if (k0 + i < a.KQ1) {
                        load_b(k0 + i, xb[i]);
                    } else {
                        xb[i][0] = xb[i][1] = xb[i][2] = xb[i][3] = 0.f;
                    }
                }
                SCHED_FENCE();
                PROF(1)
#pragma unroll
                for (int i = 0; i < XCH; ++i) {
                    const int kq = k0 + i;
                    if (kq >= a.KQ1) break;
                    const int qb = kq + 1 < a.KQ1 ? nsb + (kq + 1) * 8 * 1024 : after_l1;
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const f32x4 q = ring[t];
                        ring[t] = LDQ(srs, vq, qb + t * 1024);
                        f32x16 c = base[t];
                        c = MFMA(q[0], xb[i][0], c);
                        c = MFMA(q[1], xb[i][1], c);
                        c = MFMA(q[2], xb[i][2], c);
                        c = MFMA(q[3], xb[i][3], c);
                        base[t] = c;
                        SCHED_FENCE();
                    }
                }
            }

            PROF(3)
            if (MODE == LIDF_MODE_TRAIN && gathered) {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            base[t][4 * g + i] += rpv[MODE == LIDF_MODE_TRAIN ? t : 0][g][i];
                    }
                }
            }
            if constexpr (MODE == LIDF_MODE_L1ONLY) {
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
            } else {
                PROF(2)
                // ---------------- passes (1 for IMNet, n_iter for IEF) ----------------
                float val = a.init[net];
                const int pass_base = nsb + l1_bytes;
                const float* ax = a.aux + net * LIDF_AUX_FLOATS;
                const int npass = a.npass[net];
                for (int pass = 0; pass < npass; ++pass) {
                    const int wrap_base = pass + 1 < npass ? pass_base : next_blk;
                    TrainRow tr = {};
                    if constexpr (MODE == LIDF_MODE_TRAIN) {
                        // rows beyond n repeat row n-1 (same operands, same values): no guard needed
                        float* pk = a.tr_passes[net] + (size_t)pass * a.tr_pass_floats;
                        tr.h1 = pk + (size_t)pc * LIDF_H1 + 4 * h;
                        tr.h2 = pk + (size_t)AN * LIDF_H1 + (size_t)pc * LIDF_H2 + 4 * h;
                        tr.h3 = pk + (size_t)AN * (LIDF_H1 + LIDF_H2) + (size_t)pc * LIDF_H3 + 4 * h;
                        tr.m1 = (unsigned*)(pk + (size_t)AN * LIDF_ACT_M1) + (size_t)pc * 8 + 4 * h;
                        tr.m2 = (unsigned*)(pk + (size_t)AN * LIDF_ACT_M2) + (size_t)pc * 4 + 2 * h;
                        if (valid && h == 0) pk[(size_t)AN * (LIDF_H1 + LIDF_H2 + LIDF_H3) + p] = val;
                    }
                    val += decoder_pass<false, MODE == LIDF_MODE_TRAIN>(
                        srs, vq, ring, pass_base, wrap_base, base, val, h, one_b, ax, nullptr, 0u, 0u,
                        nullptr, 0, tr
#ifdef LIDF_PROFILE
                        , prof_t, prof_acc
#endif
                        );
                }
                // ---------------- outputs ----------------
                if (valid && h == 0) {
                    const float o = out_act(val, a.sigmoid[net]);
                    if (a.out[net]) a.out[net][p] = o;
                    if constexpr (MODE == LIDF_MODE_TRAIN) a.tr_pre[net][p] = val;
                }
                PROF(5)
            }
        }
    }
    PROF_DUMP
}

template <int MODE>
__global__ void __launch_bounds__(256) lidf_points_kernel(PointsArgs a) {
    points_rows_body<MODE>(a, blockIdx.x, gridDim.x, 0, a.nets);
}

// ------------------------------------------------------------------------------------------------
// Layer-1 partial products of the query in work items of (128-row tile, net, half of the 256
// outputs): four times the items of a (tile, both nets) split, so the last round of a launch that
// covers only a few rounds (600 ray tiles on 256 CUs) is well filled; 64 accumulator registers per
// wavefront instead of 128 (more wavefronts per SIMD hide the operand-row latency); the 32 x 32
// output tiles leave through LDS as 128-byte row segments (direct stores would touch 64 lines per
// instruction with 16 bytes each: raypart is 157 MB per frame). Stream: the rows-mode layer-1
// layout with 8 tiles per k-quad (lidf_device.h); half hf reads the quads kq*8 + 4hf + {0..3}.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void l1part_item(const PointsArgs& a, const long long AN, const long long tile,
                                            const int net, const int hf, float* s_stage) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int h = lane >> 5;
    const int col = lane & 31;
    const int net_bytes = a.net_quads * 1024;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.stream, 0, a.nets * net_bytes, 0x00020000);
    const int vq = lane * 16;
    const int sbase = net * net_bytes + 4 * hf * 1024;
    float* stage = s_stage + wave * (32 * 33);
    {
        if (tile * 128 + wave * 32 >= AN) return;
        const long long p = tile * 128 + wave * 32 + col;
        const long long pc = p < AN ? p : AN - 1;
        const float* xrow = a.X + (size_t)pc * a.ldx + 4 * h;
        auto load_b = [&](int kq, float (&b)[4]) {
            const int x0 = 8 * kq + 4 * h;
            if (x0 + 3 < a.D) {
                const f32x4u v = *(const f32x4u*)(xrow + 8 * kq);
                b[0] = v[0]; b[1] = v[1]; b[2] = v[2]; b[3] = v[3];
            } else {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    b[jj] = x0 + jj < a.D ? xrow[8 * kq + jj]
                                          : ((x0 + jj == a.D && a.has_bias) ? 1.f : 0.f);
            }
        };
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        }
        // operand rows four k-quads ahead (they stream from HBM), weight quads one k-quad ahead
        constexpr int XD = 4;
        float xr[XD][4];
#pragma unroll
        for (int i = 0; i < XD; ++i) {
            if (i < a.KQ1) load_b(i, xr[i]);
            else xr[i][0] = xr[i][1] = xr[i][2] = xr[i][3] = 0.f;
        }
        f32x4 qc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) qc[t] = LDQ(srs, vq, sbase + t * 1024);
        for (int kq0 = 0; kq0 < a.KQ1; kq0 += XD) {
#pragma unroll
            for (int i = 0; i < XD; ++i) {
                const int kq = kq0 + i;
                if (kq >= a.KQ1) break;
                const float bc[4] = {xr[i][0], xr[i][1], xr[i][2], xr[i][3]};
                if (kq + XD < a.KQ1) load_b(kq + XD, xr[i]);
                f32x4 qn[4];
                const int kn = kq + 1 < a.KQ1 ? kq + 1 : kq;
#pragma unroll
                for (int t = 0; t < 4; ++t) qn[t] = LDQ(srs, vq, sbase + (kn * 8 + t) * 1024);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    f32x16 c = acc[t];
                    c = MFMA(qc[t][0], bc[0], c);
                    c = MFMA(qc[t][1], bc[1], c);
                    c = MFMA(qc[t][2], bc[2], c);
                    c = MFMA(qc[t][3], bc[3], c);
                    acc[t] = c;
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) qc[t] = qn[t];
                SCHED_FENCE();
            }
        }
        // register 4g+i of lane (col, h) in tile t = output 32(4hf+t) + 8g + 4h + i of row `col`
        float* ob = a.out_base + ((size_t)(tile * 128 + wave * 32) * a.nets + net) * 256 + 128 * hf;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int i = 0; i < 4; ++i) stage[col * 33 + 8 * g + 4 * h + i] = acc[t][4 * g + i];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = 8 * j + (lane >> 3), f4 = (lane & 7) * 4;
                f32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = stage[row * 33 + f4 + i];
                if (tile * 128 + wave * 32 + row < AN)
                    *(f32x4*)(ob + (size_t)row * a.nets * 256 + 32 * t + f4) = o;
            }
        }
    }
}

// The same item with the operand rows of its tile already in registers (round 4): the 2 * nets parts of a
// tile are adjacent in the flat list, so a workgroup's contiguous run of items re-used the same 32 rows
// per wavefront up to six times — fetched again for every part (from L2, behind a dependent address
// chain at the head of each item). Here lane (col, h) keeps its 4 columns of every k-quad (<= L1X_KQ
// k-quads: 80 registers at D = 155) across the items of a tile; weight quads run two k-quads ahead.
// Same products in the same order as l1part_item: bit-identical rows.
#define L1X_KQ 20
__device__ __forceinline__ void l1part_load_x(const PointsArgs& a, const long long AN, const long long tile,
                                              f32x4 (&xr)[L1X_KQ]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, col = lane & 31;
    if (tile * 128 + wave * 32 >= AN) return;
    const long long p = tile * 128 + wave * 32 + col;
    const long long pc = p < AN ? p : AN - 1;
    const float* xrow = a.X + (size_t)pc * a.ldx + 4 * h;
#pragma unroll
    for (int kq = 0; kq < L1X_KQ; ++kq) {
        xr[kq] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int x0 = 8 * kq + 4 * h;
        if (kq < a.KQ1) {
            if (x0 + 3 < a.D) {
                const f32x4u v = *(const f32x4u*)(xrow + 8 * kq);
                xr[kq] = f32x4{v[0], v[1], v[2], v[3]};
            } else {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    xr[kq][jj] = x0 + jj < a.D ? xrow[8 * kq + jj] : ((x0 + jj == a.D && a.has_bias) ? 1.f : 0.f);
            }
        }
    }
}

__device__ __forceinline__ void l1part_item_x(const PointsArgs& a, const long long AN, const long long tile,
                                              const int net, const int hf, const f32x4 (&xr)[L1X_KQ],
                                              float* s_stage) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int h = lane >> 5;
    const int col = lane & 31;
    const int net_bytes = a.net_quads * 1024;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.stream, 0, a.nets * net_bytes, 0x00020000);
    const int vq = lane * 16;
    const int sbase = net * net_bytes + 4 * hf * 1024;
    float* stage = s_stage + wave * (32 * 33);
    if (tile * 128 + wave * 32 >= AN) return;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    }
    // weight quads two k-quads ahead (reads beyond the net's section are never multiplied)
    f32x4 q[3][4];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
#pragma unroll
        for (int t = 0; t < 4; ++t) q[d][t] = LDQ(srs, vq, sbase + (d * 8 + t) * 1024);
    }
#pragma unroll
    for (int kq = 0; kq < L1X_KQ; ++kq) {
        if (kq < a.KQ1) {
            if (kq + 2 < a.KQ1) {
#pragma unroll
                for (int t = 0; t < 4; ++t) q[(kq + 2) % 3][t] = LDQ(srs, vq, sbase + ((kq + 2) * 8 + t) * 1024);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x16 c = acc[t];
                c = MFMA(q[kq % 3][t][0], xr[kq][0], c);
                c = MFMA(q[kq % 3][t][1], xr[kq][1], c);
                c = MFMA(q[kq % 3][t][2], xr[kq][2], c);
                c = MFMA(q[kq % 3][t][3], xr[kq][3], c);
                acc[t] = c;
            }
            SCHED_FENCE();
        }
    }
    // register 4g+i of lane (col, h) in tile t = output 32(4hf+t) + 8g + 4h + i of row `col`
    float* ob = a.out_base + ((size_t)(tile * 128 + wave * 32) * a.nets + net) * 256 + 128 * hf;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int i = 0; i < 4; ++i) stage[col * 33 + 8 * g + 4 * h + i] = acc[t][4 * g + i];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = 8 * j + (lane >> 3), f4 = (lane & 7) * 4;
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = stage[row * 33 + f4 + i];
            if (tile * 128 + wave * 32 + row < AN)
                *(f32x4*)(ob + (size_t)row * a.nets * 256 + 32 * t + f4) = o;
        }
    }
}

// Flat list of items (problem a, then b, then c; the 2*nets parts of a tile adjacent, so that the
// operand rows of a tile are loaded once per wavefront for the parts a workgroup's run holds), cut into
// equal contiguous runs over the grid. c (optional, c.stream == NULL: none) is a third table with its
// own stream — the frame path's per-ray part of the stage-2 decoder's layer 1, which reads the same
// rayfeat rows as the query's raypart (a launch of its own over 76,800 rows before).
struct L1Problems {
    PointsArgs p[3];   // p[k].stream == NULL: absent
};
__device__ __forceinline__ long long l1_rows(const PointsArgs& a) { return a.n_dev ? (long long)*a.n_dev : a.n; }

// (one instance of the item code: the problem is picked by a uniform index into the kernel argument)
__global__ void __launch_bounds__(256, 2) lidf_l1part_pair_kernel(L1Problems P) {
    __shared__ float s_stage[4 * 32 * 33];
    long long n[3], cnt[3];
    int parts[3];
    long long tot = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        parts[k] = P.p[k].stream ? P.p[k].nets * 2 : 0;
        n[k] = parts[k] ? l1_rows(P.p[k]) : 0;
        cnt[k] = (n[k] + 127) / 128 * parts[k];
        tot += cnt[k];
    }
    const long long per = tot / gridDim.x, rem = tot % gridDim.x;
    const long long bx = blockIdx.x;
    const long long ib = bx * per + (bx < rem ? bx : rem), ie = ib + per + (bx < rem ? 1 : 0);
    f32x4 xr[L1X_KQ];
    long long have = -1;   // (problem, tile) whose rows xr holds
    for (long long i = ib; i < ie; ++i) {
        const int k = i < cnt[0] ? 0 : (i < cnt[0] + cnt[1] ? 1 : 2);
        const long long j = i - (k > 0 ? cnt[0] : 0) - (k > 1 ? cnt[1] : 0);
        const int pk = k == 0 ? parts[0] : (k == 1 ? parts[1] : parts[2]);
        const long long nk = k == 0 ? n[0] : (k == 1 ? n[1] : n[2]);
        const long long tile = j / pk, key = ((long long)k << 40) + tile;
        const int part = (int)(j % pk);
        const PointsArgs& a = P.p[k];
        if (key != have) {
            l1part_load_x(a, nk, tile, xr);
            have = key;
        }
        l1part_item_x(a, nk, tile, part >> 1, part & 1, xr, s_stage);
    }
}

// wider embeddings (multires_views > 4: more than L1X_KQ k-quads per row): operand rows streamed per item
__global__ void __launch_bounds__(256) lidf_l1part_pair_stream_kernel(L1Problems P) {
    __shared__ float s_stage[4 * 32 * 33];
    long long n[3], cnt[3];
    int parts[3];
    long long tot = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        parts[k] = P.p[k].stream ? P.p[k].nets * 2 : 0;
        n[k] = parts[k] ? l1_rows(P.p[k]) : 0;
        cnt[k] = (n[k] + 127) / 128 * parts[k];
        tot += cnt[k];
    }
    const long long per = tot / gridDim.x, rem = tot % gridDim.x;
    const long long bx = blockIdx.x;
    const long long ib = bx * per + (bx < rem ? bx : rem), ie = ib + per + (bx < rem ? 1 : 0);
    for (long long i = ib; i < ie; ++i) {
        const int k = i < cnt[0] ? 0 : (i < cnt[0] + cnt[1] ? 1 : 2);
        const long long j = i - (k > 0 ? cnt[0] : 0) - (k > 1 ? cnt[1] : 0);
        const int pk = k == 0 ? parts[0] : (k == 1 ? parts[1] : parts[2]);
        const long long nk = k == 0 ? n[0] : (k == 1 ? n[1] : n[2]);
        const int part = (int)(j % pk);
        l1part_item(P.p[k], nk, j / pk, part >> 1, part & 1, s_stage);
    }
}

template <int MODE>
static hipError_t launch_points(const PointsArgs& a, int grid, hipStream_t st) {
    hipLaunchKernelGGL((lidf_points_kernel<MODE>), dim3(grid), dim3(256), 0, st, a);
    return hipGetLastError();
}

extern "C" hipError_t lidf_launch_l1only_pair(const PointsArgs& a, const PointsArgs& b, const PointsArgs* c,
                                               int cus, hipStream_t st) {
    L1Problems P = {};
    P.p[0] = a;
    P.p[1] = b;
    if (c) P.p[2] = *c;
    long long items = 0;
    bool fits = true;
    for (int k = 0; k < 3; ++k) {
        if (!P.p[k].stream) continue;
        items += (P.p[k].n + 127) / 128 * P.p[k].nets * 2;
        fits = fits && P.p[k].KQ1 <= L1X_KQ;
    }
    if (items <= 0) return hipSuccess;
    // two workgroups per CU: 9600 wavefront items of a 240x320 frame leave 10 per SIMD at best (three or
    // four per CU — fewer, shorter rounds on paper — measured the same: 1.761 / 1.763 / 1.763 ms per frame)
    const int grid = (int)(items < 2LL * cus ? items : 2LL * cus);
    if (fits)
        hipLaunchKernelGGL(lidf_l1part_pair_kernel, dim3(grid), dim3(256), 0, st, P);
    else
        hipLaunchKernelGGL(lidf_l1part_pair_stream_kernel, dim3(grid), dim3(256), 0, st, P);
    return hipGetLastError();
}

extern "C" hipError_t lidf_launch_points(int mode, const PointsArgs& a, int grid,
                                         hipStream_t st) {
    if (a.n <= 0) return hipSuccess;
    if (mode == LIDF_MODE_FUSED) {
        // 4 waves x 32 KiB of LDS: the staging area of the accumulator prefetch
        static_assert(4 * 8192 * 4 == 131072, "LDS staging size");
        static bool configured[64];
        hipError_t e = lidf_max_lds_once(configured, (const void*)lidf_points_fused_kernel, 131072);
        if (e != hipSuccess) return e;
        static int env_min = -1, env_chunk = -1, env_split = -1;   // development knobs of the tile hand-out
        if (env_min < 0) {
            const char* e1 = getenv("LIDF_DYN_MIN");
            const char* e2 = getenv("LIDF_DYN_CHUNK");
            const char* e3 = getenv("LIDF_TAIL_SPLIT");   // 0: the partial last round's tiles stay whole (A/B)
            env_min = e1 ? atoi(e1) : 0;
            env_chunk = e2 ? atoi(e2) : 0;
            env_split = (e3 && e3[0] == '0') ? 0 : 1;
        }
        if (!a.tr_passes[0] && !a.tr_passes[1]) {
            PointsArgs b = a;
            if (env_min > 0 || env_chunk > 0) {
                b.dyn_min_tiles = env_min;
                b.dyn_chunk = env_chunk;
            }
            b.tail_split = env_split;
            hipLaunchKernelGGL(lidf_points_fused_kernel, dim3(grid), dim3(256), 131072, st, b);
            return hipGetLastError();
        }
        if (a.tr_passes[0] || a.tr_passes[1]) {
            static bool configured_t[64];
            e = lidf_max_lds_once(configured_t, (const void*)lidf_points_fused_train_kernel, 131072);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(lidf_points_fused_train_kernel, dim3(grid), dim3(256), 131072, st, a);
        } else {
            hipLaunchKernelGGL(lidf_points_fused_kernel, dim3(grid), dim3(256), 131072, st, a);
        }
        return hipGetLastError();
    }
    if (mode == LIDF_MODE_ROWS) return launch_points<LIDF_MODE_ROWS>(a, grid, st);
    if (mode == LIDF_MODE_L1ONLY) return launch_points<LIDF_MODE_L1ONLY>(a, grid, st);
    if (mode == LIDF_MODE_TRAIN) return launch_points<LIDF_MODE_TRAIN>(a, grid, st);
    if (mode == LIDF_MODE_ROWS_GATHER) return launch_points<LIDF_MODE_ROWS_GATHER>(a, grid, st);
    return hipErrorInvalidValue;
}
