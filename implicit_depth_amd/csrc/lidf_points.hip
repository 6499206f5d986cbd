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
#include "lidf_device.h"

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

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

__device__ float stream_value(const StreamLayout& lay, const NetW* nets, const L1Map& m, int e) {
    const int net = e / lay.net_floats;
    e %= lay.net_floats;
    const NetW& n = nets[net];
    if (e < lay.l1_floats) {  // ---- layer 1 (8 tiles)
        if (lay.mode == LIDF_MODE_FUSED) {
            const int per_oct = 8 * 384;
            int t, lane, feat;  // feat: index inside the (3+6L)-wide embedding, -1 = pad
            if (e < m.L * per_oct) {
                int o = e / per_oct, rem = e % per_oct;
                t = rem / 384;
                int r2 = rem % 384;
                int pair = r2 / 128;
                lane = (r2 % 128) / 2;
                int fr = 2 * pair + (r2 & 1);  // 0..5 = sin x,y,z, cos x,y,z
                feat = 3 + 6 * o + fr;
            } else {
                e -= m.L * per_oct;
                t = e / 256;
                lane = (e % 256) / 4;
                int jj = e & 3;
                feat = jj < 3 ? jj : -1;
            }
            if (feat < 0) return 0.f;
            int out = 32 * t + (lane & 31);
            int col = ((lane >> 5) ? m.leave_c0 : m.enter_c0) + feat;
            return n.w1[(size_t)out * n.ld1 + col];
        } else {
            int kq = e / (8 * 256), rem = e % (8 * 256);
            int t = rem / 256;
            int lane = (rem % 256) / 4;
            int s = 4 * kq + (rem & 3);
            int half = lane >> 5;
            int out = 32 * t + (lane & 31);
            int x = s + half * m.KH;  // operand column
            int nvalid = half ? (m.D - m.KH) : m.KH;
            if (s < nvalid) {
                int col = x < m.n0 ? m.c0 + x : m.c1 + (x - m.n0);
                return n.w1[(size_t)out * n.ld1 + col];
            }
            if (s == m.KH && half == 0 && m.add_bias) {
                float b = n.b1[out];
                if (n.is_ief) b += ief_c(n, out);
                return b;
            }
            return 0.f;
        }
    }
    e -= lay.l1_floats;
    if (e < LIDF_U_FLOATS) {  // ---- u fragments (zero for IMNet)
        int q = e / 256, lane = (e % 256) / 4, jj = e & 3;
        int to = 4 * q + jj;
        if ((lane >> 5) != 0 || !n.is_ief) return 0.f;
        return ief_u(n, 32 * to + (lane & 31));
    }
    e -= LIDF_U_FLOATS;
    if (e < LIDF_L2_FLOATS) {
        int kq = e / (4 * 256), t = (e / 256) % 4, lane = (e % 256) / 4;
        int s = 4 * kq + (e & 3), half = lane >> 5;
        int out = 32 * t + (lane & 31);
        if (s < LIDF_H1 / 2) return n.w2[(size_t)out * LIDF_H1 + k_to_feature(s, half)];
        if (s == LIDF_H1 / 2 && half == 0) return n.b2[out];
        return 0.f;
    }
    e -= LIDF_L2_FLOATS;
    {
        int kq = e / (2 * 256), t = (e / 256) % 2, lane = (e % 256) / 4;
        int s = 4 * kq + (e & 3), half = lane >> 5;
        int out = 32 * t + (lane & 31);
        if (s < LIDF_H2 / 2) return n.w3[(size_t)out * LIDF_H2 + k_to_feature(s, half)];
        if (s == LIDF_H2 / 2 && half == 0) return n.b3[out];
        return 0.f;
    }
}

__global__ void lidf_pack_kernel(StreamLayout lay, NetW net0, NetW net1, L1Map m, float* stream,
                                 float* aux) {
    NetW nets[2] = {net0, net1};
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < lay.total) stream[e] = stream_value(lay, nets, m, e);
    if (lay.mode != LIDF_MODE_L1ONLY && e < lay.nets * LIDF_AUX_FLOATS) {
        int sec = e / LIDF_AUX_FLOATS, i = e % LIDF_AUX_FLOATS;
        const NetW& n = nets[sec];
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
// Per-point kernel
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lrelu16(f32x16& v) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = fmaxf(v[i], v[i] * 0.02f);
}

__device__ __forceinline__ float out_act(float y, int use_sigmoid) {
    // implicit_net.py:93-96 / :148-151
    if (use_sigmoid) return 1.f / (1.f + expf(-y));
    return fmaxf(fminf(y, y * 0.01f + 0.99f), y * 0.01f);
}

#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// Weight-stream loads go through a buffer descriptor: the (wave-uniform) stream position lives in
// an SGPR and the lane offset is one constant VGPR, so the unrolled layers need no per-load
// 64-bit address registers (with flat pointers hipcc materialises and spills hundreds of them).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define LDQ(rs, voff, soff) \
    __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128((rs), (voff), (soff), 0))
#define LDP(rs, voff, soff) \
    __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64((rs), (voff), (soff), 0))

// Dense layer whose B operands are the previous layer's accumulators (KT input tiles -> NT output
// tiles), bias carried by k-step 16*KT.  The A fragments are streamed from L2 through an explicit
// DEPTH-deep register ring; sched_barrier pins the software pipeline (left alone, hipcc hoists
// every load of the unrolled layer to the top and spills them).
template <int NT, int KT, int DEPTH>
__device__ __forceinline__ void chain_layer(__amdgpu_buffer_rsrc_t srs, int sec, int lane,
                                            float one_b, const f32x16* __restrict__ Hin,
                                            f32x16* Hout) {
    constexpr int K = 16 * KT;     // k-steps: each consumes one register of both half-waves
    constexpr int KQ = K / 4 + 1;  // + the bias k-step
    const int vq = lane * 16;  // quad i of this lane lives at byte sec + 1024*i + 16*lane
    f32x4 ring[DEPTH][NT];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
        for (int t = 0; t < NT; ++t) ring[d][t] = LDQ(srs, vq, sec + (d * NT + t) * 1024);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int i = 0; i < 16; ++i) Hout[t][i] = 0.f;
    }
    SCHED_FENCE();
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq) {
        f32x4 cur[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) cur[t] = ring[kq % DEPTH][t];
        if (kq + DEPTH < KQ) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
                ring[kq % DEPTH][t] = LDQ(srs, vq, sec + ((kq + DEPTH) * NT + t) * 1024);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int s = 4 * kq + jj;
                if (s < K)
                    Hout[t] = MFMA(cur[t][jj], Hin[s / 16][s % 16], Hout[t]);
                else if (s == K)
                    Hout[t] = MFMA(cur[t][jj], one_b, Hout[t]);
            }
        }
        SCHED_FENCE();
    }
}

template <int MODE>
__global__ void __launch_bounds__(256) lidf_points_kernel(PointsArgs a) {
    extern __shared__ float pe_lds[];  // fused: [4 waves][6L+3 features][64 lanes]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int h = lane >> 5;
    const int col = lane & 31;
    const float one_b = h ? 0.f : 1.f;
    const int NF = 6 * a.L + 3;
    float* pe = pe_lds + wave * NF * 64 + lane;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.stream, 0, a.nets * a.net_floats * 4, 0x00020000);
    const int vq = lane * 16, vp2 = lane * 8;

    // contiguous range of 128-point tiles per workgroup; the 4 waves interleave inside it
    const long long ntile = (a.n + 127) / 128;
    const long long per = ntile / gridDim.x, rem = ntile % gridDim.x;
    const long long tb = blockIdx.x * per + (blockIdx.x < rem ? blockIdx.x : rem);
    const long long te_ = tb + per + (blockIdx.x < rem ? 1 : 0);

    for (long long tile = tb; tile < te_; ++tile) {
        if (tile * 128 + wave * 32 >= a.n) continue;  // whole wave out of range (wave-uniform)
        const long long p = tile * 128 + wave * 32 + col;
        const bool valid = p < a.n;
        const long long pc = valid ? p : a.n - 1;

        float ex = 0.f, ey = 0.f, ez = 0.f, dx = 0.f, dy = 0.f, dz = 0.f;
        int ray = 0, vid = 0;
        if constexpr (MODE == LIDF_MODE_FUSED) {
            ray = a.pair_ray[pc];
            vid = a.pair_vox[pc];
            const f32x2 tt = *(const f32x2*)(a.pair_t + 2 * pc);
            dx = a.ray_dir[3 * (size_t)ray + 0];
            dy = a.ray_dir[3 * (size_t)ray + 1];
            dz = a.ray_dir[3 * (size_t)ray + 2];
            // enter position (needed again for the output) and this half's embedding input:
            // lanes 0..31 embed the enter position, lanes 32..63 the leave position
            ex = __fmul_rn(dx, tt[0]);
            ey = __fmul_rn(dy, tt[0]);
            ez = __fmul_rn(dz, tt[0]);
            float px = h ? __fmul_rn(dx, tt[1]) : ex;
            float py = h ? __fmul_rn(dy, tt[1]) : ey;
            float pz = h ? __fmul_rn(dz, tt[1]) : ez;
            if (a.pos_rel) {
                px -= a.vox_center[3 * (size_t)vid + 0];
                py -= a.vox_center[3 * (size_t)vid + 1];
                pz -= a.vox_center[3 * (size_t)vid + 2];
            }
            float sc = 1.f;
            for (int o = 0; o < a.L; ++o) {
                float s0, s1, s2, c0, c1, c2;
                sincosf(px * sc, &s0, &c0);
                sincosf(py * sc, &s1, &c1);
                sincosf(pz * sc, &s2, &c2);
                float* w = pe + (6 * o) * 64;
                w[0] = s0; w[64] = s1; w[128] = s2; w[192] = c0; w[256] = c1; w[320] = c2;
                sc *= 2.f;
            }
            float* w = pe + (6 * a.L) * 64;
            w[0] = px; w[64] = py; w[128] = pz;
        }

        for (int net = 0; net < a.nets; ++net) {
            const int nsb = net * a.net_floats * 4;  // byte offset of this net's block
            f32x16 base[8];

            // ---------------- layer 1 ----------------
            if constexpr (MODE == LIDF_MODE_FUSED) {
                // accumulator init = voxpart[vid] + raypart[ray] (layer-1 bias inside voxpart),
                // one tile per scheduling region, next tile's 8 loads in flight
                const float* vp = a.voxpart + ((size_t)vid * a.nets + net) * 256 + 4 * h;
                const float* rp = a.raypart + ((size_t)ray * a.nets + net) * 256 + 4 * h;
                f32x2 wr[8][3];  // this octave's A pairs
#pragma unroll
                for (int t = 0; t < 8; ++t) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) wr[t][i] = LDP(srs, vp2, nsb + (t * 3 + i) * 512);
                }
                {
                    f32x4 v[4], r[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        v[g] = *(const f32x4*)(vp + 8 * g);
                        r[g] = *(const f32x4*)(rp + 8 * g);
                    }
                    SCHED_FENCE();
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        f32x4 vn[4], rn[4];
                        if (t + 1 < 8) {
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                vn[g] = *(const f32x4*)(vp + (t + 1) * 32 + 8 * g);
                                rn[g] = *(const f32x4*)(rp + (t + 1) * 32 + 8 * g);
                            }
                        }
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) base[t][4 * g + i] = v[g][i] + r[g][i];
                        }
                        if (t + 1 < 8) {
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                v[g] = vn[g];
                                r[g] = rn[g];
                            }
                        }
                        SCHED_FENCE();
                    }
                }
                float sb[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) sb[i] = pe[i * 64];
                for (int o = 0; o < a.L; ++o) {
                    // prefetch: next octave's embedding values (LDS) and A pairs (L2); after the
                    // last octave this fetches the raw position and the tail quads' first half
                    float nb[6];
#pragma unroll
                    for (int i = 0; i < 6; ++i) nb[i] = pe[((6 * (o + 1) + i) % NF) * 64];
                    f32x2 wn[8][3];
                    const int bpn = nsb + (o + 1 < a.L ? o + 1 : o) * (8 * 384 * 4);
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
#pragma unroll
                        for (int i = 0; i < 3; ++i) wn[t][i] = LDP(srs, vp2, bpn + (t * 3 + i) * 512);
                    }
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        f32x16 c = base[t];
                        c = MFMA(wr[t][0][0], sb[0], c);
                        c = MFMA(wr[t][0][1], sb[1], c);
                        c = MFMA(wr[t][1][0], sb[2], c);
                        c = MFMA(wr[t][1][1], sb[3], c);
                        c = MFMA(wr[t][2][0], sb[4], c);
                        c = MFMA(wr[t][2][1], sb[5], c);
                        base[t] = c;
                    }
#pragma unroll
                    for (int i = 0; i < 6; ++i) sb[i] = nb[i];
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
#pragma unroll
                        for (int i = 0; i < 3; ++i) wr[t][i] = wn[t][i];
                    }
                    SCHED_FENCE();
                }
                {
                    // after the loop sb[0..2] hold the raw position (features 6L..6L+2)
                    const int bp = nsb + a.L * (8 * 384 * 4);
                    f32x4 q[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) q[t] = LDQ(srs, vq, bp + t * 1024);
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        f32x16 c = base[t];
                        c = MFMA(q[t][0], sb[0], c);
                        c = MFMA(q[t][1], sb[1], c);
                        c = MFMA(q[t][2], sb[2], c);
                        base[t] = c;
                    }
                    SCHED_FENCE();
                }
            } else {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) base[t][i] = 0.f;
                }
                const float* xrow = a.X + (size_t)pc * a.ldx + (h ? a.KH : 0);
                const int nvalid = h ? (a.D - a.KH) : a.KH;
                const int sbias = (h == 0 && a.has_bias) ? a.KH : -1;
                float bc[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    bc[jj] = jj < nvalid ? xrow[jj] : (jj == sbias ? 1.f : 0.f);
                f32x4 qc[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) qc[t] = LDQ(srs, vq, nsb + t * 1024);
                SCHED_FENCE();
                for (int kq = 0; kq < a.KQ1; ++kq) {
                    float bn[4];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int s = 4 * (kq + 1) + jj;
                        bn[jj] = s < nvalid ? xrow[s] : (s == sbias ? 1.f : 0.f);
                    }
                    f32x4 qn[8];
                    const int kn = kq + 1 < a.KQ1 ? kq + 1 : kq;
#pragma unroll
                    for (int t = 0; t < 8; ++t) qn[t] = LDQ(srs, vq, nsb + (kn * 8 + t) * 1024);
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        f32x16 c = base[t];
                        c = MFMA(qc[t][0], bc[0], c);
                        c = MFMA(qc[t][1], bc[1], c);
                        c = MFMA(qc[t][2], bc[2], c);
                        c = MFMA(qc[t][3], bc[3], c);
                        base[t] = c;
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) bc[jj] = bn[jj];
#pragma unroll
                    for (int t = 0; t < 8; ++t) qc[t] = qn[t];
                    SCHED_FENCE();
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
                // ---------------- passes (1 for IMNet, n_iter for IEF) ----------------
                float val = a.init[net];
                const int us = nsb + a.l1_floats * 4;
                const int sec = us + LIDF_U_FLOATS * 4;
                const float* ax = a.aux + net * LIDF_AUX_FLOATS;
                const int npass = a.npass[net];
                for (int pass = 0; pass < npass; ++pass) {
                    // layer-1 activation: lrelu(base + u * val)  (u = 0 for IMNet)
                    const float ob = h ? 0.f : val;
                    const f32x4 u0 = LDQ(srs, vq, us);
                    const f32x4 u1 = LDQ(srs, vq, us + 1024);
                    f32x16 H1[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const float uf = t < 4 ? u0[t & 3] : u1[t & 3];
                        H1[t] = MFMA(uf, ob, base[t]);
                        lrelu16(H1[t]);
                    }
                    f32x16 H2[4];
                    chain_layer<4, 8, 2>(srs, sec, lane, one_b, H1, H2);
#pragma unroll
                    for (int t = 0; t < 4; ++t) lrelu16(H2[t]);
                    f32x16 H3[2];
                    chain_layer<2, 4, 3>(srs, sec + LIDF_L2_FLOATS * 4, lane, one_b, H2, H3);
#pragma unroll
                    for (int t = 0; t < 2; ++t) lrelu16(H3[t]);
                    // layer 4: 64 -> 1 on the VALU, halves combined with one cross-half shuffle
                    float y = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const f32x4 w = *(const f32x4*)(ax + h * 32 + 4 * i);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int s = 4 * i + k;
                            y = fmaf(w[k], H3[s / 16][s % 16], y);
                        }
                    }
                    y += __shfl_xor(y, 32);
                    y += ax[64];
                    val += y;
                }
                // ---------------- outputs ----------------
                if (valid && h == 0) {
                    const float o = out_act(val, a.sigmoid[net]);
                    if (a.out[net]) a.out[net][p] = o;
                    if constexpr (MODE == LIDF_MODE_FUSED) {
                        if (a.is_offset[net]) {
                            // pipeline.py:437-439, same operation order in f32
                            float s = __fadd_rn(__fmul_rn(o, a.rscale), a.r0);
                            s = __fmul_rn(__fmul_rn(s, a.sqrt3), a.part_size);
                            a.pair_pred_pos[3 * p + 0] = __fadd_rn(ex, __fmul_rn(s, dx));
                            a.pair_pred_pos[3 * p + 1] = __fadd_rn(ey, __fmul_rn(s, dy));
                            a.pair_pred_pos[3 * p + 2] = __fadd_rn(ez, __fmul_rn(s, dz));
                        }
                    }
                }
            }
        }
    }
}

template <int MODE>
static hipError_t launch_points(const PointsArgs& a, int grid, hipStream_t st) {
    size_t lds = MODE == LIDF_MODE_FUSED ? (size_t)4 * (6 * a.L + 3) * 64 * sizeof(float) : 0;
    hipLaunchKernelGGL((lidf_points_kernel<MODE>), dim3(grid), dim3(256), lds, st, a);
    return hipGetLastError();
}

extern "C" hipError_t lidf_launch_points(int mode, const PointsArgs& a, int grid,
                                         hipStream_t st) {
    if (a.n <= 0) return hipSuccess;
    if (mode == LIDF_MODE_FUSED) return launch_points<LIDF_MODE_FUSED>(a, grid, st);
    if (mode == LIDF_MODE_ROWS) return launch_points<LIDF_MODE_ROWS>(a, grid, st);
    if (mode == LIDF_MODE_L1ONLY) return launch_points<LIDF_MODE_L1ONLY>(a, grid, st);
    return hipErrorInvalidValue;
}
