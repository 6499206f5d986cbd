// lidf_chain16.hip — IMNet / IEF (models/implicit_net.py:60-152) at widths other than the shipped one as ONE
// register-chained launch per decoder: gf_dim = 32, 64 or 128 (layers 4 gf -> 2 gf -> gf -> 1) on
// v_mfma_f32_16x16x4_f32 sub-tiles of 16 rows — the formulation of lidf_ief16.hip (the stage-2 decoder, widths
// compiled in) with the tile counts as template parameters. Round 6: the layer-by-layer path of these widths
// (generic.py: one lidf_linear_kernel launch per layer and pass, every activation through HBM) ran the query at 0.21
// (gf 32) / 0.45 (gf 128) of the f32 matrix peak on its own FLOP.
//
// Layer 1 arrives factorised: the per-row operand X [n, E] (the query: the 2 (3 + 6 L)
// position-embedding columns of a (ray, voxel) pair; a materialised [n, D] input: all of it) is multiplied here, the
// columns that depend on the voxel / the ray alone arrive as gathered rows of two tables (voxpart carries b1 and the
// IEF's constant c = W1[:, enc] benc; either gather may be absent).
//
// G = gf / 16. Accumulator tiles of 16 features x 16 rows: T1 = 4 G (layer 1), T2 = 2 G (layer 2), T3 = G (layer 3).
// Lane l = (row j = l & 15, group g = l >> 4); register r of a tile T holds feature 16 T + 4 g + r of row j — the B
// operand of the next layer's k-step r of input tile T, so activations never leave the register file.
//
// Stream (LIDF_MODE_CHAIN16, chain16_stream_value in lidf_points.hip), 1 KiB quads consumed in order through a ring:
//   layer 1: for kq < KQ = ceil(E / 16), To < T1: lane l: W1[16 To + j][c0 + 16 kq + 4 g + 0..3] (0 beyond E)
//   pass:    ceil(T2 / 4) bias quads of layer 2 (component r of quad q = b2[16 (4 q + r) + j] in group 0);
//            for T < T1: [T % 4 == 0: u quad — component r = u[16 (T + r) + j] in group 0] then T2 quads
//            W2[16 To + j][16 T + 4 g + 0..3]; ceil(T3 / 4) bias quads of layer 3; for T < T2: T3 quads
//            W3[16 To + j][16 T + 4 g + ..]; padding to a multiple of the ring (lidf_chain16_pass_quads).
// aux: w4 [gf] | b4 [1].
#include "lidf_device.h"

#define C16_RING 8
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define LDQ(rs, voff, soff) \
    __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128((rs), (voff), (soff), 0))

// leaky_relu(0.02) on the four registers of a tile (see lidf_ief16.hip: packed multiply + v_med3_f32)
__device__ __forceinline__ void c16_lrelu4(f32x4& v) {
    const f32x4 t = v * 0.02f;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __builtin_amdgcn_fmed3f(v[i], t[i], __builtin_inff());
}

#define C16_XPRE 7   // layer-1 k-quads whose operands are requested with the gathered rows (E <= 112: all of them)

template <int G, int NT>
__device__ __forceinline__ void c16_fetch(const Chain16Args& a, const long long AN, const long long half0,
                                          f32x4 (&base)[4 * G][NT], f32x4 (&xpre)[C16_XPRE][NT]) {
    constexpr int T1 = 4 * G;
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, g = lane >> 4;
#pragma unroll
    for (int s = 0; s < NT; ++s) {
        const long long r0 = (half0 + s) * 16 + j;
        const long long r = r0 < AN ? r0 : AN - 1;
        const float* vp = a.voxpart ? a.voxpart + (size_t)(a.vox ? a.vox[r] : 0) * (64 * G) + 4 * g : nullptr;
        const float* rp = a.raypart ? a.raypart + (size_t)(a.ray ? a.ray[r] : r) * (64 * G) + 4 * g : nullptr;
        const float* xp = a.X + (size_t)r * a.ldx + 4 * g;
#pragma unroll
        for (int kq = 0; kq < C16_XPRE; ++kq) {
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (kq < a.KQ) {
                const f32x4u v = *(const f32x4u*)(xp + 16 * kq);
                // columns beyond E: their weights are zero, the operands must be too (0 x NaN)
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = 16 * kq + 4 * g + i < a.E ? v[i] : 0.f;
            }
            xpre[kq][s] = x;
        }
#pragma unroll
        for (int T = 0; T < T1; ++T) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (vp) v = *(const f32x4*)(vp + 16 * T);
            if (rp) {
                const f32x4 w = *(const f32x4*)(rp + 16 * T);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] += w[i];
            }
            base[T][s] = v;
        }
    }
}

template <int G, int NT>
__device__ __forceinline__ void c16_tiles(const Chain16Args& a, const long long AN, const long long half0,
                                          const __amdgpu_buffer_rsrc_t srs, const int vq, f32x4 (&ring)[C16_RING],
                                          int& pos, f32x4 (&base)[4 * G][NT], f32x4 (&xpre)[C16_XPRE][NT]) {
    constexpr int T1 = 4 * G, T2 = 2 * G, T3 = G;
    constexpr int NB2 = (T2 + 3) / 4, NB3 = (T3 + 3) / 4;
    constexpr int RAWQ = NB2 + T1 * T2 + T1 / 4 + NB3 + T2 * T3;
    constexpr int PASSQ = (RAWQ + C16_RING - 1) / C16_RING * C16_RING;
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, g = lane >> 4;
    const int l1_bytes = a.KQ * T1 * 1024;
    // ---- layer 1 on the per-row columns: k-quad kq = columns 16 kq + 4 g + {0..3} of the lane's row
    auto l1_step = [&](const f32x4 (&xb)[NT]) {
#pragma unroll
        for (int To = 0; To < T1; To += 2) {
            f32x4 q[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                q[u] = ring[(To + u) % C16_RING];
                ring[(To + u) % C16_RING] = LDQ(srs, vq, pos);   // refill 8 quads ahead: ONE running position
                pos += 1024;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
#pragma unroll
                    for (int s = 0; s < NT; ++s) base[To + u][s] = MFMA16(q[u][r], xb[s][r], base[To + u][s]);
                }
            }
            SCHED_FENCE();
        }
    };
    // (T1 is a multiple of the ring: every k-quad starts on ring slot 0, so the prefetched k-quads unroll with their
    // operands named at compile time)
#pragma unroll
    for (int kq = 0; kq < C16_XPRE; ++kq) {
        if (kq < a.KQ) {
            f32x4 xb[NT];
#pragma unroll
            for (int s = 0; s < NT; ++s) xb[s] = xpre[kq][s];
            l1_step(xb);
        }
    }
    for (int kq = C16_XPRE; kq < a.KQ; ++kq) {   // wider operands: requested here
        f32x4 xb[NT];
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            const long long r0 = (half0 + s) * 16 + j;
            const f32x4u v = *(const f32x4u*)(a.X + (size_t)(r0 < AN ? r0 : AN - 1) * a.ldx + 16 * kq + 4 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i) xb[s][i] = 16 * kq + 4 * g + i < a.E ? v[i] : 0.f;
        }
        l1_step(xb);
    }
    // ---- passes
    float val[NT];
#pragma unroll
    for (int s = 0; s < NT; ++s) val[s] = a.init;
    f32x4 w4v[T3];
#pragma unroll
    for (int T = 0; T < T3; ++T) w4v[T] = *(const f32x4*)(a.aux + 16 * T + 4 * g);
    const float b4 = a.aux[16 * G];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (int pass = 0; pass < a.npass; ++pass) {
        // after this pass the stream continues with the same pass section (another pass) or with layer 1 of the next
        // tile (for a wavefront's last tile the loads are harmless re-reads)
        const bool last = pass + 1 == a.npass;
        const int wrap = last ? 0 : l1_bytes;
        int n = 0;   // quad index inside the pass section (compile-time through the unrolled loops)
#define C16_NEXT(dst)                                                  \
    do {                                                               \
        dst = ring[n % C16_RING];                                      \
        if (n + C16_RING == PASSQ) pos = wrap;                         \
        ring[n % C16_RING] = LDQ(srs, vq, pos);                        \
        pos += 1024;                                                   \
        ++n;                                                           \
    } while (0)
        f32x4 acc2[T2][NT];
#pragma unroll
        for (int qb = 0; qb < NB2; ++qb) {   // the bias quads of layer 2: acc = b2 x 1
            f32x4 bq;
            C16_NEXT(bq);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (4 * qb + r < T2) {
#pragma unroll
                    for (int s = 0; s < NT; ++s) acc2[4 * qb + r][s] = MFMA16(bq[r], 1.f, zero4);
                }
            }
        }
        f32x4 uq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int T = 0; T < T1; ++T) {
            if (T % 4 == 0) C16_NEXT(uq);
            f32x4 h[NT];
#pragma unroll
            for (int s = 0; s < NT; ++s) h[s] = MFMA16(uq[T % 4], val[s], base[T][s]);   // base + u * offset (group 0)
#pragma unroll
            for (int s = 0; s < NT; ++s) c16_lrelu4(h[s]);
#pragma unroll
            for (int To = 0; To < T2; To += 2) {
                f32x4 q[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) C16_NEXT(q[u]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
#pragma unroll
                        for (int s = 0; s < NT; ++s) acc2[To + u][s] = MFMA16(q[u][r], h[s][r], acc2[To + u][s]);
                    }
                }
                SCHED_FENCE();
            }
        }
        f32x4 acc3[T3][NT];
#pragma unroll
        for (int qb = 0; qb < NB3; ++qb) {   // the bias quads of layer 3
            f32x4 bq;
            C16_NEXT(bq);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (4 * qb + r < T3) {
#pragma unroll
                    for (int s = 0; s < NT; ++s) acc3[4 * qb + r][s] = MFMA16(bq[r], 1.f, zero4);
                }
            }
        }
#pragma unroll
        for (int T = 0; T < T2; ++T) {
#pragma unroll
            for (int s = 0; s < NT; ++s) c16_lrelu4(acc2[T][s]);
#pragma unroll
            for (int To = 0; To < T3; To += 2) {
                f32x4 q[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) C16_NEXT(q[u]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
#pragma unroll
                        for (int s = 0; s < NT; ++s) acc3[To + u][s] = MFMA16(q[u][r], acc2[T][s][r], acc3[To + u][s]);
                    }
                }
                SCHED_FENCE();
            }
        }
#pragma unroll
        for (int k = 0; k < PASSQ - RAWQ; ++k) {   // the padding quads: keep the ring in phase
            f32x4 unused;
            C16_NEXT(unused);
            (void)unused;
        }
#undef C16_NEXT
        // layer 4 (gf -> 1): the lane's 4 G features, then the four groups of a row
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            float y = 0.f;
#pragma unroll
            for (int T = 0; T < T3; ++T) {
                c16_lrelu4(acc3[T][s]);
                const f32x4 w = w4v[T];
#pragma unroll
                for (int r = 0; r < 4; ++r) y += w[r] * acc3[T][s][r];
            }
            y += __shfl_xor(y, 16);
            y += __shfl_xor(y, 32);
            val[s] += y + b4;
        }
    }
#pragma unroll
    for (int s = 0; s < NT; ++s) {
        const long long r = (half0 + s) * 16 + j;
        if (r < AN && g == 0) {
            const float y = val[s];
            a.out[r] = a.sigmoid ? 1.f / (1.f + expf(-y)) : fmaxf(fminf(y, y * 0.01f + 0.99f), y * 0.01f);
        }
    }
}

// SLOTS wavefronts per SIMD (wavefronts w and w + 4 of a workgroup share one), NT sub-tiles side by side per
// wavefront (every weight quad feeds 4 NT matrix instructions). A SIMD owns a balanced, contiguous range of 16-row
// sub-tiles, its wavefronts split it.
template <int G, int SLOTS, int NT>
__global__ void __launch_bounds__(256 * SLOTS) lidf_chain16_kernel(Chain16Args a) {
    constexpr int T1 = 4 * G, T2 = 2 * G, T3 = G;
    constexpr int PASSQ = ((T2 + 3) / 4 + T1 * T2 + T1 / 4 + (T3 + 3) / 4 + T2 * T3 + C16_RING - 1) / C16_RING * C16_RING;
    const long long AN = a.n;
    const int lane = threadIdx.x & 63;
    const long long nhalf = (AN + 15) / 16;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long nw = (long long)gridDim.x * 4, wv = (long long)blockIdx.x * 4 + (w & 3);
    const long long per = nhalf / nw, rem = nhalf % nw;
    long long t = wv * per + (wv < rem ? wv : rem);
    long long te = t + per + (wv < rem ? 1 : 0);
    if (SLOTS > 1) {
        // (every wavefront's share but the last a multiple of NT: no odd sub-tile in the middle of the range)
        const long long slot = w >> 2, cnt = te - t;
        const long long chunk = ((cnt + SLOTS - 1) / SLOTS + NT - 1) / NT * NT;
        const long long t0 = t + slot * chunk;
        t = t0 < te ? t0 : te;
        te = t0 + chunk < te ? t0 + chunk : te;
    }
    if (t >= te) return;
    const int total_bytes = (a.KQ * T1 + PASSQ) * 1024;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void*)a.stream, 0, total_bytes, 0x00020000);
    const int vq = lane * 16;
    f32x4 ring[C16_RING];
#pragma unroll
    for (int i = 0; i < C16_RING; ++i) ring[i] = LDQ(srs, vq, i * 1024);
    int pos = C16_RING * 1024;   // byte offset of the next quad to request
    if (NT == 2) {
        f32x4 base[T1][2], xpre[C16_XPRE][2];
        while (te - t >= 2) {
            c16_fetch<G, 2>(a, AN, t, base, xpre);
            c16_tiles<G, 2>(a, AN, t, srs, vq, ring, pos, base, xpre);
            t += 2;
        }
    }
    {
        f32x4 base[T1][1], xpre[C16_XPRE][1];
        for (; t < te; ++t) {
            c16_fetch<G, 1>(a, AN, t, base, xpre);
            c16_tiles<G, 1>(a, AN, t, srs, vq, ring, pos, base, xpre);
        }
    }
}

extern "C" hipError_t lidf_launch_chain16(int gf, const Chain16Args& a, int cus, hipStream_t st) {
    if (a.n <= 0) return hipSuccess;
    if (a.KQ < 1 || a.npass < 1 || !a.X || !a.out) return hipErrorInvalidValue;
    const long long nhalf = (a.n + 15) / 16;
    long long g = (nhalf + 3) / 4;
    if (g > cus) g = cus;
    // gf 32: two wavefronts per SIMD, two sub-tiles each (64 + 32 + 16 accumulator registers per pair of sub-tiles);
    // gf 64: two wavefronts, one sub-tile each (the stage-2 decoder's configuration); gf 128: one wavefront, one
    // sub-tile (128 + 64 + 32 accumulator registers, the 512-register budget of a lone wavefront)
    switch (gf) {
        // (gf 32 at other occupancies — two wavefronts x one sub-tile, four x one, one x two — measured the same or
        // slower, docs/history.md section 13; those instantiations are not shipped)
        case 32: hipLaunchKernelGGL((lidf_chain16_kernel<2, 2, 2>), dim3((unsigned)g), dim3(512), 0, st, a); break;
        case 64: hipLaunchKernelGGL((lidf_chain16_kernel<4, 2, 1>), dim3((unsigned)g), dim3(512), 0, st, a); break;
        case 128: hipLaunchKernelGGL((lidf_chain16_kernel<8, 1, 1>), dim3((unsigned)g), dim3(256), 0, st, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
