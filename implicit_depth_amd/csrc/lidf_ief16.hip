// lidf_ief16.hip — the stage-2 decoder (RefineNet.get_pred_refine's offset_dec, models/pipeline.py:1027:
// an IEF / IMNet on D = 334 columns) on v_mfma_f32_16x16x4_f32, round 4.
//
// Why another matrix shape. A frame has 76,800 rays = 2,400 wave-tiles of the 32 x 32 transposed-chain
// kernel (lidf_points.hip) over 1,024 SIMDs: 2.34 rounds of work in 3 rounds of time, and a 32 x 32 tile is
// indivisible (rounds 2-3: 0.60 of the f32 matrix peak, 0.78 of it lost to the partial round). The
// 16 x 16 x 4 instruction has the same rate (64 FLOP per cycle and SIMD) on a tile of 16 points, so a
// wavefront walks 16-ray sub-tiles — TWO side by side while it has two left (every weight quad feeds both:
// the weight traffic per FLOP of the 32 x 32 kernel), ONE at the end: 4,800 sub-tiles over 1,024 wavefronts
// = 4.69 -> at most 5 half-rounds = 2.5 rounds.
//
// Transposed chain as before: D[out feature][point] += A[out][k] B[k][point]. Lane l = (point j = l & 15,
// group g = l >> 4). An accumulator tile (16 features x 16 points) is 4 registers: register r of lane l
// holds feature 16T + 4g + r of point j — which is exactly the B operand of the next layer's k-step r of
// input tile T (B[k = g][j]), so activations never leave the register file, and the matching A operand of
// the four k-steps r = 0..3 is W[out][16T + 4g + 0..3]: four consecutive floats of nn.Linear's row.
//
// Stream (lidf_pack: LIDF_MODE_IEF16), 1 KiB quads = float4 per lane, consumed in order through the ring:
//   layer 1 (embed(pos) columns; the voxel / ROI / direction columns arrive as gathered per-voxel and
//            per-ray rows):  for kq < KQ = ceil(E / 16), To < 16 : quad kq * 16 + To,
//            lane l: W1[16 To + j][c0 + 16 kq + 4 g + 0..3]  (0 beyond E)
//   pass (k-major, 168 quads):  2 bias quads of layer 2 (component r of quad q = b2[16 (4 q + r) + j] in group 0,
//            0 elsewhere: the accumulators start as bias x 1);  for T < 16 : [T % 4 == 0: u quad — component r =
//            u[16 (T + r) + j] in group 0: the IEF's rank-1 term u * offset, implicit_net.py:135-139] then 8 quads
//            W2[16 To + j][16 T + 4 g + 0..3];   1 bias quad of layer 3;   for T < 8 : 4 quads
//            W3[16 To + j][16 T + 4g + ..];   1 padding quad.
//   (biases as matrix steps, not loads: vector memory returns in order, a load of a bias vector at the head of
//   a pass waits behind the ring's requests and the ring behind it — measured 10 % of the launch)
// aux: w4 [64] | b4 [1] (read once per wavefront).
#include "lidf_device.h"

// (a ring of 16 quads — a lone sub-tile consumes a quad in 128 cycles — measured no faster than 8)
#define IEF16_RING 8
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define LDQ(rs, voff, soff) \
    __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128((rs), (voff), (soff), 0))

// leaky_relu(0.02) on the four registers of a tile: packed multiplies + one v_med3_f32 per register —
// max(x, 0.02 x) = med3(x, 0.02 x, +inf). fmaxf would add a canonicalising v_max per input under the IEEE
// mode (on gfx950 every vector instruction of the wavefront delays its f32 matrix instructions); inline
// assembly instead hides the operands from the compiler's matrix-result hazard tracking (measured: wrong
// values).
__device__ __forceinline__ void lrelu4(f32x4& v) {
    const f32x4 t = v * 0.02f;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __builtin_amdgcn_fmed3f(v[i], t[i], __builtin_inff());
}

#define IEF16_XPRE 4   // layer-1 k-quads whose operands are requested a tile ahead (E = 51: all four)

// the gathered layer-1 terms (per-voxel row + per-ray row) and the first layer-1 operands of the sub-tiles
// half0, half0 + 1 — requested while the previous tile's last pass runs its layer 3 (base is dead by then)
__device__ __forceinline__ void ief16_fetch(const Ief16Args& a, const long long AN, const long long half0,
                                            f32x4 (&base)[16][2], f32x4 (&xpre)[IEF16_XPRE][2]) {
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, g = lane >> 4;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const long long r0 = (half0 + s) * 16 + j;
        const long long r = r0 < AN ? r0 : AN - 1;
        const float* vp = a.voxpart + (size_t)a.vox[r] * 256 + 4 * g;
        const float* rp = a.raypart + (size_t)r * 256 + 4 * g;
        const float* xp = a.X + (size_t)r * a.ldx + 4 * g;
#pragma unroll
        for (int kq = 0; kq < IEF16_XPRE; ++kq) {
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (kq < a.KQ) {
                const f32x4u v = *(const f32x4u*)(xp + 16 * kq);
                // columns beyond E belong to other parts of the row (or were never written): their weights
                // are zero, the operands must be too (0 x NaN)
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = 16 * kq + 4 * g + i < a.E ? v[i] : 0.f;
            }
            xpre[kq][s] = x;
        }
#pragma unroll
        for (int T = 0; T < 16; ++T) {
            const f32x4 v = *(const f32x4*)(vp + 16 * T);
            const f32x4 w = *(const f32x4*)(rp + 16 * T);
#pragma unroll
            for (int i = 0; i < 4; ++i) base[T][s][i] = v[i] + w[i];
        }
    }
}

// NT sub-tiles starting at half0; base / xpre arrive fetched; `next` >= 0: fetch the sub-tiles next, next + 1
// during the last pass
template <int NT>
__device__ __forceinline__ void ief16_tiles(const Ief16Args& a, const long long AN, const long long half0,
                                            const __amdgpu_buffer_rsrc_t srs, const int vq, f32x4 (&ring)[IEF16_RING],
                                            int& pos, f32x4 (&base)[16][2], f32x4 (&xpre)[IEF16_XPRE][2], const long long next) {
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, g = lane >> 4;
    const int l1_bytes = a.KQ * 16 * 1024;
    // ---- layer 1 on the embed(pos) columns: k-quad kq = columns 16 kq + 4 g + {0..3} of the lane's row
    for (int kq = 0; kq < a.KQ; ++kq) {
        f32x4 xb[NT];
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            if (kq < IEF16_XPRE) {
                xb[s] = kq == 0 ? xpre[0][s] : (kq == 1 ? xpre[1][s] : (kq == 2 ? xpre[2][s] : xpre[3][s]));
            } else {   // wider embeddings (multires > 10): requested here
                const long long r0 = (half0 + s) * 16 + j;
                const f32x4u v = *(const f32x4u*)(a.X + (size_t)(r0 < AN ? r0 : AN - 1) * a.ldx + 16 * kq + 4 * g);
#pragma unroll
                for (int i = 0; i < 4; ++i) xb[s][i] = 16 * kq + 4 * g + i < a.E ? v[i] : 0.f;
            }
        }
#pragma unroll
        for (int To = 0; To < 16; To += 2) {
            // two output tiles per step: four (two) independent accumulators between two matrix instructions
            // on the same one — a 16 x 16 x 4 instruction issues in 32 cycles, its result takes longer
            f32x4 q[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                q[u] = ring[(To + u) % IEF16_RING];
                // refill 8 quads ahead: the stream is consumed strictly in order (layer 1, then the pass
                // section behind it), so the position of the next request is ONE running scalar
                ring[(To + u) % IEF16_RING] = LDQ(srs, vq, pos);
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
    }
    // ---- passes
    float val[NT];
#pragma unroll
    for (int s = 0; s < NT; ++s) val[s] = a.init;
    // layer 4's weights and bias: once per tile, before the matrix stream starts
    f32x4 w4v[4];
#pragma unroll
    for (int T = 0; T < 4; ++T) w4v[T] = *(const f32x4*)(a.aux + 16 * T + 4 * g);
    const float b4 = a.aux[64];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (int pass = 0; pass < a.npass; ++pass) {
        // after this pass the stream continues with the same pass section (another pass) or with layer 1 of
        // the next tile (for a wavefront's last tile the loads are harmless re-reads)
        const bool last = pass + 1 == a.npass;
        (void)next;
        const int wrap = last ? 0 : l1_bytes;
        int n = 0;   // quad index inside the pass section (compile-time through the unrolled loops)
        f32x4 acc2[8][NT];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {   // the two bias quads of layer 2: acc = b2 x 1
            const f32x4 bq = ring[n % IEF16_RING];
            if (n + IEF16_RING == IEF16_PASS_QUADS) pos = wrap;   // the section wraps (another pass / the next tile)
            ring[n % IEF16_RING] = LDQ(srs, vq, pos);
            pos += 1024;
            ++n;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int s = 0; s < NT; ++s) acc2[4 * qb + r][s] = MFMA16(bq[r], 1.f, zero4);
            }
        }
        f32x4 uq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int T = 0; T < 16; ++T) {
            if (T % 4 == 0) {
                uq = ring[n % IEF16_RING];
                if (n + IEF16_RING == IEF16_PASS_QUADS) pos = wrap;   // the section wraps (another pass / the next tile)
                ring[n % IEF16_RING] = LDQ(srs, vq, pos);
                pos += 1024;
                ++n;
            }
            f32x4 h[NT];
#pragma unroll
            for (int s = 0; s < NT; ++s) {
                h[s] = MFMA16(uq[T % 4], val[s], base[T][s]);   // base + u * offset (u lives in group 0 only)
            }
#pragma unroll
            for (int s = 0; s < NT; ++s) lrelu4(h[s]);
#pragma unroll
            for (int To = 0; To < 8; To += 2) {
                f32x4 q[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    q[u] = ring[n % IEF16_RING];
                    if (n + IEF16_RING == IEF16_PASS_QUADS) pos = wrap;   // the section wraps (another pass / the next tile)
                    ring[n % IEF16_RING] = LDQ(srs, vq, pos);
                    pos += 1024;
                    ++n;
                }
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
        // (requesting the next tile's gathered rows here — base is dead in the last pass — was measured slower,
        // 157 -> 167 us: vector memory returns in order, so ~70 gather requests in front of the weight ring's
        // loads stall the ring for their whole latency, in the middle of the matrix stream)
        f32x4 acc3[4][NT];
        {   // the bias quad of layer 3
            const f32x4 bq = ring[n % IEF16_RING];
            if (n + IEF16_RING == IEF16_PASS_QUADS) pos = wrap;   // the section wraps (another pass / the next tile)
            ring[n % IEF16_RING] = LDQ(srs, vq, pos);
            pos += 1024;
            ++n;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int s = 0; s < NT; ++s) acc3[r][s] = MFMA16(bq[r], 1.f, zero4);
            }
        }
#pragma unroll
        for (int T = 0; T < 8; ++T) {
#pragma unroll
            for (int s = 0; s < NT; ++s) lrelu4(acc2[T][s]);
#pragma unroll
            for (int To = 0; To < 4; To += 2) {
                f32x4 q[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    q[u] = ring[n % IEF16_RING];
                    if (n + IEF16_RING == IEF16_PASS_QUADS) pos = wrap;   // the section wraps (another pass / the next tile)
                    ring[n % IEF16_RING] = LDQ(srs, vq, pos);
                    pos += 1024;
                    ++n;
                }
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
        for (; n < IEF16_PASS_QUADS; ++n) {   // the padding quads: keep the ring in phase
            if (n + IEF16_RING == IEF16_PASS_QUADS) pos = wrap;   // the section wraps (another pass / the next tile)
            ring[n % IEF16_RING] = LDQ(srs, vq, pos);
            pos += 1024;
        }
        // layer 4 (64 -> 1): the lane's 16 features, then the four groups of a point
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            float y = 0.f;
#pragma unroll
            for (int T = 0; T < 4; ++T) {
                lrelu4(acc3[T][s]);
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

// SLOTS = wavefronts per SIMD (wavefronts w and w + 4 of a workgroup share one). A SIMD owns a balanced,
// contiguous range of 16-ray sub-tiles (4 or 5 of the 4,800 of a 240x320 frame).
//  SLOTS = 1: one wavefront walks it two sub-tiles at a time (every weight quad feeds 8 matrix instructions),
//             then the odd one; 256 + 78 registers: one wavefront per SIMD is all that fits.
//  SLOTS = 2: two wavefronts split it ceil / floor and walk one sub-tile at a time (half the accumulators: both
//             fit the register file); the gaps of one instruction stream are filled by the other, at twice
//             the weight-stream traffic per sub-tile.
template <int SLOTS>
__global__ void __launch_bounds__(256 * SLOTS) lidf_ief16_kernel(Ief16Args a) {
    const long long AN = a.n_dev ? (long long)*a.n_dev : a.n;
    const int lane = threadIdx.x & 63;
    const long long nhalf = (AN + 15) / 16;
    // (the wavefront index through readfirstlane: the compiler then knows the sub-tile range, the loop trip
    // counts and the running stream position to be wave-uniform — scalar registers, scalar adds)
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long nw = (long long)gridDim.x * 4, wv = (long long)blockIdx.x * 4 + (w & 3);
    // contiguous, balanced ranges of 16-ray sub-tiles per SIMD
    const long long per = nhalf / nw, rem = nhalf % nw;
    long long t = wv * per + (wv < rem ? wv : rem);
    long long te = t + per + (wv < rem ? 1 : 0);
    if (SLOTS == 2) {
        const long long mid = t + (te - t + 1) / 2;
        if (w >> 2) t = mid; else te = mid;
    }
    if (t >= te) return;
    const int total_bytes = (a.KQ * 16 + IEF16_PASS_QUADS) * 1024;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void*)a.stream, 0, total_bytes, 0x00020000);
    const int vq = lane * 16;
    f32x4 ring[IEF16_RING];
#pragma unroll
    for (int i = 0; i < IEF16_RING; ++i) ring[i] = LDQ(srs, vq, i * 1024);
    int pos = IEF16_RING * 1024;   // byte offset of the next quad to request
    if (SLOTS == 1) {
        f32x4 base[16][2], xpre[IEF16_XPRE][2];
        while (te - t >= 2) {
            ief16_fetch(a, AN, t, base, xpre);
            ief16_tiles<2>(a, AN, t, srs, vq, ring, pos, base, xpre, -1);
            t += 2;
        }
        if (t < te) {
            ief16_fetch(a, AN, t, base, xpre);
            ief16_tiles<1>(a, AN, t, srs, vq, ring, pos, base, xpre, -1);
        }
    } else {
        f32x4 base[16][2], xpre[IEF16_XPRE][2];
        for (; t < te; ++t) {
            ief16_fetch(a, AN, t, base, xpre);
            ief16_tiles<1>(a, AN, t, srs, vq, ring, pos, base, xpre, -1);
        }
    }
}

extern "C" hipError_t lidf_launch_ief16(const Ief16Args& a, int cus, hipStream_t st) {
    if (a.n <= 0) return hipSuccess;
    if (a.KQ < 1 || a.npass < 1) return hipErrorInvalidValue;
    const long long nhalf = (a.n + 15) / 16;
    long long g = (nhalf + 3) / 4;
    if (g > cus) g = cus;
    static int slots = -1;   // development knob: 1 = one wavefront per SIMD (the round-4 first version)
    if (slots < 0) { const char* e = getenv("LIDF_IEF16_SLOTS"); slots = (e && atoi(e) == 1) ? 1 : 2; }
    if (slots == 2)
        hipLaunchKernelGGL(lidf_ief16_kernel<2>, dim3((unsigned)g), dim3(512), 0, st, a);
    else
        hipLaunchKernelGGL(lidf_ief16_kernel<1>, dim3((unsigned)g), dim3(256), 0, st, a);
    return hipGetLastError();
}
