// lidf_dgrad.hip — register-chained input gradient of the decoders' layers 3 and 2 (gfx950 / CDNA4).
//
// Backward of models/implicit_net.py:84-90 (IMNet) / :142-146 (IEF), per pass:
//   dZ2 = (dZ3 W3) * lrelu'(Z2)        [n,64]  -> [n,128]
//   dZ1 = (dZ2 W2) * lrelu'(Z1)        [n,128] -> [n,256]
// with lrelu'(Z) read off the sign words the training forward keeps beside the activations (one bit
// per value, H > 0; H and Z have the same sign: 48 bytes per row instead of the 1.5 KB of H2 | H1).
// Layer by layer this was two launches of the generic linear kernel, dZ2 written and read back;
// here a wavefront keeps its 32 rows in the accumulator layout of the forward chain
// (lidf_points.hip): the 32 x 32 output tile of a layer IS the B operand of the next layer's matrix
// instructions (register r of tile T, half h = feature 32T + (r&3) + 8(r>>2) + 4h), so dZ2 goes
// from accumulators to operands without leaving the registers; it is stored once (the weight
// gradient of layer 2 needs it), as is dZ1.
//
// Stream (lidf_pack_dgrad_kernel): A fragments of W3^T then W2^T, 1 KiB quads consumed strictly in
// order through an 8-deep register ring (never drained; wraps to the start for the next tile):
//   layer A (K = 64, 4 output tiles): quad = 16 pair + 2 kq + t      (kq < 8,  tile 2 pair + t)
//   layer B (K = 128, 8 output tiles): quad = 32 + 32 pair + 2 kq + t (kq < 16)
// Two output tiles advance together (their accumulate chains interleave), a tile pair is finished —
// masked, stored — while the next pair multiplies. Per 32 rows: 640 matrix instructions, 1.8 KB of
// HBM traffic per row (dZ3 and the sign words in; dZ2, dZ1 out; ACC: + 1 KB, the running sum).
#include "lidf_device.h"

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define LDQ(rs, voff, soff) \
    __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128((rs), (voff), (soff), 0))

#define DG_A_QUADS 32
#define DG_B_QUADS 128
#define DG_QUADS (DG_A_QUADS + DG_B_QUADS)

struct DgradArgs {
    const float* stream;   // DG_QUADS KiB
    const float* dz3;      // [n,64]
    const unsigned* m2;    // [n,4]  sign words of H2 (the training forward's mask_tile, lidf_points.hip)
    const unsigned* m1;    // [n,8]  ... of H1
    float* dz2;            // [n,128]
    float* dz1;            // [n,256]
    long long n;
    float slope;
};

__device__ __forceinline__ int dg_feature(int s, int half) {
    // k-step s of a layer whose operand is the previous layer's accumulator registers
    const int T = s >> 4, r = s & 15;
    return 32 * T + (r & 3) + 8 * (r >> 2) + 4 * half;
}

__global__ void lidf_pack_dgrad_kernel(const float* __restrict__ w3, const float* __restrict__ w2,
                                       float* __restrict__ stream) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= DG_QUADS * 256) return;
    int quad = e / 256;
    const int lane = (e % 256) / 4, jj = e & 3;
    const int half = lane >> 5, c32 = lane & 31;
    float v;
    if (quad < DG_A_QUADS) {
        const int pair = quad / 16, kq = (quad % 16) / 2, t = quad % 2;
        const int k = dg_feature(4 * kq + jj, half);           // output of layer 3 (0..63)
        v = w3[(size_t)k * LIDF_H2 + 32 * (2 * pair + t) + c32];
    } else {
        quad -= DG_A_QUADS;
        const int pair = quad / 32, kq = (quad % 32) / 2, t = quad % 2;
        const int k = dg_feature(4 * kq + jj, half);           // output of layer 2 (0..127)
        v = w2[(size_t)k * LIDF_H1 + 32 * (2 * pair + t) + c32];
    }
    stream[e] = v;
}

__device__ __forceinline__ void dg_load_tile(const float* row, int T, f32x4 (&m)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) m[g] = *(const f32x4*)(row + 32 * T + 8 * g);
}
// lrelu' of value e of tile T from the lane's sign words: bit 31 - (16 (T & 1) + e) of word T / 2
// (T and e are constants once the step loop is unrolled)
__device__ __forceinline__ float dg_factor(const unsigned* mk, int T, int e, float slope) {
    return (mk[T / 2] & (1u << (31 - (16 * (T & 1) + e)))) ? 1.f : slope;
}
// acc *= lrelu', stored to row + 32 T
__device__ __forceinline__ void dg_finish_tile(f32x16& acc, const unsigned* mk, float slope,
                                               float* row, int T) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[4 * g + i] *= dg_factor(mk, T, 4 * g + i, slope);
            o[i] = acc[4 * g + i];
        }
        *(f32x4*)(row + 32 * T + 8 * g) = o;
    }
}

// ACC: dZ1 is added to what dz1 holds instead of stored (the running sum over the IEF's passes,
// lidf_api.hip: the last pass processed adds its dZ1 straight into the sum).
template <bool ACC>
__global__ void __launch_bounds__(256, 2) lidf_dgrad_chain_kernel(DgradArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int h = lane >> 5;
    const int col = lane & 31;
    const __amdgpu_buffer_rsrc_t srs =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.stream, 0, DG_QUADS * 1024, 0x00020000);
    const int vq = lane * 16;
    const long long ntile = (a.n + 127) / 128;
    const long long per = ntile / gridDim.x, rem = ntile % gridDim.x;
    const long long bx = blockIdx.x;
    const long long tb = bx * per + (bx < rem ? bx : rem);
    const long long te = tb + per + (bx < rem ? 1 : 0);
    if (tb >= te) return;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    f32x4 ring[LIDF_RING];
#pragma unroll
    for (int i = 0; i < LIDF_RING; ++i) ring[i] = LDQ(srs, vq, i * 1024);

    // operand of layer A (dZ3, 2 tiles) and the lane's sign words of H2 / H1 (24 bytes) of a tile: one
    // burst, requested a whole tile ahead (the next tile's burst goes out as soon as this tile's
    // registers are free to take it, so its HBM latency passes under this tile's matrix work)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    f32x4 Bn[2][4];
    u32x2 M2n;
    u32x4 M1n;
    auto request = [&](long long tile) {
        long long p = tile * 128 + wave * 32 + col;
        if (p >= a.n) p = a.n - 1;
        const float* z3 = a.dz3 + (size_t)p * LIDF_H3 + 4 * h;
        M2n = *(const u32x2*)(a.m2 + (size_t)p * 4 + 2 * h);
        M1n = *(const u32x4*)(a.m1 + (size_t)p * 8 + 4 * h);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int g = 0; g < 4; ++g) Bn[t][g] = *(const f32x4*)(z3 + 32 * t + 8 * g);
        }
    };
    request(tb);
    for (long long tile = tb; tile < te; ++tile) {
        if (tile * 128 + wave * 32 >= a.n) break;   // wave-uniform
        const long long p = tile * 128 + wave * 32 + col;
        // rows beyond n repeat row n-1: same operands, same values stored twice (ACC: not added)
        const long long pc = p < a.n ? p : a.n - 1;
        const bool valid = p < a.n;
        float* z2 = a.dz2 + (size_t)pc * LIDF_H2 + 4 * h;
        float* z1 = a.dz1 + (size_t)pc * LIDF_H1 + 4 * h;

        f32x16 B3[2];
        unsigned MK2[2], MK1[4];
        MK2[0] = M2n[0]; MK2[1] = M2n[1];
        MK1[0] = M1n[0]; MK1[1] = M1n[1]; MK1[2] = M1n[2]; MK1[3] = M1n[3];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int i = 0; i < 4; ++i) B3[t][4 * g + i] = Bn[t][g][i];
            }
        }
        SCHED_FENCE();
        if (tile + 1 < te) request(tile + 1);
        SCHED_FENCE();

        f32x16 Z2[4];
        f32x16 acc[2];
        f32x4 SB[ACC ? 2 : 1][4];   // ACC: the pair's tiles of the running sum, requested at the pair's start
#pragma unroll
        for (int s = 0; s < DG_QUADS; ++s) {
            const f32x4 aq = ring[s % LIDF_RING];
            {
                const int nx = s + LIDF_RING;
                const int rel = nx < DG_QUADS ? nx : nx - DG_QUADS;   // wraps: same stream next tile
                ring[s % LIDF_RING] = LDQ(srs, vq + (rel & 3) * 1024, (rel >> 2) * 4096);
            }
            if (s < DG_A_QUADS) {
                const int pair = s / 16, kq = (s % 16) / 2, t = s % 2;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int k = 4 * kq + jj;
                    acc[t] = MFMA(aq[jj], B3[k / 16][k % 16], k == 0 ? zero16 : acc[t]);
                }
                if (kq == 7) {
                    dg_finish_tile(acc[t], MK2, a.slope, z2, 2 * pair + t);
                    Z2[2 * pair + t] = acc[t];
                }
            } else {
                const int q = s - DG_A_QUADS;
                const int pair = q / 32, kq = (q % 32) / 2, t = q % 2;
                if (ACC && kq == 0) dg_load_tile(z1, 2 * pair + t, SB[ACC ? t : 0]);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int k = 4 * kq + jj;
                    acc[t] = MFMA(aq[jj], Z2[k / 16][k % 16], k == 0 ? zero16 : acc[t]);
                }
                if (kq == 15) {
                    if (ACC) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            f32x4 o;
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                o[i] = SB[ACC ? t : 0][g][i] +
                                       acc[t][4 * g + i] * dg_factor(MK1, 2 * pair + t, 4 * g + i, a.slope);
                            if (valid) *(f32x4*)(z1 + 32 * (2 * pair + t) + 8 * g) = o;
                        }
                    } else {
                        dg_finish_tile(acc[t], MK1, a.slope, z1, 2 * pair + t);
                    }
                }
            }
            SCHED_FENCE();
        }
    }
}

extern "C" hipError_t lidf_launch_dgrad_chain(const float* w3, const float* w2, const float* dz3,
                                              const unsigned* m2, const unsigned* m1, long long n,
                                              float slope, float* dz2, float* dz1, int accumulate,
                                              float* stream, int cus, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    // (w3 == NULL: `stream` was packed earlier by lidf_launch_pack_dgrad — once per backward call, not per pass)
    if (w3) hipLaunchKernelGGL(lidf_pack_dgrad_kernel, dim3(DG_QUADS), dim3(256), 0, st, w3, w2, stream);
    DgradArgs a;
    a.stream = stream; a.dz3 = dz3; a.m2 = m2; a.m1 = m1; a.dz2 = dz2; a.dz1 = dz1; a.n = n;
    a.slope = slope;
    const long long ntile = (n + 127) / 128;
    const long long g = ntile < 2LL * cus ? ntile : 2LL * cus;
    if (accumulate)
        hipLaunchKernelGGL(lidf_dgrad_chain_kernel<true>, dim3((unsigned)g), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL(lidf_dgrad_chain_kernel<false>, dim3((unsigned)g), dim3(256), 0, st, a);
    return hipGetLastError();
}

extern "C" size_t lidf_dgrad_stream_bytes(void) { return (size_t)DG_QUADS * 1024; }
extern "C" hipError_t lidf_launch_pack_dgrad(const float* w3, const float* w2, float* stream, hipStream_t st) {
    hipLaunchKernelGGL(lidf_pack_dgrad_kernel, dim3(DG_QUADS), dim3(256), 0, st, w3, w2, stream);
    return hipGetLastError();
}
