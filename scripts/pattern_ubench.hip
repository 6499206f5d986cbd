// Development microbenchmark: the layer-2 segment pattern of lidf_points_h_kernel in isolation —
// per (hi, lo) quad pair: 3 dependent-chain matrix instructions on one of 4 accumulators + the
// activation/split of one pair of values of the *next* operand tile (8 VALU) — with constant A
// operands (no LDS), to see what the instruction mix alone costs. Variants place the 8 VALU
// (a) after the two hi-quad instructions (as the kernel does), (b) spread 3/3/2 behind each
// instruction, (c) none. 1 and 2 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMAH(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, (a)), __builtin_bit_cast(h8, (b)), (c), 0, 0, 0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ float lrelu1(const float x) {
    const float t = x * 0.02f; float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(t));
    return r;
}
__device__ __forceinline__ void split2(const float x0, const float x1, float& hi, float& lo) {
    h2 hh; hh[0] = (_Float16)x0; hh[1] = (_Float16)x1;
    const float hw = __builtin_bit_cast(float, hh);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hw), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hw), "v"(x1));
    h2 ll; ll[0] = (_Float16)r0; ll[1] = (_Float16)r1;
    hi = hw; lo = __builtin_bit_cast(float, ll);
}

// MODE 3: as MODE 1 with the A operands read from LDS through a 4-deep ring (one ds_read_b128 per
// quad); MODE 4: additionally, every 16 quads, 4 global loads -> 4 ds_write_b128 -> s_barrier.
template <int MODE>
__global__ void __launch_bounds__(256, 2) k(const float* src, float* out, long long* cyc, int iters) {
    __shared__ f32x4 sb[3 * 1024];
    const int tid = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 3 * 1024; i += 256) sb[i] = f32x4{src[i & 0xffff], src[(i + 7) & 0xffff] * 1e-3f, src[(i + 3) & 0xffff], 0.5f};
    __syncthreads();
    f32x4 ring[4], stage[4];
    for (int i = 0; i < 4; ++i) { ring[i] = sb[i * 64 + lane]; stage[i] = ring[i]; }
    const f32x4* gsrc = (const f32x4*)src;
    int q = 0;
    f32x4 Ah, Al;
    for (int i = 0; i < 4; ++i) { Ah[i] = src[(tid * 8 + i) & 0xffff]; Al[i] = src[(tid * 8 + 4 + i) & 0xffff] * 1e-3f; }
    f32x16 acc[4], pre;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int j = 0; j < 16; ++j) pre[j] = src[(tid + j * 977) & 0xffff];
    f32x4 bh[2][2], bl[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) { bh[a][b] = Ah; bl[a][b] = Al; }
    long long t0 = clock64();
    for (int it2 = 0; it2 < iters; it2 += 2) {
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const int it = it2 + par;
#pragma unroll
        for (int j = 0; j < 8; ++j) {            // 8 (hi, lo) pairs = one H1 tile against 4 output tiles
            const int t = j & 3, sub = j >> 2;
            if (MODE >= 3) {
                // two quads (hi, lo) per pair: take them from the ring, refill 4 quads ahead
                Ah = ring[(2 * j) & 3];
                ring[(2 * j) & 3] = sb[(((q + 4) & 15) * 64 + lane) + 1024 * (((q + 4) >> 4) % 3)];
                Al = ring[(2 * j + 1) & 3];
                ring[(2 * j + 1) & 3] = sb[(((q + 5) & 15) * 64 + lane) + 1024 * (((q + 5) >> 4) % 3)];
                if ((MODE == 4 || MODE == 6) && (q & 15) == 4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) sb[1024 * (((q >> 4) + 1) % 3) + (4 * wave + i) * 64 + lane] = stage[i];
#pragma unroll
                    for (int i = 0; i < 4; ++i) stage[i] = gsrc[((q * 4 + i) * 64 + lane + wave * 256) & 0x3fff];
                }
                if (MODE == 9) {
                    // direct global -> LDS copies (no staging registers, no ds_write): issued after the
                    // barrier of this chunk into the buffer read two chunks later
                    const int qq = q & 15;
                    if (qq == 6) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(2)\n\ts_barrier" ::: "memory");
                    if (qq == 8) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            __builtin_amdgcn_global_load_lds(
                                (const void*)(gsrc + (((q * 4 + i) * 64 + wave * 256) & 0x3fff) + lane),
                                (__attribute__((address_space(3))) void*)(sb + 1024 * (((q >> 4) + 2) % 3) + (4 * wave + i) * 64),
                                16, 0, 0);
                    }
                }
                if (MODE == 8) {
                    // spread: one quad of the staged chunk per (hi, lo) pair
                    const int qq = q & 15;
                    if (qq == 0 || qq == 2 || qq == 4 || qq == 6) {
                        const int i = qq >> 1;
                        sb[1024 * (((q >> 4) + 1) % 3) + (4 * wave + i) * 64 + lane] = stage[i];
                        stage[i] = gsrc[((q * 4 + i) * 64 + lane + wave * 256) & 0x3fff];
                    }
                    if (qq == 8) asm volatile("s_waitcnt lgkmcnt(2)\n\ts_barrier" ::: "memory");
                }
                if ((MODE == 4 || MODE == 5) && (q & 15) == 6) asm volatile("s_waitcnt lgkmcnt(2)\n\ts_barrier" ::: "memory");
                if (MODE == 7 && (q & 15) == 6) asm volatile("s_barrier" ::: "memory");
                q += 2;
            }
            acc[t] = MFMAH(Ah, bh[par][sub], acc[t]);
            if (MODE == 1 || MODE >= 3) { pre[2 * j] = lrelu1(pre[2 * j]); pre[2 * j + 1] = lrelu1(pre[2 * j + 1]); }
            acc[t] = MFMAH(Ah, bl[par][sub], acc[t]);
            if (MODE == 0) {
                float hi, lo;
                split2(lrelu1(pre[2 * j]), lrelu1(pre[2 * j + 1]), hi, lo);
                bh[par ^ 1][j >> 2][j & 3] = hi; bl[par ^ 1][j >> 2][j & 3] = lo;
            }
            FENCE();
            acc[t] = MFMAH(Al, bh[par][sub], acc[t]);
            if (MODE == 1 || MODE >= 3) {
                float hi, lo;
                split2(pre[2 * j], pre[2 * j + 1], hi, lo);
                bh[par ^ 1][j >> 2][j & 3] = hi; bl[par ^ 1][j >> 2][j & 3] = lo;
            }
            FENCE();
        }
        // the next tile's pre-activations come from an accumulator (as from the layer-1 chain)
#pragma unroll
        for (int j = 0; j < 16; ++j) pre[j] = acc[par][j] * 1e-3f + 0.1f;
      }
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) s += bh[a][b][0] + bl[a][b][1];
    s += ring[0][0] + stage[1][1];
    out[tid] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, const float* src, float* out, long long* cyc) {
    for (int wgs = 256; wgs <= 512; wgs += 256) {
        const int iters = 4000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<MODE>), dim3(wgs), dim3(256), 0, 0, src, out, cyc, 200);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE>), dim3(wgs), dim3(256), 0, 0, src, out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double nm = (double)iters * 24;
        printf("%-34s %d wavefront(s)/SIMD: %.1f ticks per matrix instruction per wavefront, %.0f TFLOP/s\n", name, wgs / 256,
               c / nm, (double)wgs * 4 * nm * 32768.0 / ms / 1e9);
    }
}
int main() {
    float* h = (float*)malloc(65536 * 4);
    for (int i = 0; i < 65536; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
    float *src, *out; long long* cyc;
    hipMalloc(&src, 65536 * 4); hipMalloc(&out, 512 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    hipMemcpy(src, h, 65536 * 4, hipMemcpyHostToDevice);
    run<2>("no VALU", src, out, cyc);
    run<0>("8 VALU behind the hi-quad pair", src, out, cyc);
    run<1>("VALU spread 2 / 2 / 4", src, out, cyc);
    run<3>("+ A operands through an LDS ring", src, out, cyc);
    run<4>("+ staging loads, ds_write, barrier", src, out, cyc);
    run<5>("LDS ring + waitcnt/barrier only", src, out, cyc);
    run<7>("LDS ring + bare s_barrier only", src, out, cyc);
    run<6>("LDS ring + loads/ds_write only", src, out, cyc);
    run<8>("staging spread over 4 pairs + barrier", src, out, cyc);
    run<9>("direct global->LDS copies + barrier", src, out, cyc);
    return 0;
}
