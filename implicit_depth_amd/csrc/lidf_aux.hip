// lidf_aux.hip — the small kernels around the per-point decoder kernel (gfx950 / CDNA4):
// positional encoding, per-ray ROIAlign + direction embedding, per-ray softmax/argmax/select,
// ray generation, ray/voxel and point/voxel box tests, exclusive scan.
// Built with -ffp-contract=off: these restate f32 arithmetic of the reference op by op.
#include "lidf_device.h"

// ------------------------------------------------------------------------------------------------
// Positional encoding — Embedder.embed (models/implicit_net.py:38-39).
// out[i, :] = [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)], each block 3 wide.
// HBM-bound (12 B in, 4*(3+6L) B out per row): one workgroup builds a 64-row output tile in LDS —
// thread (row, coordinate) forms x/(2 pi) = hi + lo once, then per octave an exact 2^o scaling and
// v_fract feed v_sin_f32 / v_cos_f32 (revolutions in; |err| <= 4.2e-7 at every octave, measured
// against double precision) — and the tile, which is a contiguous 16-byte aligned slab of the
// output, leaves as coalesced float4 stores.
// ------------------------------------------------------------------------------------------------
#define EMBED_ROWS 64
__global__ void __launch_bounds__(256) lidf_embed_kernel(const float* __restrict__ x, long long n,
                                                         int L, float* __restrict__ out) {
    extern __shared__ float tile[];  // [EMBED_ROWS][E]
    const int E = 3 + 6 * L;
    const long long row0 = (long long)blockIdx.x * EMBED_ROWS;
    const int nrows = (int)min((long long)EMBED_ROWS, n - row0);
    const int t = threadIdx.x;
    if (t < 3 * EMBED_ROWS) {
        const int r = t / 3, c = t - 3 * r;
        if (r < nrows) {
            const float v = x[3 * (row0 + r) + c];
            float* w = tile + r * E;
            w[c] = v;
            const float C_HI = 0.15915493667125702f, C_LO = 6.4206383e-09f;  // 1/(2 pi) = hi + lo
            const float hi = v * C_HI;
            const float lo = fmaf(v, C_HI, -hi) + v * C_LO;
            float sc = 1.f;
            for (int o = 0; o < L; ++o) {
                const float rev = __builtin_amdgcn_fractf(hi * sc) + lo * sc;
                w[3 + 6 * o + c] = __builtin_amdgcn_sinf(rev);
                w[6 + 6 * o + c] = __builtin_amdgcn_cosf(rev);
                sc *= 2.f;
            }
        }
    }
    __syncthreads();
    const int total = nrows * E;  // floats; the slab starts 16-byte aligned (64*E*4 bytes per block)
    float* o = out + row0 * E;
    const int nv = total >> 2;
    for (int i = t; i < nv; i += 256) ((f32x4*)o)[i] = ((const f32x4*)tile)[i];
    for (int i = (nv << 2) + t; i < total; i += 256) o[i] = tile[i];
}

extern "C" hipError_t lidf_launch_embed(const float* x, long long n, int L, float* out,
                                        hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const long long blocks = (n + EMBED_ROWS - 1) / EMBED_ROWS;
    const size_t lds = (size_t)EMBED_ROWS * (3 + 6 * L) * sizeof(float);
    hipLaunchKernelGGL(lidf_embed_kernel, dim3((unsigned)blocks), dim3(256), lds, st, x, n, L, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Per-ray features: ROIAlign(output 2x2, spatial_scale 1, sampling_ratio -1, aligned=True) of the
// box built at models/pipeline.py:374-383 (pixel +- roi_inp_bbox//2, corners clamped on integers),
// restating torchvision 0.7.0 roi_align (source not in the reference tree; parity unpinned), plus
// embed(dir). blockDim = (64 rays, 4 channel groups of 8 channels).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bilinear(const float* __restrict__ img, int H, int W, float y,
                                          float x) {
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0.f;
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= H - 1) {
        y_high = y_low = H - 1;
        y = (float)y_low;
    } else {
        y_high = y_low + 1;
    }
    if (x_low >= W - 1) {
        x_high = x_low = W - 1;
        x = (float)x_low;
    } else {
        x_high = x_low + 1;
    }
    const float ly = y - (float)y_low, lx = x - (float)x_low;
    // sample on a pixel centre: weights are exactly (1,0,0,0) -> the value itself (finite maps)
    if (ly == 0.f && lx == 0.f) return img[y_low * W + x_low];
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float v1 = img[y_low * W + x_low], v2 = img[y_low * W + x_high];
    const float v3 = img[y_high * W + x_low], v4 = img[y_high * W + x_high];
    const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
    return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

// The same value without a branch in front of the loads (the four taps are always read — their addresses are
// clamped into the plane — and the result selected afterwards): samples unrolled side by side then have all
// their taps in flight at once instead of one dependent round trip per sample.
__device__ __forceinline__ float bilinear_nb(const float* __restrict__ img, int H, int W, float y, float x) {
    const bool outside = y < -1.0f || y > (float)H || x < -1.0f || x > (float)W;
    y = y <= 0.f ? 0.f : y;
    x = x <= 0.f ? 0.f : x;
    int y_low = outside ? 0 : (int)y, x_low = outside ? 0 : (int)x;
    const bool yc = y_low >= H - 1, xc = x_low >= W - 1;
    y_low = yc ? H - 1 : y_low;
    x_low = xc ? W - 1 : x_low;
    const int y_high = yc ? y_low : y_low + 1, x_high = xc ? x_low : x_low + 1;
    y = yc ? (float)y_low : y;
    x = xc ? (float)x_low : x;
    const float ly = y - (float)y_low, lx = x - (float)x_low;
    const float v1 = img[y_low * W + x_low], v2 = img[y_low * W + x_high];
    const float v3 = img[y_high * W + x_low], v4 = img[y_high * W + x_high];
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
    const float r = (ly == 0.f && lx == 0.f) ? v1 : w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
    return outside ? 0.f : r;
}

// Box-sum image: box[b,c,y,x] = sum of feat[b,c,y..y+k-1,x..x+k-1] (k = roi_inp_bbox/2), zero where
// the window leaves the image. For a ray whose 2k x 2k box is not clamped every RoIAlign sample
// falls on a pixel centre (weights 1,0,0,0), so each of the 2x2 bins is exactly such a window
// mean: 4 gathers per channel instead of (2k)^2 x 4 taps. Sums are formed row-wise
// then column-wise (re-associated w.r.t. the sequential sample loop: <= a few ulp).
__global__ void lidf_boxsum_kernel(const float* __restrict__ feat, int BC, int H, int W, int k,
                                   float* __restrict__ box) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)BC * H * W;
    if (i >= total) return;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    float acc = 0.f;
    if (x + k <= W && y + k <= H) {
        const float* p = feat + i;
        for (int dy = 0; dy < k; ++dy) {
            float row = 0.f;
            for (int dx = 0; dx < k; ++dx) row += p[dy * W + dx];
            acc += row;
        }
    }
    box[i] = acc;
}

// The same sums for k <= 8 through LDS: a workgroup owns 64 x 16 outputs of one channel plane,
// stages the (64+k-1) x (16+k-1) source patch, forms the k-wide row sums once per source pixel and
// adds k of them per output — (2k + 1) LDS reads per output instead of k*k cached global loads.
// Same additions in the same order as lidf_boxsum_kernel (row sums from 0, then rows from 0):
// bit-identical.
#define BOX_TX 64
#define BOX_TY 16
__global__ void __launch_bounds__(256) lidf_boxsum_lds_kernel(const float* __restrict__ feat, int H,
                                                              int W, int k,
                                                              float* __restrict__ box) {
    __shared__ float s_in[(BOX_TY + 7) * (BOX_TX + 7)];
    __shared__ float s_rs[(BOX_TY + 7) * BOX_TX];
    const int x0 = blockIdx.x * BOX_TX, y0 = blockIdx.y * BOX_TY;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const int pw = BOX_TX + k - 1, ph = BOX_TY + k - 1;
    for (int i = threadIdx.x; i < ph * pw; i += 256) {
        const int yy = i / pw, xx = i - yy * pw;
        const int gy = y0 + yy, gx = x0 + xx;
        s_in[i] = (gy < H && gx < W) ? feat[plane + (size_t)gy * W + gx] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ph * BOX_TX; i += 256) {
        const int yy = i >> 6, xx = i & 63;
        float row = 0.f;
        for (int dx = 0; dx < k; ++dx) row += s_in[yy * pw + xx + dx];
        s_rs[i] = row;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < BOX_TY * BOX_TX; i += 256) {
        const int yy = i >> 6, xx = i & 63;
        const int gy = y0 + yy, gx = x0 + xx;
        if (gy >= H || gx >= W) continue;
        float acc = 0.f;
        if (gx + k <= W && gy + k <= H)
            for (int dy = 0; dy < k; ++dy) acc += s_rs[(yy + dy) * BOX_TX + xx];
        box[plane + (size_t)gy * W + gx] = acc;
    }
}

// Per-ray [ROI 2x2 x 32 channels | embed(dir)]. Workgroup = 64 rays x 4 channel groups.
// Phase 1: rays whose 2k x 2k box is unclamped read 4 box sums per channel. Phase 2: the
// workgroup's remaining (border / no box image) rays are compacted and their (ray, channel, bin)
// items spread over all 256 threads, so a few clamped boxes do not serialise whole wavefronts.
// The 64 x (128 + Ed) block — ROI bins and embed(dir), whose octaves the four wavefronts share — is
// staged in LDS and leaves as whole contiguous rows.
__global__ void lidf_rayfeat_kernel(const float* __restrict__ feat,
                                    const float* __restrict__ box, int B, int H, int W,
                                    const float* __restrict__ ray_dir,
                                    const int* __restrict__ ray_pix,
                                    const int* __restrict__ ray_bid, long long R, int half,
                                    int Lv, float* __restrict__ out, int ld,
                                    int* __restrict__ border, const int* __restrict__ R_dev) {
    if (R_dev) R = *R_dev;   // device-side ray count (the sync-free frame path; R = launch capacity)
    if ((long long)blockIdx.x * 64 >= R) return;
    // [64 rays][128 + Ed] (row stride TS; 155 for Lv = 4: odd, conflict-free), then the list of
    // rays that take the general path and its length
    extern __shared__ float rf_lds[];
    const int TS = 128 + 3 + 6 * Lv + ((3 + 6 * Lv) & 1 ? 0 : 1);
    float* tile = rf_lds;
    int* s_list = (int*)(rf_lds + 64 * TS);
    int& s_nb = s_list[64];
    const long long r0 = (long long)blockIdx.x * 64;
    const int lx = threadIdx.x, cg = threadIdx.y;
    const int tid = cg * 64 + lx;
    const int nrow = (int)min((long long)64, R - r0);
    const bool live = lx < nrow;
    const long long r = r0 + (live ? lx : 0);
    const int px = ray_pix[2 * r], py = ray_pix[2 * r + 1], b = ray_bid[r];
    const int x1 = min(max(px - half, 0), W - 1), x2 = min(max(px + half, 0), W - 1);
    const int y1 = min(max(py - half, 0), H - 1), y2 = min(max(py + half, 0), H - 1);
    const bool fast = box && half > 0 && x2 - x1 == 2 * half && y2 - y1 == 2 * half;
    if (cg == 0) {  // one wavefront compacts the rays that need the general path
        const unsigned long long m = __ballot(live && !fast);
        if (live && !fast) s_list[__popcll(m & ((1ull << lx) - 1))] = lx;
        if (lx == 0) s_nb = __popcll(m);
    }
    if (live && fast) {
        // unclamped box: bin (ph,pw) = mean of the half x half pixel block at (y1+ph*half, x1+pw*half)
        const float count = (float)(half * half);
        float* o = tile + lx * TS;
        for (int c = cg * 8; c < cg * 8 + 8; ++c) {
            const float* bi = box + ((size_t)b * 32 + c) * H * W + (size_t)y1 * W + x1;
            o[c * 4 + 0] = bi[0] / count;
            o[c * 4 + 1] = bi[half] / count;
            o[c * 4 + 2] = bi[(size_t)half * W] / count;
            o[c * 4 + 3] = bi[(size_t)half * W + half] / count;
        }
    }
    __syncthreads();
    int nb = s_nb;
    if (border) {
        // the clamped boxes of the whole launch are collected (border[0] = count, then the ray ids)
        // and evaluated by lidf_rayfeat_border_kernel over all of them at once: here they would put
        // a serial tail of up to 64 samples x 4 taps per item behind every workgroup's fast path
        if (cg == 0) {
            int base = 0;
            if (lx == 0 && nb > 0) base = atomicAdd(border, nb);
            base = __shfl(base, 0);
            if (lx < nb) border[1 + base + lx] = (int)(r0 + s_list[lx]);
        }
        nb = 0;
    }
    for (int item = tid; item < nb * 128; item += 256) {
        const int row = s_list[item >> 7], cb = item & 127;
        const int c = cb >> 2, ph = (cb >> 1) & 1, pw = cb & 1;
        const long long rr = r0 + row;
        const int qx = ray_pix[2 * rr], qy = ray_pix[2 * rr + 1];
        const int u1 = min(max(qx - half, 0), W - 1), u2 = min(max(qx + half, 0), W - 1);
        const int v1 = min(max(qy - half, 0), H - 1), v2 = min(max(qy + half, 0), H - 1);
        const float rsw = (float)u1 - 0.5f, rsh = (float)v1 - 0.5f;
        const float rew = (float)u2 - 0.5f, reh = (float)v2 - 0.5f;
        const float roi_w = rew - rsw, roi_h = reh - rsh;
        const float bin_w = roi_w / 2.f, bin_h = roi_h / 2.f;
        const int gw = (int)ceilf(roi_w / 2.f), gh = (int)ceilf(roi_h / 2.f);
        const float count = (float)max(gh * gw, 1);
        const float* img = feat + ((size_t)ray_bid[rr] * 32 + c) * H * W;
        // the samples of one bin are independent loads: blocks of 4 x 4 are evaluated with the
        // loops unrolled (predicated) so that they issue back to back, then summed in the
        // reference's (iy, ix) order
        float acc = 0.f;
        for (int iy0 = 0; iy0 < gh; iy0 += 4) {
            for (int ix0 = 0; ix0 < gw; ix0 += 4) {
                float v[4][4];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const float y =
                        rsh + (float)ph * bin_h + ((float)(iy0 + a) + .5f) * bin_h / (float)gh;
#pragma unroll
                    for (int bb = 0; bb < 4; ++bb) {
                        const float x =
                            rsw + (float)pw * bin_w + ((float)(ix0 + bb) + .5f) * bin_w / (float)gw;
                        v[a][bb] = (iy0 + a < gh && ix0 + bb < gw) ? bilinear(img, H, W, y, x) : 0.f;
                    }
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) {
#pragma unroll
                    for (int bb = 0; bb < 4; ++bb)
                        if (iy0 + a < gh && ix0 + bb < gw) acc += v[a][bb];
                }
            }
        }
        tile[row * TS + cb] = acc / count;
    }
    // embed(dir) into the row's tail: wavefront cg takes the octaves cg, cg + 4, ...
    if (live) {
        const float d[3] = {ray_dir[3 * r], ray_dir[3 * r + 1], ray_dir[3 * r + 2]};
        float* e = tile + lx * TS + 128;
        if (cg == 0)
            for (int i = 0; i < 3; ++i) e[i] = d[i];
        for (int l = cg; l < Lv; l += 4) {
            const float sc = (float)(1 << l);
            for (int i = 0; i < 3; ++i) rev_sincos(to_rev(d[i]), sc, e[3 + 6 * l + i], e[3 + 6 * l + 3 + i]);
        }
    }
    __syncthreads();
    // rows leave as contiguous segments (a wavefront per row); the ROI part of a clamped box's row
    // is written by the border kernel
    const int ncol = 128 + 3 + 6 * Lv;
    for (int row = cg; row < nrow; row += 4) {
        bool slow = false;
        if (border)
            for (int i = 0; i < s_nb; ++i) slow |= s_list[i] == row;
        float* orow = out + (size_t)(r0 + row) * ld;
        for (int c = slow ? 128 + lx : lx; c < ncol; c += 64) orow[c] = tile[row * TS + c];
    }
}

// The collected clamped-box rays: one (ray, channel, bin) item per thread with the RAY index fastest —
// consecutive list entries are mostly neighbouring pixels of a border row, so the taps of a
// wavefront fall into the same rows of one channel plane (coalesced) — grid-stride over all items
// (the count is read on the device: no host round trip).
__global__ void __launch_bounds__(256) lidf_rayfeat_border_kernel(
    const float* __restrict__ feat, int H, int W, const int* __restrict__ ray_pix,
    const int* __restrict__ ray_bid, int half, const int* __restrict__ border,
    float* __restrict__ out, int ld) {
    const long long nb = border[0];
    const long long nitem = nb * 128;   // (bin, channel, ray): ray fastest, then channel, then bin
    for (long long item = (long long)blockIdx.x * 256 + threadIdx.x; item < nitem;
         item += (long long)gridDim.x * 256) {
        const long long rr = border[1 + item % nb];
        const int cbin = (int)(item / nb);
        const int c = cbin & 31, bin = cbin >> 5;
        const int qx = ray_pix[2 * rr], qy = ray_pix[2 * rr + 1];
        const int u1 = min(max(qx - half, 0), W - 1), u2 = min(max(qx + half, 0), W - 1);
        const int v1 = min(max(qy - half, 0), H - 1), v2 = min(max(qy + half, 0), H - 1);
        const float rsw = (float)u1 - 0.5f, rsh = (float)v1 - 0.5f;
        const float rew = (float)u2 - 0.5f, reh = (float)v2 - 0.5f;
        const float roi_w = rew - rsw, roi_h = reh - rsh;
        const float bin_w = roi_w / 2.f, bin_h = roi_h / 2.f;
        const int gw = (int)ceilf(roi_w / 2.f), gh = (int)ceilf(roi_h / 2.f);
        const float count = (float)max(gh * gw, 1);
        const float* img = feat + ((size_t)ray_bid[rr] * 32 + c) * H * W;
        const int ph = bin >> 1, pw = bin & 1;
        // boxes of up to 4 x 4 samples per bin (roi_inp_bbox <= 8): the samples are independent loads,
        // requested back to back with the loops unrolled (predicated), then summed in the reference's
        // (iy, ix) order; larger bins take the plain loops (same order)
        float acc = 0.f;
        if (gh <= 4 && gw <= 4) {
            float v[4][4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const float y = rsh + (float)ph * bin_h + ((float)a + .5f) * bin_h / (float)gh;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const float x = rsw + (float)pw * bin_w + ((float)bb + .5f) * bin_w / (float)gw;
                    v[a][bb] = (a < gh && bb < gw) ? bilinear_nb(img, H, W, y, x) : 0.f;
                }
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
#pragma unroll
                for (int bb = 0; bb < 4; ++bb)
                    if (a < gh && bb < gw) acc += v[a][bb];
            }
        } else {
            for (int iy = 0; iy < gh; ++iy) {
                const float y = rsh + (float)ph * bin_h + ((float)iy + .5f) * bin_h / (float)gh;
                for (int ix = 0; ix < gw; ++ix) {
                    const float x = rsw + (float)pw * bin_w + ((float)ix + .5f) * bin_w / (float)gw;
                    acc += bilinear(img, H, W, y, x);
                }
            }
        }
        out[(size_t)rr * ld + 4 * c + bin] = acc / count;
    }
}

// RoIAlign at any channel count and output size (torchvision.ops.roi_align, aligned = True,
// sampling_ratio = -1, on the per-ray boxes of models/pipeline.py:374-391; rgb_out / roi_out_bbox other
// than the shipped 32 / 2): one (ray, channel, bin) item per thread with the ray index fastest, as the
// border kernel above; out[r, (c * S + ph) * S + pw] = the reference's reshape of [K, C, S, S].
__global__ void __launch_bounds__(256) lidf_roi_align_kernel(
    const float* __restrict__ feat, int Cn, int H, int W, const int* __restrict__ ray_pix,
    const int* __restrict__ ray_bid, long long R, int half, int S, float* __restrict__ out, long long ld) {
    const long long nitem = R * Cn * S * S;
    for (long long item = (long long)blockIdx.x * 256 + threadIdx.x; item < nitem;
         item += (long long)gridDim.x * 256) {
        const long long rr = item % R;
        const int cbin = (int)(item / R);
        const int bin = cbin % (S * S), c = cbin / (S * S);
        const int qx = ray_pix[2 * rr], qy = ray_pix[2 * rr + 1];
        const int u1 = min(max(qx - half, 0), W - 1), u2 = min(max(qx + half, 0), W - 1);
        const int v1 = min(max(qy - half, 0), H - 1), v2 = min(max(qy + half, 0), H - 1);
        const float rsw = (float)u1 - 0.5f, rsh = (float)v1 - 0.5f;
        const float rew = (float)u2 - 0.5f, reh = (float)v2 - 0.5f;
        const float roi_w = rew - rsw, roi_h = reh - rsh;
        const float bin_w = roi_w / (float)S, bin_h = roi_h / (float)S;
        const int gw = (int)ceilf(roi_w / (float)S), gh = (int)ceilf(roi_h / (float)S);
        const float count = (float)max(gh * gw, 1);
        const float* img = feat + ((size_t)ray_bid[rr] * Cn + c) * H * W;
        const int ph = bin / S, pw = bin % S;
        float acc = 0.f;  // the reference's (iy, ix) order
        for (int iy = 0; iy < gh; ++iy) {
            const float y = rsh + (float)ph * bin_h + ((float)iy + .5f) * bin_h / (float)gh;
            for (int ix = 0; ix < gw; ++ix) {
                const float x = rsw + (float)pw * bin_w + ((float)ix + .5f) * bin_w / (float)gw;
                acc += bilinear(img, H, W, y, x);
            }
        }
        out[(size_t)rr * ld + (size_t)c * S * S + bin] = acc / count;
    }
}

extern "C" hipError_t lidf_launch_roi_align(const float* feat, int Cn, int H, int W, const int* ray_pix,
                                            const int* ray_bid, long long R, int half, int S, float* out,
                                            long long ld, hipStream_t st) {
    const long long nitem = R * Cn * S * S;
    if (nitem <= 0) return hipSuccess;
    const long long blocks = (nitem + 255) / 256;
    hipLaunchKernelGGL(lidf_roi_align_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0,
                       st, feat, Cn, H, W, ray_pix, ray_bid, R, half, S, out, ld);
    return hipGetLastError();
}

// `box` (scratch, B*32*H*W floats, followed by R+1 ints for the clamped-box list) may be NULL:
// every ray then takes the general path inside the main kernel.
extern "C" hipError_t lidf_launch_rayfeat_dev(const float* feat, float* box, int B, int H, int W,
                                              const float* ray_dir, const int* ray_pix,
                                              const int* ray_bid, long long R, const int* R_dev,
                                              int half, int Lv, float* out, int ld, hipStream_t st);
extern "C" hipError_t lidf_launch_zero_segments(float* const* ptrs, const long long* counts, int n,
                                                hipStream_t st);
extern "C" hipError_t lidf_launch_rayfeat(const float* feat, float* box, int B, int H, int W,
                                          const float* ray_dir, const int* ray_pix,
                                          const int* ray_bid, long long R, int half, int Lv,
                                          float* out, int ld, hipStream_t st) {
    return lidf_launch_rayfeat_dev(feat, box, B, H, W, ray_dir, ray_pix, ray_bid, R, nullptr, half, Lv,
                                   out, ld, st);
}
// R_dev (optional): the ray count on the device, R then bounds the launch
// phase: 0 = everything; 1 = the box-sum image only (needs the feature map alone: the frame path's side
// stream forms it before the rays exist); 2 = the per-ray launches over an image formed by phase 1
extern "C" hipError_t lidf_launch_rayfeat_phase(const float* feat, float* box, int B, int H, int W,
                                                const float* ray_dir, const int* ray_pix,
                                                const int* ray_bid, long long R, const int* R_dev,
                                                int half, int Lv, float* out, int ld, int phase,
                                                hipStream_t st) {
    if (R <= 0) return hipSuccess;
    int* border = nullptr;
    if (box && half > 0) {
        const long long total = (long long)B * 32 * H * W;
        if (phase == 2) {
        } else if (half <= 8 && (long long)B * 32 <= 65535)
            hipLaunchKernelGGL(lidf_boxsum_lds_kernel,
                               dim3((unsigned)((W + BOX_TX - 1) / BOX_TX), (unsigned)((H + BOX_TY - 1) / BOX_TY),
                                    (unsigned)(B * 32)),
                               dim3(256), 0, st, feat, H, W, half, box);
        else
            hipLaunchKernelGGL(lidf_boxsum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                               st, feat, B * 32, H, W, half, box);
        if (phase == 1) return hipGetLastError();
        border = (int*)(box + total);
        // (the frame path — R_dev — zeroes the list length with its other scratch, in one launch up front)
        if (!R_dev) {
            hipError_t e = hipMemsetAsync(border, 0, 4, st);
            if (e != hipSuccess) return e;
        }
    }
    const int ts = 128 + 3 + 6 * Lv + ((3 + 6 * Lv) & 1 ? 0 : 1);
    hipLaunchKernelGGL(lidf_rayfeat_kernel, dim3((unsigned)((R + 63) / 64)), dim3(64, 4),
                       (size_t)(64 * ts + 65) * 4, st, feat, box, B, H, W, ray_dir, ray_pix, ray_bid,
                       R, half, Lv, out, ld, border, R_dev);
    if (border)
        // (grid-stride over the device-side list; 4096 workgroups hold the ~4,400 clamped boxes x 128 items
        // of a 240x320 frame in one pass)
        hipLaunchKernelGGL(lidf_rayfeat_border_kernel, dim3(4096), dim3(256), 0, st, feat, H, W,
                           ray_pix, ray_bid, half, border, out, ld);
    return hipGetLastError();
}
extern "C" hipError_t lidf_launch_rayfeat_dev(const float* feat, float* box, int B, int H, int W,
                                              const float* ray_dir, const int* ray_pix,
                                              const int* ray_bid, long long R, const int* R_dev,
                                              int half, int Lv, float* out, int ld, hipStream_t st) {
    return lidf_launch_rayfeat_phase(feat, box, B, H, W, ray_dir, ray_pix, ray_bid, R, R_dev, half, Lv, out, ld,
                                     0, st);
}

// ------------------------------------------------------------------------------------------------
// Zero up to ZERO_SEGS float arrays in one launch (the gradient buffers of a decoder: ten
// hipMemsetAsync calls, each a fill kernel of its own, before).
// ------------------------------------------------------------------------------------------------
#define ZERO_SEGS 12
struct ZeroSegs {
    float* p[ZERO_SEGS];
    long long end[ZERO_SEGS];   // running element counts: segment i covers [end[i-1], end[i])
    int n;
};
__global__ void lidf_zero_segments_kernel(ZeroSegs z) {
    const long long total = z.n ? z.end[z.n - 1] : 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int sgm = 0;
#pragma unroll
        for (int k = 0; k < ZERO_SEGS - 1; ++k) sgm += (k < z.n - 1 && i >= z.end[k]) ? 1 : 0;
        z.p[sgm][i - (sgm ? z.end[sgm - 1] : 0)] = 0.f;
    }
}
extern "C" hipError_t lidf_launch_zero_segments(float* const* ptrs, const long long* counts, int n,
                                                hipStream_t st) {
    ZeroSegs z = {};
    long long run = 0;
    for (int i = 0; i < n && z.n < ZERO_SEGS; ++i) {
        if (!ptrs[i] || counts[i] <= 0) continue;
        run += counts[i];
        z.p[z.n] = ptrs[i];
        z.end[z.n] = run;
        ++z.n;
    }
    if (!z.n) return hipSuccess;
    const long long blocks = (run + 255) / 256;
    hipLaunchKernelGGL(lidf_zero_segments_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)),
                       dim3(256), 0, st, z);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Per-ray softmax / argmax / select — scatter_softmax + scatter_max + dummy-row gather
// (models/pipeline.py:442-454) on ray-major CSR pairs: one wavefront per ray, wave-level
// shuffles for max, sum and arg-max. Ties: lowest pair index. Empty ray: id = P, pos = 0.
// ------------------------------------------------------------------------------------------------
// G lanes per ray: 64 (a wavefront) for long candidate lists, 8 for short ones (a geometry-derived
// frame has 3-4 pairs per ray: a wavefront per ray would idle 60 of its lanes).
template <int G>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int s = G / 2; s >= 1; s >>= 1) v = fmaxf(v, __shfl_xor(v, s));
    return v;
}
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int s = G / 2; s >= 1; s >>= 1) v += __shfl_xor(v, s);
    return v;
}

// U rays per group, handled side by side: the loads of all U rays are requested before any is used
// (the kernel is a chain of four dependent memory round trips per ray and nothing else).
template <int G, int U>
__global__ void lidf_ray_reduce_kernel(const float* __restrict__ prob,
                                       const float* __restrict__ pos,
                                       const int* __restrict__ off, long long R, long long P,
                                       const int* __restrict__ ray_bid,
                                       const int* __restrict__ ray_flat, long long hw,
                                       float* __restrict__ softmax, long long* __restrict__ maxid,
                                       float* __restrict__ pred_pos, float* __restrict__ depth,
                                       const int* __restrict__ R_dev, const int* __restrict__ P_dev,
                                       const int* __restrict__ pair_vox, const float* __restrict__ pair_t,
                                       int* __restrict__ sel_ray, int* __restrict__ sel_vox,
                                       float* __restrict__ sel_t) {
    if (R_dev) R = *R_dev;   // device-side counts (the sync-free frame path)
    if (P_dev) P = *P_dev;
    const int lane = threadIdx.x & (G - 1);
    const long long grp = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / G;
    // (a whole group shares its rays; groups beyond R idle through the shuffles with empty ranges)
    long long ray[U];
    bool live[U];
    int beg[U], end[U];
    bool fast = true;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        ray[u] = grp * U + u;
        live[u] = ray[u] < R;
        beg[u] = live[u] ? off[ray[u]] : 0;
        end[u] = live[u] ? off[ray[u] + 1] : 0;
        fast &= end[u] - beg[u] <= G;
    }
    float bv[U];
    int bi[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        bv[u] = -INFINITY;
        bi[u] = 0x7fffffff;
    }
    if (fast) {
        // at most one candidate per lane (every ray of the dense headline list at G = 64, nearly
        // every ray of a geometry-derived frame at G = 8): the logit is read once and its
        // exponential formed once — the same values the general loops below produce
        float p[U];
#pragma unroll
        for (int u = 0; u < U; ++u) p[u] = beg[u] + lane < end[u] ? prob[beg[u] + lane] : -INFINITY;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = beg[u] + lane;
            const bool has = i < end[u];
            const float m = group_max<G>(fmaxf(-INFINITY, p[u]));   // (a NaN logit drops out, as in the loop)
            const float e = has ? expf(p[u] - m) : 0.f;
            const float s = group_sum<G>(e);
            if (has) {
                const float v = e / s;
                if (softmax) softmax[i] = v;
                if (v > bv[u]) {
                    bv[u] = v;
                    bi[u] = i;
                }
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float m = -INFINITY;
            for (int i = beg[u] + lane; i < end[u]; i += G) m = fmaxf(m, prob[i]);
            m = group_max<G>(m);
            float s = 0.f;
            for (int i = beg[u] + lane; i < end[u]; i += G) s += expf(prob[i] - m);
            s = group_sum<G>(s);
            for (int i = beg[u] + lane; i < end[u]; i += G) {
                const float v = expf(prob[i] - m) / s;
                if (softmax) softmax[i] = v;
                if (v > bv[u]) {  // ascending i per lane: strict > keeps the first on ties
                    bv[u] = v;
                    bi[u] = i;
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int sh = G / 2; sh >= 1; sh >>= 1) {
            const float ov = __shfl_xor(bv[u], sh);
            const int oi = __shfl_xor(bi[u], sh);
            if (ov > bv[u] || (ov == bv[u] && oi < bi[u])) {
                bv[u] = ov;
                bi[u] = oi;
            }
        }
    }
    // every lane of the group holds the U results; lane u writes ray u (one pass of loads and
    // stores for the group instead of U passes on lane 0)
    if (lane < U) {
        long long my_ray = ray[0], my_dst = 0;
        int my_bi = bi[0], my_beg = beg[0], my_end = end[0];
        bool my_live = live[0];
#pragma unroll
        for (int u = 1; u < U; ++u) {
            if (lane == u) {
                my_ray = ray[u], my_bi = bi[u], my_beg = beg[u], my_end = end[u], my_live = live[u];
            }
        }
        if (my_live) {
            if (depth) my_dst = (long long)ray_bid[my_ray] * hw + ray_flat[my_ray];
            // no candidate, or no softmax value compared greater than -inf (NaN / +-inf logits
            // make every value NaN): torch_scatter's scatter_max leaves its out-of-range index
            // then, which selects the dummy row (pipeline.py:452-454) -> id = P, position (0,0,0)
            const bool empty = my_end <= my_beg || my_bi >= my_end;
            float x = 0.f, y = 0.f, z = 0.f;
            if (!empty && pos) {
                x = pos[3 * (size_t)my_bi];
                y = pos[3 * (size_t)my_bi + 1];
                z = pos[3 * (size_t)my_bi + 2];
            }
            if (maxid) maxid[my_ray] = empty ? P : (long long)my_bi;
            if (sel_ray) {   // the selected pair of the ray as a one-pair-per-ray list (offsets for those only)
                sel_ray[my_ray] = (int)my_ray;
                sel_vox[my_ray] = empty ? 0 : pair_vox[my_bi];
                const f32x2 tt = empty ? f32x2{0.f, 0.f} : *(const f32x2*)(pair_t + 2 * (size_t)my_bi);
                *(f32x2*)(sel_t + 2 * my_ray) = tt;
            }
            if (pred_pos) {
                pred_pos[3 * my_ray] = x;
                pred_pos[3 * my_ray + 1] = y;
                pred_pos[3 * my_ray + 2] = z;
            }
            if (depth) depth[my_dst] = z;
        }
    }
}

#define REDUCE_U 4
extern "C" hipError_t lidf_launch_ray_reduce_dev(const float* prob, const float* pos, const int* off,
                                                 long long R, long long P, const int* R_dev,
                                                 const int* P_dev, const int* ray_bid,
                                                 const int* ray_flat, long long hw, float* softmax,
                                                 long long* maxid, float* pred_pos, float* depth,
                                                 hipStream_t st, const int* pair_vox = nullptr,
                                                 const float* pair_t = nullptr, int* sel_ray = nullptr,
                                                 int* sel_vox = nullptr, float* sel_t = nullptr) {
    if (R <= 0) return hipSuccess;
    const long long groups = (R + REDUCE_U - 1) / REDUCE_U;
    // (with device-side counts the list is a geometry-derived one: a handful of pairs per ray)
    if (P <= 8 * R || P_dev)
        hipLaunchKernelGGL((lidf_ray_reduce_kernel<8, REDUCE_U>), dim3((unsigned)((groups + 31) / 32)),
                           dim3(256), 0, st, prob, pos, off, R, P, ray_bid, ray_flat, hw, softmax, maxid,
                           pred_pos, depth, R_dev, P_dev, pair_vox, pair_t, sel_ray, sel_vox, sel_t);
    else
        hipLaunchKernelGGL((lidf_ray_reduce_kernel<64, REDUCE_U>), dim3((unsigned)((groups + 3) / 4)),
                           dim3(256), 0, st, prob, pos, off, R, P, ray_bid, ray_flat, hw, softmax, maxid,
                           pred_pos, depth, R_dev, P_dev, pair_vox, pair_t, sel_ray, sel_vox, sel_t);
    return hipGetLastError();
}

// Offsets for the selected pairs only (LidfQueryArgs.offsets_selected): after the offset decoder ran on the
// one-pair-per-ray list, ray r takes its position (or the dummy row's zeros, pipeline.py:452-454), the depth
// map its z, and the selected pair's own slots of the per-pair arrays receive the values.
__global__ void lidf_selected_finish_kernel(const long long* __restrict__ maxid, const float* __restrict__ off_sel,
                                            const float* __restrict__ pos_sel, long long R, long long P,
                                            const int* __restrict__ R_dev, const int* __restrict__ P_dev,
                                            const int* __restrict__ ray_bid, const int* __restrict__ ray_flat,
                                            long long hw, float* __restrict__ pred_offset,
                                            float* __restrict__ pair_pred_pos, float* __restrict__ pred_pos,
                                            float* __restrict__ depth) {
    if (R_dev) R = *R_dev;
    if (P_dev) P = *P_dev;
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const long long m = maxid[r];
    const bool empty = m < 0 || m >= P;
    const float x = empty ? 0.f : pos_sel[3 * r], y = empty ? 0.f : pos_sel[3 * r + 1],
                z = empty ? 0.f : pos_sel[3 * r + 2];
    if (!empty) {
        pred_offset[m] = off_sel[r];
        pair_pred_pos[3 * m] = x;
        pair_pred_pos[3 * m + 1] = y;
        pair_pred_pos[3 * m + 2] = z;
    }
    if (pred_pos) {
        pred_pos[3 * r] = x;
        pred_pos[3 * r + 1] = y;
        pred_pos[3 * r + 2] = z;
    }
    if (depth) depth[(long long)ray_bid[r] * hw + ray_flat[r]] = z;
}
extern "C" hipError_t lidf_launch_selected_finish(const long long* maxid, const float* off_sel, const float* pos_sel,
                                                  long long R, long long P, const int* R_dev, const int* P_dev,
                                                  const int* ray_bid, const int* ray_flat, long long hw,
                                                  float* pred_offset, float* pair_pred_pos, float* pred_pos,
                                                  float* depth, hipStream_t st) {
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_selected_finish_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, maxid,
                       off_sel, pos_sel, R, P, R_dev, P_dev, ray_bid, ray_flat, hw, pred_offset, pair_pred_pos,
                       pred_pos, depth);
    return hipGetLastError();
}
// The one-pair-per-ray list of a GIVEN selection (training with ground-truth labels, pipeline.py:444-446): what
// lidf_ray_reduce_kernel leaves for its own arg-max. A ray whose id is outside [0, P) takes voxel 0 and t = 0.
__global__ void lidf_sel_from_ids_kernel(const long long* __restrict__ id, const int* __restrict__ pair_vox,
                                         const float* __restrict__ pair_t, long long R, long long P,
                                         int* __restrict__ sel_ray, int* __restrict__ sel_vox,
                                         float* __restrict__ sel_t) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const long long m = id[r];
    const bool empty = m < 0 || m >= P;
    sel_ray[r] = (int)r;
    sel_vox[r] = empty ? 0 : pair_vox[m];
    *(f32x2*)(sel_t + 2 * r) = empty ? f32x2{0.f, 0.f} : *(const f32x2*)(pair_t + 2 * (size_t)m);
}
extern "C" hipError_t lidf_launch_sel_from_ids(const long long* id, const int* pair_vox, const float* pair_t,
                                               long long R, long long P, int* sel_ray, int* sel_vox, float* sel_t,
                                               hipStream_t st) {
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_sel_from_ids_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, id, pair_vox,
                       pair_t, R, P, sel_ray, sel_vox, sel_t);
    return hipGetLastError();
}
extern "C" hipError_t lidf_launch_ray_reduce(const float* prob, const float* pos, const int* off,
                                             long long R, long long P, const int* ray_bid,
                                             const int* ray_flat, long long hw, float* softmax,
                                             long long* maxid, float* pred_pos, float* depth,
                                             hipStream_t st) {
    return lidf_launch_ray_reduce_dev(prob, pos, off, R, P, nullptr, nullptr, ray_bid, ray_flat, hw, softmax,
                                      maxid, pred_pos, depth, st);
}

// ------------------------------------------------------------------------------------------------
// Ray directions — models/pipeline.py:215-219.
// ------------------------------------------------------------------------------------------------
__global__ void lidf_ray_dirs_kernel(const float* __restrict__ intr, int B, int H, int W,
                                     float* __restrict__ dir) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * H * W;
    if (i >= total) return;
    const int b = (int)(i / ((long long)H * W));
    const int rem = (int)(i % ((long long)H * W));
    const int y = rem / W, x = rem % W;
    const float fx = intr[4 * b], fy = intr[4 * b + 1], cx = intr[4 * b + 2], cy = intr[4 * b + 3];
    const float vx = (float)x - cx;
    const float vy = ((float)y - cy) * fx / fy;
    const float vz = fx;
    const float nrm = sqrtf(vx * vx + vy * vy + vz * vz);
    dir[3 * i] = vx / nrm;
    dir[3 * i + 1] = vy / nrm;
    dir[3 * i + 2] = vz / nrm;
}

extern "C" hipError_t lidf_launch_ray_dirs(const float* intr, int B, int H, int W, float* dir,
                                           hipStream_t st) {
    long long total = (long long)B * H * W;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_ray_dirs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       st, intr, B, H, W, dir);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Ray / voxel slab test — extensions/ray_aabb/ray_aabb_cuda_kernel.cu:24-88, same arithmetic:
// origin 0, inv = 1/(d + 1e-12) evaluated in double then rounded to float, near/far bound by the
// sign of inv, no t >= 0 test.
// ------------------------------------------------------------------------------------------------
struct RayInv {
    float ix, iy, iz;
};
// 1 / (dir + 1e-12): double add and divide, rounded to float (ray_aabb_cuda_kernel.cu:32,48,67);
// depends on the ray only, so it is formed once per ray, not once per (ray, voxel)
__device__ __forceinline__ RayInv ray_inv(float dx, float dy, float dz) {
    RayInv r;
    r.ix = (float)(1 / ((double)dx + 1e-12));
    r.iy = (float)(1 / ((double)dy + 1e-12));
    r.iz = (float)(1 / ((double)dz + 1e-12));
    return r;
}
__device__ __forceinline__ bool slab_test(const RayInv& r, const float* vb, float& t_enter,
                                          float& t_leave) {
    const float ix = r.ix, iy = r.iy, iz = r.iz;
    float tmin_max = (ix >= 0 ? vb[0] : vb[3]) * ix;
    float tmax_min = (ix >= 0 ? vb[3] : vb[0]) * ix;
    const float tymin = (iy >= 0 ? vb[1] : vb[4]) * iy;
    const float tymax = (iy >= 0 ? vb[4] : vb[1]) * iy;
    if ((tmin_max > tymax) || (tmax_min < tymin)) return false;
    tmin_max = fmaxf(tmin_max, tymin);
    tmax_min = fminf(tmax_min, tymax);
    const float tzmin = (iz >= 0 ? vb[2] : vb[5]) * iz;
    const float tzmax = (iz >= 0 ? vb[5] : vb[2]) * iz;
    if ((tmin_max > tzmax) || (tmax_min < tzmin)) return false;
    t_enter = fmaxf(tmin_max, tzmin);
    t_leave = fminf(tmax_min, tzmax);
    return true;
}

// dense form: mask [V,R], dist [V,R,2]; grid (ceil(R/256), V) like the reference's (R/1024, V)
__global__ void lidf_ray_aabb_dense_kernel(const float* __restrict__ ray_dir,
                                           const float* __restrict__ vbound,
                                           const int* __restrict__ ray_bid,
                                           const int* __restrict__ vox_bid, long long R,
                                           int* __restrict__ mask, float* __restrict__ dist) {
    const long long v = blockIdx.y;
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    if (ray_bid[r] != vox_bid[v]) return;
    float vb[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) vb[i] = vbound[6 * v + i];
    float t0, t1;
    if (!slab_test(ray_inv(ray_dir[3 * r], ray_dir[3 * r + 1], ray_dir[3 * r + 2]), vb, t0, t1))
        return;
    mask[v * R + r] = 1;
    *(f32x2*)(dist + (v * R + r) * 2) = f32x2{t0, t1};
}

extern "C" hipError_t lidf_launch_ray_aabb_dense(const float* ray_dir, const float* vbound,
                                                 const int* ray_bid, const int* vox_bid,
                                                 long long R, long long V, int* mask, float* dist,
                                                 hipStream_t st) {
    if (R <= 0 || V <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_ray_aabb_dense_kernel, dim3((unsigned)((R + 255) / 256), (unsigned)V),
                       dim3(256), 0, st, ray_dir, vbound, ray_bid, vox_bid, R, mask, dist);
    return hipGetLastError();
}

// compact ray-major form: one thread per ray walks every voxel (bounds staged through LDS in
// chunks of 256 voxels); FILL = false counts hits, FILL = true writes them at pair_off[ray]...
// in ascending voxel order.
template <bool FILL>
__global__ void lidf_ray_aabb_compact_kernel(const float* __restrict__ ray_dir,
                                             const float* __restrict__ vbound,
                                             const int* __restrict__ ray_bid,
                                             const int* __restrict__ vox_bid, long long R,
                                             long long V, int* __restrict__ count,
                                             const int* __restrict__ pair_off,
                                             int* __restrict__ pair_ray,
                                             int* __restrict__ pair_vox,
                                             float* __restrict__ pair_t,
                                             const int* __restrict__ R_dev,
                                             const int* __restrict__ V_dev, long long pair_cap) {
    __shared__ float s_vb[256 * 6];
    __shared__ int s_bid[256];
    if (R_dev) R = *R_dev;   // device-side counts (the sync-free frame path; R, V = launch capacity)
    if (V_dev) V = *V_dev;
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if ((long long)blockIdx.x * blockDim.x >= R) return;   // (workgroup-uniform)
    const bool live = r < R;
    float dx = 0.f, dy = 0.f, dz = 1.f;
    int bid = -1;
    if (live) {
        dx = ray_dir[3 * r];
        dy = ray_dir[3 * r + 1];
        dz = ray_dir[3 * r + 2];
        bid = ray_bid[r];
    }
    int n = 0;
    const RayInv inv = ray_inv(dx, dy, dz);
    const int base = (FILL && live) ? pair_off[r] : 0;
    for (long long v0 = 0; v0 < V; v0 += 256) {
        const int nv = (int)min((long long)256, V - v0);
        __syncthreads();
        for (int i = threadIdx.x; i < nv * 6; i += blockDim.x) s_vb[i] = vbound[6 * v0 + i];
        for (int i = threadIdx.x; i < nv; i += blockDim.x) s_bid[i] = vox_bid[v0 + i];
        __syncthreads();
        if (!live) continue;
        for (int j = 0; j < nv; ++j) {
            if (s_bid[j] != bid) continue;
            float t0, t1;
            if (!slab_test(inv, s_vb + 6 * j, t0, t1)) continue;
            if (FILL && (pair_cap <= 0 || base + n < pair_cap)) {   // (a list cut at its capacity)
                const size_t p = (size_t)base + n;
                pair_ray[p] = (int)r;
                pair_vox[p] = (int)(v0 + j);
                *(f32x2*)(pair_t + 2 * p) = f32x2{t0, t1};
            }
            ++n;
        }
    }
    if (!FILL && live) count[r] = n;
}

extern "C" hipError_t lidf_launch_ray_aabb_compact(bool fill, const float* ray_dir,
                                                   const float* vbound, const int* ray_bid,
                                                   const int* vox_bid, long long R, long long V,
                                                   int* count, const int* pair_off, int* pair_ray,
                                                   int* pair_vox, float* pair_t, hipStream_t st) {
    if (R <= 0) return hipSuccess;
    dim3 grid((unsigned)((R + 255) / 256)), block(256);
    if (fill)
        hipLaunchKernelGGL(lidf_ray_aabb_compact_kernel<true>, grid, block, 0, st, ray_dir, vbound,
                           ray_bid, vox_bid, R, V, count, pair_off, pair_ray, pair_vox, pair_t,
                           (const int*)nullptr, (const int*)nullptr, 0LL);
    else
        hipLaunchKernelGGL(lidf_ray_aabb_compact_kernel<false>, grid, block, 0, st, ray_dir,
                           vbound, ray_bid, vox_bid, R, V, count, pair_off, pair_ray, pair_vox,
                           pair_t, (const int*)nullptr, (const int*)nullptr, 0LL);
    return hipGetLastError();
}

extern "C" hipError_t lidf_launch_ray_aabb_compact_dev(bool fill, const float* ray_dir,
                                                       const float* vbound, const int* ray_bid,
                                                       const int* vox_bid, long long R_cap,
                                                       long long V_cap, const int* R_dev,
                                                       const int* V_dev, int* count,
                                                       const int* pair_off, int* pair_ray,
                                                       int* pair_vox, float* pair_t, long long pair_cap,
                                                       hipStream_t st) {
    if (R_cap <= 0) return hipSuccess;
    dim3 grid((unsigned)((R_cap + 255) / 256)), block(256);
    if (fill)
        hipLaunchKernelGGL(lidf_ray_aabb_compact_kernel<true>, grid, block, 0, st, ray_dir, vbound,
                           ray_bid, vox_bid, R_cap, V_cap, count, pair_off, pair_ray, pair_vox, pair_t,
                           R_dev, V_dev, pair_cap);
    else
        hipLaunchKernelGGL(lidf_ray_aabb_compact_kernel<false>, grid, block, 0, st, ray_dir, vbound,
                           ray_bid, vox_bid, R_cap, V_cap, count, pair_off, pair_ray, pair_vox, pair_t,
                           R_dev, V_dev, 0LL);
    return hipGetLastError();
}

// The compact list in ONE launch with device-side sizes (the sync-free frame path; rounds 2-3: count ->
// three scan launches -> cut at the capacity -> fill): a workgroup of 256 rays counts its hits — keeping
// the first AABB_HITS voxel indices of every ray in LDS —, obtains the number of pairs before it by a
// decoupled look-back over the workgroups (lb_* in lidf_device.h), writes pair_off and fills its pairs
// from the kept indices (a ray with more hits walks the voxels again). The list is cut at pair_cap
// (pair_off clamped, counts[7] bit 0 set); the workgroup of the last ray leaves P and NV + R in `counts`.
// Same slab_test on the same bounds in the same voxel order as the two-pass kernel: bit-identical lists.
#define AABB_HITS 32
__global__ void __launch_bounds__(256) lidf_ray_aabb_onepass_kernel(
    const float* __restrict__ ray_dir, const float* __restrict__ vbound, const int* __restrict__ ray_bid,
    const int* __restrict__ vox_bid, int* __restrict__ counts, int* __restrict__ ticket,
    unsigned long long* __restrict__ status, int* __restrict__ pair_off, int* __restrict__ pair_ray,
    int* __restrict__ pair_vox, float* __restrict__ pair_t, long long pair_cap,
    const int* __restrict__ vox_start) {
    __shared__ int s_hit[AABB_HITS * 256];
    __shared__ int s_tmp[4];
    __shared__ int s_bid;
    __shared__ unsigned long long s_pre;
    const long long R = counts[0], V = counts[2];
    const int bid = lb_ticket(ticket, &s_bid);
    if ((long long)bid * 256 >= R && !(R == 0 && bid == 0)) return;   // (workgroup-uniform)
    const int lane = threadIdx.x & 63;
    const long long r = (long long)bid * 256 + threadIdx.x;
    const bool live = r < R;
    float dx = 0.f, dy = 0.f, dz = 1.f;
    int rb = -1;
    if (live) {
        dx = ray_dir[3 * r];
        dy = ray_dir[3 * r + 1];
        dz = ray_dir[3 * r + 2];
        rb = ray_bid[r];
    }
    const RayInv inv = ray_inv(dx, dy, dz);
    int n = 0;
    // the voxel index is wave-uniform: the bounds arrive through the scalar cache as SGPR operands (eight
    // voxels per batch of scalar loads), no LDS staging, no barrier in the loop; the hit decision is
    // slab_test's, comparison by comparison
    // (read through the constant address space: the tables are not written by this launch, and the
    // ticket atomic above would otherwise keep the compiler from scalarising the loads)
    typedef const float __attribute__((address_space(4))) * cf_ptr;
    typedef const int __attribute__((address_space(4))) * ci_ptr;
    const cf_ptr vbc = (cf_ptr)(unsigned long long)vbound;
    const ci_ptr vbidc = (ci_ptr)(unsigned long long)vox_bid;
    // vox_start (optional; several images per batch): the voxels are grouped by image, image i owns
    // [vox_start[i], vox_start[i + 1]) — a wavefront's rays are consecutive pixels of one image (two at a
    // seam), so it walks the voxels of its first to its last image only instead of the whole batch's
    int j0 = 0, Vi = (int)V;
    if (vox_start) {
        int b_lo = live ? rb : 0x7fffffff, b_hi = live ? rb : -1;
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) {
            b_lo = min(b_lo, __shfl_xor(b_lo, sft));
            b_hi = max(b_hi, __shfl_xor(b_hi, sft));
        }
        b_lo = __builtin_amdgcn_readfirstlane(b_lo);
        b_hi = __builtin_amdgcn_readfirstlane(b_hi);
        if (b_hi >= 0) {
            const ci_ptr vsc = (ci_ptr)(unsigned long long)vox_start;
            j0 = vsc[b_lo];
            Vi = vsc[b_hi + 1];
        } else {
            Vi = 0;   // (no live ray in this wavefront)
        }
    }
#pragma unroll 8
    for (int j = j0; j < Vi; ++j) {
        const cf_ptr vb = vbc + 6 * (size_t)j;
        // ((c ? lo : hi) * i written as c ? lo * i : hi * i — the same product, and the six bounds stay
        // wave-uniform scalar loads instead of one load through a per-lane selected address)
        const float lx = vb[0] * inv.ix, hx = vb[3] * inv.ix, ly = vb[1] * inv.iy, hy = vb[4] * inv.iy;
        const float lz = vb[2] * inv.iz, hz = vb[5] * inv.iz;
        const float tx0 = inv.ix >= 0 ? lx : hx, tx1 = inv.ix >= 0 ? hx : lx;
        const float ty0 = inv.iy >= 0 ? ly : hy, ty1 = inv.iy >= 0 ? hy : ly;
        const float tz0 = inv.iz >= 0 ? lz : hz, tz1 = inv.iz >= 0 ? hz : lz;
        const bool missxy = (tx0 > ty1) | (tx1 < ty0);
        const float a0 = fmaxf(tx0, ty0), a1 = fminf(tx1, ty1);
        const bool missz = (a0 > tz1) | (a1 < tz0);
        const bool hit = live & (vbidc[j] == rb) & !missxy & !missz;
        if (hit) {
            if (n < AABB_HITS) s_hit[n * 256 + threadIdx.x] = j;
            ++n;
        }
    }
    int tot;
    const int ex = block_scan_256(n, s_tmp, tot);
    if (threadIdx.x == 0) lb_store(status, bid, bid == 0 ? LIDF_LB_INC : LIDF_LB_AGG, (unsigned)tot);
    if (threadIdx.x < 64) {
        const unsigned long long pre = bid > 0 ? lb_exclusive(status, bid, lane) : 0ull;
        if (lane == 0) {
            if (bid > 0) lb_store(status, bid, LIDF_LB_INC, pre + (unsigned)tot);
            s_pre = pre;
        }
    }
    __syncthreads();
    const long long o = (long long)s_pre + ex;   // (the uncut offset: < 2^31 * ... kept in 64 bits)
    if (live) pair_off[r] = (int)(o > pair_cap ? pair_cap : o);
    if ((live && r == R - 1) || (R == 0 && threadIdx.x == 0)) {
        const long long total = o + n;
        pair_off[R] = (int)(total > pair_cap ? pair_cap : total);
        counts[1] = (int)(total > pair_cap ? pair_cap : total);   // P
        if (total > pair_cap) counts[7] |= 1;
        counts[6] = counts[3] + (int)R;                           // NV + R
    }
    if (!live || n == 0) return;
    if (n <= AABB_HITS) {
        for (int i = 0; i < n; ++i) {
            const long long p = o + i;
            if (p >= pair_cap) break;
            const int v = s_hit[i * 256 + threadIdx.x];
            float vb[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) vb[k] = vbound[6 * (size_t)v + k];
            float t0 = 0.f, t1 = 0.f;
            slab_test(inv, vb, t0, t1);
            pair_ray[p] = (int)r;
            pair_vox[p] = v;
            *(f32x2*)(pair_t + 2 * p) = f32x2{t0, t1};
        }
    } else {   // more hits than the LDS list holds (cannot happen on a 9^3 grid: <= 25 cells per ray)
        int m = 0;
        for (long long v = 0; v < V; ++v) {
            if (vox_bid[v] != rb) continue;
            float vb[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) vb[k] = vbound[6 * (size_t)v + k];
            float t0, t1;
            if (!slab_test(inv, vb, t0, t1)) continue;
            const long long p = o + m;
            ++m;
            if (p >= pair_cap) break;
            pair_ray[p] = (int)r;
            pair_vox[p] = (int)v;
            *(f32x2*)(pair_t + 2 * p) = f32x2{t0, t1};
        }
    }
}

// lb: 64 bytes (ticket) + one status word per 256 rays of capacity, zeroed before the launch
extern "C" size_t lidf_ray_aabb_onepass_lb_bytes(long long R_cap) { return 64 + (size_t)((R_cap + 255) / 256) * 8; }
extern "C" hipError_t lidf_launch_ray_aabb_onepass(const float* ray_dir, const float* vbound, const int* ray_bid,
                                                   const int* vox_bid, long long R_cap, int* counts, void* lb,
                                                   int* pair_off, int* pair_ray, int* pair_vox, float* pair_t,
                                                   long long pair_cap, const int* vox_start, hipStream_t st) {
    if (R_cap <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_ray_aabb_onepass_kernel, dim3((unsigned)((R_cap + 255) / 256)), dim3(256), 0, st,
                       ray_dir, vbound, ray_bid, vox_bid, counts, (int*)lb,
                       (unsigned long long*)((char*)lb + 64), pair_off, pair_ray, pair_vox, pair_t, pair_cap,
                       vox_start);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// The same compact list when the voxels are cells of a regular grid (LIDF.get_occ_vox_bound,
// models/pipeline.py:162-201: bound_min = xmin + coord * part_size, so a bound depends on the cell
// index of its own axis only). The slab test is an intersection of three per-axis t intervals,
//   hit(i,j,k)  <=>  Tx[i] ^ Ty[j] ^ Tz[k] non-empty      (evaluated x^y first, then ^z),
// so a ray does not have to visit every voxel: columns (i,j) whose x and y intervals miss are
// skipped whole, and a cell is looked up in a dense cell -> voxel table only when its three
// intervals meet. The intervals are formed from the voxels' OWN bounds (gathered into per-frame
// per-axis tables) with the arithmetic of slab_test, in the same order: t_enter / t_leave and the
// hit decision are bit-identical to the voxel-by-voxel kernel. Cells are visited in (i,j,k) order
// = ascending voxel index for a voxel list sorted by (frame, x, y, z) (torch.unique's order).
// A bit mask of the occupied z cells per (i,j) column lets the walk touch occupied cells only:
// <= rx*ry column checks + one z check per occupied cell of a passing column, instead of V slab
// tests of 6 products each.
// ------------------------------------------------------------------------------------------------
__global__ void lidf_grid_init_kernel(int* __restrict__ cell, long long ncell,
                                      unsigned* __restrict__ colmask, long long ncol,
                                      float* __restrict__ tab, long long ntab) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ncell) cell[i] = -1;
    if (i < ncol) colmask[i] = 0u;
    // an axis index no voxel has: the empty interval (+inf, -inf) fails every overlap check
    if (i < ntab) *(f32x2*)(tab + 2 * i) = f32x2{__builtin_inff(), -__builtin_inff()};
}

__global__ void lidf_grid_build_kernel(const float* __restrict__ vbound,
                                       const int* __restrict__ vox_bid,
                                       const int* __restrict__ coord, long long V, int B, int rx,
                                       int ry, int rz, int* __restrict__ cell,
                                       unsigned* __restrict__ colmask, float* __restrict__ tab) {
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const int b = vox_bid[v], i = coord[3 * v], j = coord[3 * v + 1], k = coord[3 * v + 2];
    if (b < 0 || b >= B || i < 0 || i >= rx || j < 0 || j >= ry || k < 0 || k >= rz) return;
    cell[(((size_t)b * rx + i) * ry + j) * rz + k] = (int)v;
    // occupied z cells of the (i, j) column, 32 per word
    atomicOr(colmask + (((size_t)b * rx + i) * ry + j) * ((rz + 31) / 32) + (k >> 5), 1u << (k & 31));
    float* t = tab + (size_t)b * (rx + ry + rz) * 2;
    // every voxel of one axis index carries the same pair of bounds: the stores agree
    *(f32x2*)(t + 2 * i) = f32x2{vbound[6 * v + 0], vbound[6 * v + 3]};
    *(f32x2*)(t + 2 * (rx + j)) = f32x2{vbound[6 * v + 1], vbound[6 * v + 4]};
    *(f32x2*)(t + 2 * (rx + ry + k)) = f32x2{vbound[6 * v + 2], vbound[6 * v + 5]};
}

// One thread per ray. A workgroup whose rays all belong to one frame (rays are frame-major, so all
// but the few workgroups on a frame boundary) first copies that frame's tables into LDS — cell
// table, column masks, per-axis bounds: 3.4 KB for the reference's 9^3 grid — and every lookup of
// the walk is an LDS read; other workgroups (and grids beyond the LDS budget) read global memory.
#define GRID_LDS_CELLS 8192
#define GRID_LDS_COLS 2048
#define GRID_LDS_AXES 192
template <bool FILL>
__global__ void __launch_bounds__(256) lidf_ray_aabb_grid_kernel(
    const float* __restrict__ ray_dir, const int* __restrict__ ray_bid, long long R, int B, int rx,
    int ry, int rz, const int* __restrict__ cell, const unsigned* __restrict__ colmask,
    const float* __restrict__ tab, int* __restrict__ count, const int* __restrict__ pair_off,
    int* __restrict__ pair_ray, int* __restrict__ pair_vox, float* __restrict__ pair_t) {
    __shared__ int s_cell[GRID_LDS_CELLS];
    __shared__ unsigned s_col[GRID_LDS_COLS];
    __shared__ float s_tab[2 * GRID_LDS_AXES];
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = r < R;
    const int b = live ? ray_bid[r] : -1;
    const int mw = (rz + 31) / 32;
    const int ncell = rx * ry * rz, ncol = rx * ry * mw, nax = rx + ry + rz;
    // frame of the workgroup's first ray; staged if every live ray has it and the tables fit
    const int b0 = ray_bid[(long long)blockIdx.x * blockDim.x];
    const bool fits = ncell <= GRID_LDS_CELLS && ncol <= GRID_LDS_COLS && nax <= GRID_LDS_AXES &&
                      b0 >= 0 && b0 < B;
    const bool staged = __syncthreads_and(!live || b == b0) && fits;
    const int* cb;
    const unsigned* mb;
    const float* tx;
    if (staged) {
        const int* gc = cell + (size_t)b0 * ncell;
        const unsigned* gm = colmask + (size_t)b0 * ncol;
        const float* gt = tab + (size_t)b0 * nax * 2;
        for (int i = threadIdx.x; i < ncell; i += blockDim.x) s_cell[i] = gc[i];
        for (int i = threadIdx.x; i < ncol; i += blockDim.x) s_col[i] = gm[i];
        for (int i = threadIdx.x; i < 2 * nax; i += blockDim.x) s_tab[i] = gt[i];
        __syncthreads();
        cb = s_cell, mb = s_col, tx = s_tab;
    } else if (b >= 0 && b < B) {
        cb = cell + (size_t)b * ncell, mb = colmask + (size_t)b * ncol, tx = tab + (size_t)b * nax * 2;
    } else {
        cb = nullptr, mb = nullptr, tx = nullptr;
    }
    if (!live) return;
    int n = 0;
    if (tx) {
        const RayInv inv = ray_inv(ray_dir[3 * r], ray_dir[3 * r + 1], ray_dir[3 * r + 2]);
        const float* ty = tx + 2 * rx;
        const float* tz = ty + 2 * ry;
        const size_t base = FILL ? (size_t)pair_off[r] : 0;
        for (int i = 0; i < rx; ++i) {
            const float x0 = tx[2 * i], x1 = tx[2 * i + 1];
            const float a0 = (inv.ix >= 0 ? x0 : x1) * inv.ix;   // tmin of x
            const float a1 = (inv.ix >= 0 ? x1 : x0) * inv.ix;   // tmax of x
            for (int j = 0; j < ry; ++j) {
                const unsigned* mrow = mb + ((size_t)i * ry + j) * mw;
                const float y0 = ty[2 * j], y1 = ty[2 * j + 1];
                const float tymin = (inv.iy >= 0 ? y0 : y1) * inv.iy;
                const float tymax = (inv.iy >= 0 ? y1 : y0) * inv.iy;
                if ((a0 > tymax) || (a1 < tymin)) continue;
                const float m = fmaxf(a0, tymin), M = fminf(a1, tymax);
                const int* cj = cb + ((size_t)i * ry + j) * rz;
                for (int w = 0; w < mw; ++w) {
                    unsigned bits = mrow[w];              // occupied cells of the column, ascending k
                    while (bits) {
                        const int k = 32 * w + __builtin_ctz(bits);
                        bits &= bits - 1;
                        const float z0 = tz[2 * k], z1 = tz[2 * k + 1];
                        const float tzmin = (inv.iz >= 0 ? z0 : z1) * inv.iz;
                        const float tzmax = (inv.iz >= 0 ? z1 : z0) * inv.iz;
                        if ((m > tzmax) || (M < tzmin)) continue;
                        if (FILL) {
                            const size_t p = base + n;
                            pair_ray[p] = (int)r;
                            pair_vox[p] = cj[k];
                            *(f32x2*)(pair_t + 2 * p) = f32x2{fmaxf(m, tzmin), fminf(M, tzmax)};
                        }
                        ++n;
                    }
                }
            }
        }
    }
    if (!FILL) count[r] = n;
}

// cell: [B*rx*ry*rz] ints, colmask: [B*rx*ry*ceil(rz/32)] words, tab: [B*(rx+ry+rz)*2] floats
// (all three inside the caller's workspace)
extern "C" hipError_t lidf_launch_ray_aabb_grid_build(const float* vbound, const int* vox_bid,
                                                      const int* coord, long long V, int B, int rx,
                                                      int ry, int rz, int* cell, unsigned* colmask,
                                                      float* tab, hipStream_t st) {
    const long long ncell = (long long)B * rx * ry * rz, ntab = (long long)B * (rx + ry + rz);
    const long long ncol = (long long)B * rx * ry * ((rz + 31) / 32);
    const long long n = ncell > ntab ? ncell : ntab;   // ncol <= ncell
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_grid_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                       cell, ncell, colmask, ncol, tab, ntab);
    if (V > 0)
        hipLaunchKernelGGL(lidf_grid_build_kernel, dim3((unsigned)((V + 255) / 256)), dim3(256), 0,
                           st, vbound, vox_bid, coord, V, B, rx, ry, rz, cell, colmask, tab);
    return hipGetLastError();
}

extern "C" hipError_t lidf_launch_ray_aabb_grid(bool fill, const float* ray_dir, const int* ray_bid,
                                                long long R, int B, int rx, int ry, int rz,
                                                const int* cell, const unsigned* colmask,
                                                const float* tab, int* count, const int* pair_off,
                                                int* pair_ray, int* pair_vox, float* pair_t,
                                                hipStream_t st) {
    if (R <= 0) return hipSuccess;
    dim3 grid((unsigned)((R + 255) / 256)), block(256);
    if (fill)
        hipLaunchKernelGGL(lidf_ray_aabb_grid_kernel<true>, grid, block, 0, st, ray_dir, ray_bid, R,
                           B, rx, ry, rz, cell, colmask, tab, count, pair_off, pair_ray, pair_vox,
                           pair_t);
    else
        hipLaunchKernelGGL(lidf_ray_aabb_grid_kernel<false>, grid, block, 0, st, ray_dir, ray_bid,
                           R, B, rx, ry, rz, cell, colmask, tab, count, pair_off, pair_ray, pair_vox,
                           pair_t);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Point / voxel inside test — extensions/pcl_aabb/pcl_aabb_cuda_kernel.cu:23-44 (inclusive bounds).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool inside_test(float x, float y, float z, const float* vb) {
    if ((x < vb[0]) || (x > vb[3])) return false;
    if ((y < vb[1]) || (y > vb[4])) return false;
    if ((z < vb[2]) || (z > vb[5])) return false;
    return true;
}

__global__ void lidf_pcl_aabb_dense_kernel(const float* __restrict__ pos,
                                           const float* __restrict__ vbound,
                                           const int* __restrict__ pcl_bid,
                                           const int* __restrict__ vox_bid, long long N,
                                           int* __restrict__ mask) {
    const long long v = blockIdx.y;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    if (pcl_bid[i] != vox_bid[v]) return;
    float vb[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) vb[k] = vbound[6 * v + k];
    if (inside_test(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], vb)) mask[v * N + i] = 1;
}

extern "C" hipError_t lidf_launch_pcl_aabb_dense(const float* pos, const float* vbound,
                                                 const int* pcl_bid, const int* vox_bid,
                                                 long long N, long long V, int* mask,
                                                 hipStream_t st) {
    if (N <= 0 || V <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_pcl_aabb_dense_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)V),
                       dim3(256), 0, st, pos, vbound, pcl_bid, vox_bid, N, mask);
    return hipGetLastError();
}

__global__ void lidf_pcl_aabb_last_kernel(const float* __restrict__ pos,
                                          const float* __restrict__ vbound,
                                          const int* __restrict__ pcl_bid,
                                          const int* __restrict__ vox_bid, long long N,
                                          long long V, int* __restrict__ last) {
    __shared__ float s_vb[256 * 6];
    __shared__ int s_bid[256];
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < N;
    float x = 0.f, y = 0.f, z = 0.f;
    int bid = -1, best = -1;
    if (live) {
        x = pos[3 * i];
        y = pos[3 * i + 1];
        z = pos[3 * i + 2];
        bid = pcl_bid[i];
    }
    for (long long v0 = 0; v0 < V; v0 += 256) {
        const int nv = (int)min((long long)256, V - v0);
        __syncthreads();
        for (int k = threadIdx.x; k < nv * 6; k += blockDim.x) s_vb[k] = vbound[6 * v0 + k];
        for (int k = threadIdx.x; k < nv; k += blockDim.x) s_bid[k] = vox_bid[v0 + k];
        __syncthreads();
        if (!live) continue;
        for (int j = 0; j < nv; ++j)
            if (s_bid[j] == bid && inside_test(x, y, z, s_vb + 6 * j)) best = (int)(v0 + j);
    }
    if (live) last[i] = best;
}

extern "C" hipError_t lidf_launch_pcl_aabb_last(const float* pos, const float* vbound,
                                                const int* pcl_bid, const int* vox_bid,
                                                long long N, long long V, int* last,
                                                hipStream_t st) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_pcl_aabb_last_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0,
                       st, pos, vbound, pcl_bid, vox_bid, N, V, last);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Exclusive scan of int32 counts (pair_off): block sums -> serial scan of sums -> block rescan.
// ------------------------------------------------------------------------------------------------
#define SCAN_ITEMS 1024  // per block: 256 threads x 4

__global__ void lidf_scan_sums_kernel(const int* __restrict__ in, long long n,
                                      int* __restrict__ sums, const int* __restrict__ n_dev) {
    __shared__ int s_tmp[4];
    if (n_dev) n = *n_dev;   // device-side length: entries beyond it count as 0 (n = launch capacity)
    const long long b0 = (long long)blockIdx.x * SCAN_ITEMS + threadIdx.x * 4;
    int v = 0;
    for (int k = 0; k < 4; ++k)
        if (b0 + k < n) v += in[b0 + k];
    int total;
    block_scan_256(v, s_tmp, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ void lidf_scan_top_kernel(int* __restrict__ sums, long long nb) {
    // single block: exclusive scan of the block sums, 256 at a time with a running carry
    __shared__ int s_tmp[4];
    int carry = 0;
    for (long long b = 0; b < nb; b += 256) {
        const long long i = b + threadIdx.x;
        const int v = i < nb ? sums[i] : 0;
        int total;
        const int ex = block_scan_256(v, s_tmp, total);
        if (i < nb) sums[i] = carry + ex;
        carry += total;
    }
}

__global__ void lidf_scan_final_kernel(const int* __restrict__ in, long long n,
                                       const int* __restrict__ sums, int* __restrict__ out,
                                       const int* __restrict__ n_dev, int* __restrict__ total_out) {
    __shared__ int s_tmp[4];
    if (n_dev) n = *n_dev;
    const long long b0 = (long long)blockIdx.x * SCAN_ITEMS + threadIdx.x * 4;
    int x[4], v = 0;
    for (int k = 0; k < 4; ++k) {
        x[k] = b0 + k < n ? in[b0 + k] : 0;
        v += x[k];
    }
    int total;
    int run = sums[blockIdx.x] + block_scan_256(v, s_tmp, total);
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = 0;
    for (int k = 0; k < 4; ++k) {
        run += x[k];
        if (b0 + k < n) {
            out[b0 + k + 1] = run;
            if (total_out && b0 + k == n - 1) *total_out = run;   // the grand total, where the caller wants it
        }
    }
    if (total_out && n <= 0 && blockIdx.x == 0 && threadIdx.x == 0) *total_out = 0;
}

// A short list in ONE launch: a single workgroup walks it 1024 entries at a time with a running carry
// (out[i] = sum of in[0..i), out[n] = the total). The three-launch form costs 14 us of dispatch whatever n is;
// up to SCAN_SMALL entries one workgroup is faster than that.
#define SCAN_SMALL 32768
__global__ void __launch_bounds__(1024) lidf_scan_small_kernel(const int* __restrict__ in, int n,
                                                               int* __restrict__ out) {
    __shared__ int s_w[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int carry = 0;
    for (int b = 0; b < n; b += 4096) {   // four consecutive entries per thread and round
        const int k = b + 4 * threadIdx.x;
        int c[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = k + j < n ? in[k + j] : 0;
        const int mine = c[0] + c[1] + c[2] + c[3];
        int inc = mine;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const int o = __shfl_up(inc, sft);
            if (lane >= sft) inc += o;
        }
        __syncthreads();
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        int wpre = 0, tot = 0;
        for (int w = 0; w < 16; ++w) {
            const int t = s_w[w];
            wpre += w < wave ? t : 0;
            tot += t;
        }
        int run = carry + wpre + inc - mine;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (k + j < n) out[k + j] = run;
            run += c[j];
        }
        carry += tot;
    }
    if (threadIdx.x == 0) out[n] = carry;
}

extern "C" hipError_t lidf_launch_scan(const int* in, long long n, int* out, int* sums,
                                       hipStream_t st) {
    if (n > 0 && n <= SCAN_SMALL) {
        hipLaunchKernelGGL(lidf_scan_small_kernel, dim3(1), dim3(1024), 0, st, in, (int)n, out);
        return hipGetLastError();
    }
    if (n <= 0) {
        hipLaunchKernelGGL(lidf_scan_final_kernel, dim3(1), dim3(256), 0, st, in, (long long)0,
                           sums, out, (const int*)nullptr, (int*)nullptr);
        return hipGetLastError();
    }
    const long long nb = (n + SCAN_ITEMS - 1) / SCAN_ITEMS;
    hipLaunchKernelGGL(lidf_scan_sums_kernel, dim3((unsigned)nb), dim3(256), 0, st, in, n, sums,
                       (const int*)nullptr);
    hipLaunchKernelGGL(lidf_scan_top_kernel, dim3(1), dim3(256), 0, st, sums, nb);
    hipLaunchKernelGGL(lidf_scan_final_kernel, dim3((unsigned)nb), dim3(256), 0, st, in, n, sums,
                       out, (const int*)nullptr, (int*)nullptr);
    return hipGetLastError();
}

// The same scan over the first *n_dev entries of a buffer of capacity n_cap (the sync-free frame
// path): the launch is sized for n_cap, entries beyond *n_dev are neither read nor written, and the
// grand total out[*n_dev] is also stored to *total_out when given.
extern "C" hipError_t lidf_launch_scan_dev(const int* in, long long n_cap, const int* n_dev, int* out,
                                           int* sums, int* total_out, hipStream_t st) {
    if (n_cap <= 0) n_cap = 1;
    const long long nb = (n_cap + SCAN_ITEMS - 1) / SCAN_ITEMS;
    hipLaunchKernelGGL(lidf_scan_sums_kernel, dim3((unsigned)nb), dim3(256), 0, st, in, n_cap, sums, n_dev);
    hipLaunchKernelGGL(lidf_scan_top_kernel, dim3(1), dim3(256), 0, st, sums, nb);
    hipLaunchKernelGGL(lidf_scan_final_kernel, dim3((unsigned)nb), dim3(256), 0, st, in, n_cap, sums,
                       out, n_dev, total_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Occupied-voxel build — utils/point_utils.py:12-76 batch_get_occupied_idx (overlap=False) +
// LIDF.get_occ_vox_bound (models/pipeline.py:162-201). The grid is tiny (B x rx x ry x rz cells),
// so torch.unique(dim=0, return_inverse=True) — which sorts the (bid, x, y, z) rows — is a dense
// mark -> exclusive scan -> compact: the scan order over cell keys IS the lexicographic order.
//   pass A (points): cell of every point, validity, mark the cell, per-point key (-1 = outside)
//   scans: cells -> voxel rank ; points -> rank among the valid points (order preserved)
//   pass B (cells): occ_bid_coord [V,4] i32, voxel_bound [V,6]
//   pass C (points): valid_v_pid, revidx, valid_v_rel_coord
// f32 arithmetic follows the reference op by op: v - xmin ; / crop ; floor ; coord*crop + crop/2.
// ------------------------------------------------------------------------------------------------

__global__ void lidf_vox_mark_kernel(const float* __restrict__ xyz, const int* __restrict__ bid,
                                     long long N, GridSpec g, int* __restrict__ cell_flag,
                                     int* __restrict__ pt_key, int* __restrict__ pt_valid) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    int c[3];
    bool ok = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float v = xyz[3 * i + a] - g.xmin[a];
        const float q = floorf(v / g.crop);
        ok = ok && (q >= 0.f) && (q < (float)g.r[a]);
        c[a] = (int)q;
    }
    const int b = bid[i];
    ok = ok && b >= 0 && b < g.B;
    int key = -1;
    if (ok) {
        key = ((b * g.r[0] + c[0]) * g.r[1] + c[1]) * g.r[2] + c[2];
        cell_flag[key] = 1;
    }
    pt_key[i] = key;
    pt_valid[i] = ok ? 1 : 0;
}

__global__ void lidf_vox_cells_kernel(const int* __restrict__ cell_flag,
                                      const int* __restrict__ cell_rank, long long ncell, GridSpec g,
                                      int* __restrict__ occ, float* __restrict__ vbound,
                                      int* __restrict__ vox_bid) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ncell || !cell_flag[k]) return;
    const int v = cell_rank[k];
    int rem = (int)k;
    const int cz = rem % g.r[2]; rem /= g.r[2];
    const int cy = rem % g.r[1]; rem /= g.r[1];
    const int cx = rem % g.r[0]; rem /= g.r[0];
    occ[4 * v + 0] = rem;
    if (vox_bid) vox_bid[v] = rem;   // the image index on its own ([V] i32: what the box tests take)
    occ[4 * v + 1] = cx;
    occ[4 * v + 2] = cy;
    occ[4 * v + 3] = cz;
    const int c[3] = {cx, cy, cz};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float lo = g.xmin[a] + (float)c[a] * g.crop;  // pipeline.py:186
        vbound[6 * v + a] = lo;
        vbound[6 * v + 3 + a] = lo + g.crop;               // :187
    }
}

__global__ void lidf_vox_points_kernel(const float* __restrict__ xyz,
                                       const int* __restrict__ pt_key,
                                       const int* __restrict__ pt_rank,
                                       const int* __restrict__ cell_rank, long long N, GridSpec g,
                                       int* __restrict__ pid, int* __restrict__ revidx,
                                       float* __restrict__ rel) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int key = pt_key[i];
    if (key < 0) return;
    const int j = pt_rank[i];
    pid[j] = (int)i;
    revidx[j] = cell_rank[key];
    int rem = key;
    int c[3];
    c[2] = rem % g.r[2]; rem /= g.r[2];
    c[1] = rem % g.r[1]; rem /= g.r[1];
    c[0] = rem % g.r[0];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float v = xyz[3 * i + a] - g.xmin[a];
        const float centre = (float)c[a] * g.crop + 0.5f * g.crop;  // point_utils.py:50
        rel[3 * j + a] = v - centre;                                // :51
    }
}

extern "C" hipError_t lidf_launch_vox_mark(const float* xyz, const int* bid, long long N,
                                           const GridSpec& g, int* cell_flag, int* pt_key,
                                           int* pt_valid, hipStream_t st) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_vox_mark_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st,
                       xyz, bid, N, g, cell_flag, pt_key, pt_valid);
    return hipGetLastError();
}
extern "C" hipError_t lidf_launch_vox_cells_bid(const int* cell_flag, const int* cell_rank,
                                                long long ncell, const GridSpec& g, int* occ,
                                                float* vbound, int* vox_bid, hipStream_t st) {
    if (ncell <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_vox_cells_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0,
                       st, cell_flag, cell_rank, ncell, g, occ, vbound, vox_bid);
    return hipGetLastError();
}
extern "C" hipError_t lidf_launch_vox_cells(const int* cell_flag, const int* cell_rank,
                                            long long ncell, const GridSpec& g, int* occ,
                                            float* vbound, hipStream_t st) {
    return lidf_launch_vox_cells_bid(cell_flag, cell_rank, ncell, g, occ, vbound, nullptr, st);
}
extern "C" hipError_t lidf_launch_vox_points(const float* xyz, const int* pt_key,
                                             const int* pt_rank, const int* cell_rank, long long N,
                                             const GridSpec& g, int* pid, int* revidx, float* rel,
                                             hipStream_t st) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_vox_points_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st,
                       xyz, pt_key, pt_rank, cell_rank, N, g, pid, revidx, rel);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Eval depth metrics (models/pipeline.py:577-627): nearest-neighbour resize of the predicted and
// ground-truth depth maps and of the segmentation mask to dst_h x dst_w (cv2.resize
// INTER_NEAREST: source index = min(floor(dst index * src/dst), src - 1), the factor formed in
// double as 1/(dst/src)), non-finite ground truth -> 0, valid = gt > 0 and mask != 0, then the
// nine ClearGrasp statistics over the valid pixels. One workgroup: the resized image has 36,864
// pixels; sums are carried in double.
// ------------------------------------------------------------------------------------------------
#define METRIC_SUMS 10
#define METRIC_MAX_WGS 64
// workspace (lidf_depth_metrics_workspace_bytes(), zero-filled ONCE by the caller; a call leaves it as it
// found it): [0] ticket, then METRIC_MAX_WGS x METRIC_SUMS doubles of per-workgroup partial sums.
// seg_dtype: 0 no mask, 1 uint8 / bool, 2 float32 (the reference's corrupt_mask: `astype(np.uint8)`,
// pipeline.py:588 — a C cast, truncation toward zero, modulo 256).
__device__ __forceinline__ bool metric_mask(const void* seg, int dtype, size_t i) {
    if (dtype == 1) return ((const unsigned char*)seg)[i] != 0;
    if (dtype == 2) {
        const float m = ((const float*)seg)[i];
        return m == m && (((long long)m) & 255) != 0;
    }
    return true;
}
// The 144 x 256 statistics image is 36,864 pixels of f64 arithmetic: one workgroup (rounds 2-3) took
// 36 us; here every workgroup of 1024 threads sums its share in the strided order of the one-workgroup
// kernel, leaves ten partial sums, and the workgroup that arrives last adds the partials in workgroup
// order (a fixed order: run-to-run identical) and forms the statistics.
__global__ void __launch_bounds__(1024) lidf_depth_metrics_kernel(
    const float* __restrict__ pred, const float* __restrict__ gt, const void* __restrict__ seg, int seg_dtype,
    int src_h, int src_w, int dst_h, int dst_w, float* __restrict__ out, unsigned* __restrict__ ticket,
    unsigned long long* __restrict__ partial) {
    __shared__ double red[METRIC_SUMS][16];
    __shared__ bool s_last;
    double acc[METRIC_SUMS];
#pragma unroll
    for (int i = 0; i < METRIC_SUMS; ++i) acc[i] = 0.0;
    const double ifx = 1.0 / ((double)dst_w / (double)src_w);
    const double ify = 1.0 / ((double)dst_h / (double)src_h);
    const int n = dst_h * dst_w;
    const int stride = (int)(gridDim.x * blockDim.x);
    for (int i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 4 * stride) {
        float gv[4], pv[4];
        bool mv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + k * stride;
            gv[k] = 0.f;
            pv[k] = 1.f;
            mv[k] = false;
            if (i < n) {
                const int dy = i / dst_w, dx = i % dst_w;
                int sx = (int)floor(dx * ifx), sy = (int)floor(dy * ify);
                sx = sx < src_w - 1 ? sx : src_w - 1;
                sy = sy < src_h - 1 ? sy : src_h - 1;
                const size_t si = (size_t)sy * src_w + sx;
                gv[k] = gt[si];
                pv[k] = pred[si];
                mv[k] = metric_mask(seg, seg_dtype, si);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float g = gv[k];
            if (isnan(g) || isinf(g)) g = 0.f;
            if (!(g > 0.f) || !mv[k]) continue;
            const float p = pv[k];
            // same f32 operations as the torch expressions (pipeline.py:612-623)
            const float thresh = fmaxf(g / p, p / g);
            const float d = g - p;
            const float lg = logf(fminf(fmaxf(g, 1e-6f), 1e6f)), lp = logf(fminf(fmaxf(p, 1e-6f), 1e6f));
            acc[0] += thresh < 1.05f ? 1.0 : 0.0;
            acc[1] += thresh < 1.10f ? 1.0 : 0.0;
            acc[2] += thresh < 1.25f ? 1.0 : 0.0;
            acc[3] += (double)(d * d);
            acc[4] += (double)((lg - lp) * (lg - lp));
            acc[5] += (double)fabsf(lg - lp);
            acc[6] += (double)(fabsf(d) / g);
            acc[7] += (double)fabsf(d);
            acc[8] += (double)(d * d / g);
            acc[9] += 1.0;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < METRIC_SUMS; ++i) {
        double v = acc[i];
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[i][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < METRIC_SUMS) {
        double sum = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) sum += red[threadIdx.x][w];
        // (an exchange: its returned value arrives once the write is performed at the coherent level, so the
        // ticket below is taken after the partials are visible — without a release fence, which on this
        // device writes back the whole L2, megabytes of the frame's outputs)
        const unsigned long long was = __hip_atomic_exchange(
            partial + (size_t)blockIdx.x * METRIC_SUMS + threadIdx.x, (unsigned long long)__double_as_longlong(sum),
            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" ::"v"(was));
    }
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;
    // (every partial is requested by a thread of its own — one memory round trip —, then added in
    // workgroup order)
    __shared__ double s_part[METRIC_MAX_WGS * METRIC_SUMS];
    if (threadIdx.x < gridDim.x * METRIC_SUMS)
        s_part[threadIdx.x] = __longlong_as_double((long long)__hip_atomic_load(
            partial + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    __syncthreads();
    if (threadIdx.x < METRIC_SUMS) {
        double sum = 0.0;
        for (unsigned g = 0; g < gridDim.x; ++g) sum += s_part[g * METRIC_SUMS + threadIdx.x];
        red[threadIdx.x][0] = sum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s[METRIC_SUMS];
        for (int i = 0; i < METRIC_SUMS; ++i) s[i] = red[i][0];
        const double c = s[9];  // 0 valid pixels: torch's mean of an empty tensor is NaN, so is 0/0
        out[0] = (float)(s[0] / c);
        out[1] = (float)(s[1] / c);
        out[2] = (float)(s[2] / c);
        out[3] = (float)sqrt(s[3] / c);
        out[4] = (float)sqrt(s[4] / c);
        out[5] = (float)(s[5] / c);
        out[6] = (float)(s[6] / c);
        out[7] = (float)(s[7] / c);
        out[8] = (float)(s[8] / c);
        out[9] = (float)c;
        *ticket = 0;   // the workspace is left as it was found
    }
}

extern "C" size_t lidf_depth_metrics_ws_bytes(void) { return 64 + (size_t)METRIC_MAX_WGS * METRIC_SUMS * 8; }
extern "C" hipError_t lidf_launch_depth_metrics(const float* pred, const float* gt, const void* seg,
                                                int seg_dtype, int src_h, int src_w, int dst_h, int dst_w,
                                                float* out, void* ws, hipStream_t st) {
    long long g = ((long long)dst_h * dst_w + 1023) / 1024;
    if (g > METRIC_MAX_WGS) g = METRIC_MAX_WGS;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(lidf_depth_metrics_kernel, dim3((unsigned)g), dim3(1024), 0, st, pred, gt, seg, seg_dtype,
                       src_h, src_w, dst_h, dst_w, out, (unsigned*)ws, (unsigned long long*)((char*)ws + 64));
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// get_miss_ray — models/pipeline.py:203-269 (eval flavour): miss_idx = nonzero(mask.view(bs,-1))
// in row-major (image, pixel) order, then per selected pixel the image index, the flat pixel index,
// the unit ray direction (same expression as lidf_ray_dirs_kernel, formed here for the selected
// pixels only: no dense [bs,h*w,3] field) and the integer pixel (x, y).
//   pass 1: per block of 1024 pixels the number of non-zero mask entries
//   scan of the block counts (lidf_launch_scan), total -> n_rays
//   pass 2: the block recomputes its flags, ranks them with a block scan (ascending pixel order
//           inside a thread's 4 consecutive pixels and across threads) and writes the rays
// mask element types as torch.nonzero accepts them: f32 (the reference's corrupt_mask / pred_mask
// are float tensors, datasets/cleargrasp_dataset.py:114), u8/bool, i32, i64. NaN counts as non-zero.
// ------------------------------------------------------------------------------------------------
#define MISS_ITEMS 1024

__device__ __forceinline__ bool mask_nonzero(const void* mask, int dtype, long long i) {
    switch (dtype) {
        case 0: return ((const float*)mask)[i] != 0.f;
        case 1: return ((const unsigned char*)mask)[i] != 0;
        case 2: return ((const int*)mask)[i] != 0;
        default: return ((const long long*)mask)[i] != 0;
    }
}

__global__ void lidf_miss_count_kernel(const void* __restrict__ mask, int dtype, long long n,
                                       int* __restrict__ block_cnt) {
    __shared__ int s_tmp[4];
    const long long b0 = (long long)blockIdx.x * MISS_ITEMS + threadIdx.x * 4;
    int v = 0;
    for (int k = 0; k < 4; ++k)
        if (b0 + k < n && mask_nonzero(mask, dtype, b0 + k)) ++v;
    int total;
    block_scan_256(v, s_tmp, total);
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = total;
}

__global__ void lidf_miss_total_kernel(const int* __restrict__ block_off, long long nb,
                                       int* __restrict__ n_rays) {
    if (threadIdx.x == 0 && blockIdx.x == 0) n_rays[0] = block_off[nb];
}

__global__ void lidf_miss_fill_kernel(const void* __restrict__ mask, int dtype, long long n,
                                      const int* __restrict__ block_off,
                                      const float* __restrict__ intr, int H, int W,
                                      int* __restrict__ ray_bid, int* __restrict__ ray_flat,
                                      int* __restrict__ ray_pix, float* __restrict__ ray_dir,
                                      long long* __restrict__ bid64, long long* __restrict__ flat64,
                                      long long* __restrict__ pix64) {
    __shared__ int s_tmp[4];
    const long long b0 = (long long)blockIdx.x * MISS_ITEMS + threadIdx.x * 4;
    bool f[4];
    int v = 0;
    for (int k = 0; k < 4; ++k) {
        f[k] = b0 + k < n && mask_nonzero(mask, dtype, b0 + k);
        v += f[k] ? 1 : 0;
    }
    int total;
    int j = block_off[blockIdx.x] + block_scan_256(v, s_tmp, total);
    const long long hw = (long long)H * W;
    for (int k = 0; k < 4; ++k) {
        if (!f[k]) continue;
        const long long i = b0 + k;
        const int b = (int)(i / hw);
        const int rem = (int)(i % hw);
        const int y = rem / W, x = rem % W;
        if (ray_bid) ray_bid[j] = b;
        if (ray_flat) ray_flat[j] = rem;
        if (ray_pix) {
            ray_pix[2 * j] = x;
            ray_pix[2 * j + 1] = y;
        }
        if (bid64) bid64[j] = b;
        if (flat64) flat64[j] = rem;
        if (pix64) {
            pix64[2 * j] = x;
            pix64[2 * j + 1] = y;
        }
        if (ray_dir) {  // pipeline.py:215-219, as lidf_ray_dirs_kernel
            const float fx = intr[4 * b], fy = intr[4 * b + 1], cx = intr[4 * b + 2],
                        cy = intr[4 * b + 3];
            const float vx = (float)x - cx;
            const float vy = ((float)y - cy) * fx / fy;
            const float vz = fx;
            const float nrm = sqrtf(vx * vx + vy * vy + vz * vz);
            ray_dir[3 * j] = vx / nrm;
            ray_dir[3 * j + 1] = vy / nrm;
            ray_dir[3 * j + 2] = vz / nrm;
        }
        ++j;
    }
}

extern "C" hipError_t lidf_launch_miss_count(const void* mask, int dtype, long long n,
                                             int* block_cnt, int* block_off, int* sums,
                                             int* n_rays, hipStream_t st) {
    const long long nb = (n + MISS_ITEMS - 1) / MISS_ITEMS;
    if (nb > 0)
        hipLaunchKernelGGL(lidf_miss_count_kernel, dim3((unsigned)nb), dim3(256), 0, st, mask, dtype,
                           n, block_cnt);
    hipError_t e = lidf_launch_scan(block_cnt, nb, block_off, sums, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(lidf_miss_total_kernel, dim3(1), dim3(64), 0, st, block_off, nb, n_rays);
    return hipGetLastError();
}

extern "C" hipError_t lidf_launch_miss_fill(const void* mask, int dtype, long long n,
                                            const int* block_off, const float* intr, int H, int W,
                                            int* ray_bid, int* ray_flat, int* ray_pix,
                                            float* ray_dir, long long* bid64, long long* flat64,
                                            long long* pix64, hipStream_t st) {
    const long long nb = (n + MISS_ITEMS - 1) / MISS_ITEMS;
    if (nb <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_miss_fill_kernel, dim3((unsigned)nb), dim3(256), 0, st, mask, dtype, n,
                       block_off, intr, H, W, ray_bid, ray_flat, ray_pix, ray_dir, bid64, flat64,
                       pix64);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Guarded packing: fingerprint of the raw parameter buffers, compared on the device
// ------------------------------------------------------------------------------------------------
// A module's packed weight streams are valid as long as the CONTENTS of its nn.Parameter storage do
// not change. The host cannot see that (torch's version counter misses `p.data.mul_()`; SURVEY §8b
// "Ownership"), so every guarded call hashes the buffers on the device: word i of the concatenated
// buffers contributes splitmix64(i << 32 | word) to a 64-bit sum (order-independent, so blocks add
// their partial sums with one atomic each). The block that finishes last compares the sum (+ a salt
// of the host-side scalars that are baked into the streams) with the fingerprint the streams were
// built from, leaves the verdict in guard->dirty for the pack kernels that follow on the stream and
// re-arms the accumulator. ~1.1 MB of parameters: one launch of a few microseconds, no host sync.
struct FpSegs {
    const unsigned* p[LIDF_FP_MAX_SEGS];
    unsigned n[LIDF_FP_MAX_SEGS];      // words
    unsigned base[LIDF_FP_MAX_SEGS];   // index of the segment's first word in the concatenation
    unsigned blk0[LIDF_FP_MAX_SEGS + 1];  // first workgroup of the segment (FP_WORDS words per workgroup)
    int nseg;
};

__device__ __forceinline__ unsigned long long fp_mix(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

#define FP_WORDS 4096   // per workgroup: 256 threads x 16 words, four loads in flight per thread

__global__ void __launch_bounds__(256) lidf_fingerprint_kernel(FpSegs s, unsigned long long salt,
                                                               LidfPackGuardState* g) {
    int seg = 0;
    for (int k = 1; k < LIDF_FP_MAX_SEGS; ++k) seg += (k < s.nseg && blockIdx.x >= s.blk0[k]) ? 1 : 0;
    const unsigned n = s.n[seg], base = s.base[seg];
    const unsigned* __restrict__ p = s.p[seg];
    const unsigned w0 = (blockIdx.x - s.blk0[seg]) * FP_WORDS;
    unsigned long long acc = 0;
#pragma unroll
    for (int r = 0; r < FP_WORDS / 1024; ++r) {
        unsigned w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned i = w0 + (4 * r + j) * 256 + threadIdx.x;
            w[j] = i < n ? p[i] : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned i = w0 + (4 * r + j) * 256 + threadIdx.x;
            if (i < n) acc += fp_mix(((unsigned long long)(base + i) << 32) | w[j]);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)acc, o), hi = __shfl_xor((unsigned)(acc >> 32), o);
        acc += ((unsigned long long)hi << 32) | lo;
    }
    __shared__ unsigned long long part[4];
    __shared__ bool last;
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        // (the returned value orders the ticket behind the sum without a release fence = an L2 write-back)
        const unsigned long long was = atomicAdd(&g->acc, part[0] + part[1] + part[2] + part[3]);
        asm volatile("" ::"v"(was));
        last = atomicAdd(&g->ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        const unsigned long long h = atomicAdd(&g->acc, 0ull) + salt;
        g->dirty = (!g->valid || g->hash != h) ? 1 : 0;
        g->hash = h;
        g->valid = 1;
        g->acc = 0;
        g->ticket = 0;
    }
}

extern "C" hipError_t lidf_launch_fingerprint(const float* const* ptrs, const long long* floats,
                                              int nseg, unsigned long long salt,
                                              LidfPackGuardState* guard, hipStream_t st) {
    if (nseg <= 0 || nseg > LIDF_FP_MAX_SEGS) return hipErrorInvalidValue;
    FpSegs s = {};
    unsigned base = 0, blk = 0;
    for (int i = 0; i < nseg; ++i) {
        s.p[i] = (const unsigned*)ptrs[i];
        s.n[i] = (unsigned)floats[i];
        s.base[i] = base;
        s.blk0[i] = blk;
        base += s.n[i];
        blk += (s.n[i] + FP_WORDS - 1) / FP_WORDS;
    }
    s.blk0[nseg] = blk;
    s.nseg = nseg;
    if (blk == 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(lidf_fingerprint_kernel, dim3(blk), dim3(256), 0, st, s, salt, guard);
    return hipGetLastError();
}

// The same fingerprint over several module groups in ONE launch (the frame path: query decoders,
// PointNet, stage-2 PointNet, stage-2 decoder — four launches of ~11 us each before): segment i belongs
// to group grp[i] and is hashed with its word index inside the group; every group has its own guard,
// salt and block count, and the block of a group that finishes last leaves that group's verdict.
struct FpSegsM {
    const unsigned* p[LIDF_FP_MULTI_SEGS];
    unsigned n[LIDF_FP_MULTI_SEGS];        // words
    unsigned base[LIDF_FP_MULTI_SEGS];     // index of the segment's first word in its group's concatenation
    unsigned blk0[LIDF_FP_MULTI_SEGS + 1]; // first workgroup of the segment
    unsigned char grp[LIDF_FP_MULTI_SEGS];
    unsigned gblocks[LIDF_FP_GROUPS];      // workgroups of the group
    unsigned long long salt[LIDF_FP_GROUPS];
    int nseg;
};

__global__ void __launch_bounds__(256) lidf_fingerprint_multi_kernel(FpSegsM s, LidfPackGuardState* guards,
                                                                     int guard_stride) {
    int seg = 0;
    for (int k = 1; k < LIDF_FP_MULTI_SEGS; ++k) seg += (k < s.nseg && blockIdx.x >= s.blk0[k]) ? 1 : 0;
    const unsigned n = s.n[seg], base = s.base[seg];
    const int grp = s.grp[seg];
    LidfPackGuardState* g = (LidfPackGuardState*)((char*)guards + (size_t)grp * guard_stride);
    const unsigned* __restrict__ p = s.p[seg];
    const unsigned w0 = (blockIdx.x - s.blk0[seg]) * FP_WORDS;
    unsigned long long acc = 0;
#pragma unroll
    for (int r = 0; r < FP_WORDS / 1024; ++r) {
        unsigned w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned i = w0 + (4 * r + j) * 256 + threadIdx.x;
            w[j] = i < n ? p[i] : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned i = w0 + (4 * r + j) * 256 + threadIdx.x;
            if (i < n) acc += fp_mix(((unsigned long long)(base + i) << 32) | w[j]);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)acc, o), hi = __shfl_xor((unsigned)(acc >> 32), o);
        acc += ((unsigned long long)hi << 32) | lo;
    }
    __shared__ unsigned long long part[4];
    __shared__ bool last;
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long was = atomicAdd(&g->acc, part[0] + part[1] + part[2] + part[3]);
        asm volatile("" ::"v"(was));
        last = atomicAdd(&g->ticket, 1u) == s.gblocks[grp] - 1;
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        const unsigned long long h = atomicAdd(&g->acc, 0ull) + s.salt[grp];
        g->dirty = (!g->valid || g->hash != h) ? 1 : 0;
        g->hash = h;
        g->valid = 1;
        g->acc = 0;
        g->ticket = 0;
    }
}

// segments sorted by group (grp ascending, every group non-empty); guards: ngrp states guard_stride bytes apart
extern "C" hipError_t lidf_launch_fingerprint_multi(const float* const* ptrs, const long long* floats,
                                                    const int* grp, int nseg,
                                                    const unsigned long long* salts, int ngrp,
                                                    void* guards, int guard_stride, hipStream_t st) {
    if (nseg <= 0 || nseg > LIDF_FP_MULTI_SEGS || ngrp <= 0 || ngrp > LIDF_FP_GROUPS) return hipErrorInvalidValue;
    FpSegsM s = {};
    unsigned blk = 0;
    unsigned base[LIDF_FP_GROUPS] = {};
    for (int i = 0; i < nseg; ++i) {
        const int g = grp[i];
        if (g < 0 || g >= ngrp || (i > 0 && g < grp[i - 1])) return hipErrorInvalidValue;
        s.p[i] = (const unsigned*)ptrs[i];
        s.n[i] = (unsigned)floats[i];
        s.base[i] = base[g];
        s.blk0[i] = blk;
        s.grp[i] = (unsigned char)g;
        base[g] += s.n[i];
        const unsigned nb = (s.n[i] + FP_WORDS - 1) / FP_WORDS;
        blk += nb;
        s.gblocks[g] += nb;
    }
    s.blk0[nseg] = blk;
    s.nseg = nseg;
    for (int g = 0; g < ngrp; ++g) {
        if (s.gblocks[g] == 0) return hipErrorInvalidValue;
        s.salt[g] = salts[g];
    }
    hipLaunchKernelGGL(lidf_fingerprint_multi_kernel, dim3(blk), dim3(256), 0, st, s,
                       (LidfPackGuardState*)guards, guard_stride);
    return hipGetLastError();
}
