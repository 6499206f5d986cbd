// lidf_frame.hip — the head of the evaluation path of one batch of frames as a few pixel-parallel
// launches whose sizes never reach the host (the sync-free frame path, lidf_frame_f32):
//   LIDF.prepare_data / get_valid_points   models/pipeline.py:91-160   (valid pixels -> points)
//   LIDF.get_occ_vox_bound                 models/pipeline.py:162-201  (points -> occupied voxels)
//   LIDF.get_miss_ray                      models/pipeline.py:203-269  (mask -> rays)
//   the PointNet input rows                models/pipeline.py:399-408
// The reference compacts with torch.nonzero / torch.unique, each of which sizes its output on the
// host (a device -> host round trip that idles the GPU four times per frame). Here every list is
// written into a buffer sized for the worst case (every pixel valid / every pixel queried) and its
// length goes to `counts` on the device, where the following launches read it.
//
// counts (int32[LIDF_FRAME_COUNTS]): [0] R rays, [1] P pairs, [2] V occupied voxels, [3] NV valid
// points inside the grid, [4] NV0 valid pixels, [5] NVS valid points kept by the stride, [6] NV + R,
// [7] overflow flags (bit 0: more pairs than max_pairs — the list was cut).
// Built with -ffp-contract=off: the f32 expressions follow the reference op by op.
#include "lidf_device.h"

#define FRAME_ITEMS 1024   // pixels per workgroup: 256 threads x 4 consecutive pixels

// pass 1: per workgroup the number of valid pixels and of queried pixels
__global__ void __launch_bounds__(256) lidf_frame_count_kernel(const float* __restrict__ valid_mask,
                                                               const float* __restrict__ miss_mask,
                                                               long long npix, int* __restrict__ blk_valid,
                                                               int* __restrict__ blk_miss) {
    __shared__ int s_tmp[4];
    const long long b0 = (long long)blockIdx.x * FRAME_ITEMS + threadIdx.x * 4;
    int nv = 0, nm = 0;
    for (int k = 0; k < 4; ++k) {
        if (b0 + k >= npix) break;
        nv += valid_mask[b0 + k] != 0.f ? 1 : 0;                 // torch.nonzero: NaN counts, -0.0 does not
        nm += (!miss_mask || miss_mask[b0 + k] != 0.f) ? 1 : 0;
    }
    int tv, tm;
    block_scan_256(nv, s_tmp, tv);
    block_scan_256(nm, s_tmp, tm);
    if (threadIdx.x == 0) {
        blk_valid[blockIdx.x] = tv;
        blk_miss[blockIdx.x] = tm;
    }
}

// pass 2 (one workgroup): exclusive scans of the two count arrays in place; list lengths to `counts`
// n_list > 0: the valid points come as an explicit list (lidf_frame_valid_list_kernel): NV0 = NVS = n_list
__global__ void __launch_bounds__(256) lidf_frame_offsets_kernel(int* __restrict__ blk_valid,
                                                                 int* __restrict__ blk_miss, int nb,
                                                                 int stride, int n_list,
                                                                 int* __restrict__ counts) {
    __shared__ int s_tmp[4];
    int cv = 0, cm = 0;
    for (int b = 0; b < nb; b += 256) {
        const int i = b + threadIdx.x;
        const int v = i < nb ? blk_valid[i] : 0, m = i < nb ? blk_miss[i] : 0;
        int tv, tm;
        const int ev = block_scan_256(v, s_tmp, tv);
        const int em = block_scan_256(m, s_tmp, tm);
        if (i < nb) {
            blk_valid[i] = cv + ev;
            blk_miss[i] = cm + em;
        }
        cv += tv;
        cm += tm;
    }
    if (threadIdx.x == 0) {
        counts[0] = cm;                                   // R
        counts[4] = n_list > 0 ? n_list : cv;                                   // NV0
        counts[5] = n_list > 0 ? n_list : (cv + stride - 1) / stride;           // NVS: valid_idx[::stride]
        counts[7] = 0;
    }
}

// Selected valid point j = pixel `rem` of image b: its list entries and its cell of the voxel grid
// (batch_get_occupied_idx, utils/point_utils.py:12-76, pass A)
__device__ __forceinline__ void frame_valid_point(int j, int b, int rem, const float* px, const float* pc,
                                                  long long hw, const GridSpec& g, int* valid_bid,
                                                  int* valid_flat, float* valid_xyz, float* valid_rgb,
                                                  int* cell_flag, int* pt_key, int* pt_valid) {
    const float p[3] = {px[0], px[hw], px[2 * hw]};
    valid_bid[j] = b;
    valid_flat[j] = rem;
    int c[3];
    bool ok = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        valid_xyz[3 * j + a] = p[a];
        valid_rgb[3 * j + a] = pc[a * hw];
        const float v = p[a] - g.xmin[a];
        const float q = floorf(v / g.crop);
        ok = ok && (q >= 0.f) && (q < (float)g.r[a]);
        c[a] = (int)q;
    }
    int key = -1;
    if (ok) {
        key = ((b * g.r[0] + c[0]) * g.r[1] + c[1]) * g.r[2] + c[2];
        cell_flag[key] = 1;
    }
    pt_key[j] = key;
    pt_valid[j] = ok ? 1 : 0;
}

// The valid points as an explicit list (LidfFrameArgs.valid_idx_*: the reference's sampled valid_idx,
// pipeline.py:143-158): one thread per entry, in the list's order.
__global__ void __launch_bounds__(256) lidf_frame_valid_list_kernel(
    const int* __restrict__ idx_bid, const int* __restrict__ idx_flat, long long n,
    const float* __restrict__ xyz, const float* __restrict__ rgb, int B, long long hw, GridSpec g,
    int* __restrict__ valid_bid, int* __restrict__ valid_flat, float* __restrict__ valid_xyz,
    float* __restrict__ valid_rgb, int* __restrict__ cell_flag, int* __restrict__ pt_key,
    int* __restrict__ pt_valid) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    int b = idx_bid[j], rem = idx_flat[j];
    // (indices outside the batch are clamped: the reference would raise in index_select)
    b = b < 0 ? 0 : (b >= B ? B - 1 : b);
    rem = rem < 0 ? 0 : (rem >= hw ? (int)hw - 1 : rem);
    frame_valid_point((int)j, b, rem, xyz + (size_t)b * 3 * hw + rem, rgb + (size_t)b * 3 * hw + rem, hw, g,
                      valid_bid, valid_flat, valid_xyz, valid_rgb, cell_flag, pt_key, pt_valid);
}

// pass 3: every list of the frame head in one sweep over the pixels.
//   valid pixel of rank i (image-major pixel order = torch.nonzero's), i % stride == 0, j = i / stride:
//     valid_bid / valid_flat [j], valid_xyz [j,3] = xyz_corrupt[b,:,pix], valid_rgb [j,3] = rgb[b,:,pix]
//     (pipeline.py:144-158) and the cell of the point: pt_key [j] (-1 outside the grid), pt_valid [j],
//     cell_flag[key] = 1   (batch_get_occupied_idx, utils/point_utils.py:12-76, pass A)
//   queried pixel of rank r: ray_bid / ray_flat / ray_pix / ray_dir [r] (pipeline.py:208-269)
//   every pixel: depth[pix] = xyz_corrupt z (the map the predictions are written into, :593-596)
__global__ void __launch_bounds__(256) lidf_frame_fill_kernel(
    const float* __restrict__ valid_mask, const float* __restrict__ miss_mask,
    const float* __restrict__ xyz, const float* __restrict__ rgb, const float* __restrict__ intr,
    long long npix, int H, int W, int stride, GridSpec g, const int* __restrict__ blk_valid,
    const int* __restrict__ blk_miss, int* __restrict__ valid_bid, int* __restrict__ valid_flat,
    float* __restrict__ valid_xyz, float* __restrict__ valid_rgb, int* __restrict__ cell_flag,
    int* __restrict__ pt_key, int* __restrict__ pt_valid, int* __restrict__ ray_bid,
    int* __restrict__ ray_flat, int* __restrict__ ray_pix, float* __restrict__ ray_dir,
    float* __restrict__ depth, float* __restrict__ depth2, int list_mode) {
    __shared__ int s_tmp[4];
    const long long b0 = (long long)blockIdx.x * FRAME_ITEMS + threadIdx.x * 4;
    bool fv[4], fm[4];
    int nv = 0, nm = 0;
    for (int k = 0; k < 4; ++k) {
        const bool in = b0 + k < npix;
        fv[k] = in && !list_mode && valid_mask[b0 + k] != 0.f;
        fm[k] = in && (!miss_mask || miss_mask[b0 + k] != 0.f);
        nv += fv[k] ? 1 : 0;
        nm += fm[k] ? 1 : 0;
    }
    int tot;
    int iv = blk_valid[blockIdx.x] + block_scan_256(nv, s_tmp, tot);
    int im = blk_miss[blockIdx.x] + block_scan_256(nm, s_tmp, tot);
    const long long hw = (long long)H * W;
    for (int k = 0; k < 4; ++k) {
        const long long i = b0 + k;
        if (i >= npix) break;
        const int b = (int)(i / hw);
        const int rem = (int)(i % hw);
        const float* px = xyz + (size_t)b * 3 * hw + rem;
        const float z = px[2 * hw];
        depth[i] = z;
        if (depth2) depth2[i] = z;
        if (fv[k]) {
            if (iv % stride == 0) {
                frame_valid_point(iv / stride, b, rem, px, rgb + (size_t)b * 3 * hw + rem, hw, g, valid_bid,
                                  valid_flat, valid_xyz, valid_rgb, cell_flag, pt_key, pt_valid);
            }
            ++iv;
        }
        if (fm[k]) {
            const int y = rem / W, x = rem % W;
            ray_bid[im] = b;
            ray_flat[im] = rem;
            ray_pix[2 * im] = x;
            ray_pix[2 * im + 1] = y;
            const float fx = intr[4 * b], fy = intr[4 * b + 1], cx = intr[4 * b + 2], cy = intr[4 * b + 3];
            const float vx = (float)x - cx;            // pipeline.py:215-219, as lidf_ray_dirs_kernel
            const float vy = ((float)y - cy) * fx / fy;
            const float vz = fx;
            const float nrm = sqrtf(vx * vx + vy * vy + vz * vz);
            ray_dir[3 * im] = vx / nrm;
            ray_dir[3 * im + 1] = vy / nrm;
            ray_dir[3 * im + 2] = vz / nrm;
            ++im;
        }
    }
}

// pass C of the voxel build over the selected valid points (utils/point_utils.py:46-60) + the
// PointNet input rows cat(valid_v_rel_coord, valid_v_rgb) (pipeline.py:399-406): point i inside the
// grid -> row j = pt_rank[i]. pnet_abs (optional) receives cat(valid_xyz[valid_v_pid], rgb), the
// refine.pnet_pos_type 'abs' input (:1001-1003).
__global__ void lidf_frame_points_kernel(const float* __restrict__ valid_xyz,
                                         const float* __restrict__ valid_rgb,
                                         const int* __restrict__ pt_key, const int* __restrict__ pt_rank,
                                         const int* __restrict__ cell_rank, GridSpec g,
                                         const int* __restrict__ counts, int* __restrict__ pid,
                                         int* __restrict__ revidx, float* __restrict__ rel,
                                         float* __restrict__ pnet_inp, float* __restrict__ pnet_abs) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= counts[5]) return;
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
        const float p = valid_xyz[3 * i + a];
        const float v = p - g.xmin[a];
        const float centre = (float)c[a] * g.crop + 0.5f * g.crop;  // point_utils.py:50
        const float r = v - centre;                                 // :51
        const float col = valid_rgb[3 * i + a];
        rel[3 * j + a] = r;
        pnet_inp[6 * j + a] = r;
        pnet_inp[6 * j + 3 + a] = col;
        if (pnet_abs) {
            pnet_abs[6 * j + a] = p;
            pnet_abs[6 * j + 3 + a] = col;
        }
    }
}

// After the pair offsets are known: cut the list at the capacity of the pair arrays (flagging it) and
// leave the derived counts. One thread per ray + 1.
__global__ void lidf_frame_pairs_kernel(int* __restrict__ pair_off, int* __restrict__ counts,
                                        long long max_pairs) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int R = counts[0];
    if (r > R) return;
    const int o = pair_off[r];
    if (o > max_pairs) pair_off[r] = (int)max_pairs;
    if (r == R) {
        counts[1] = o > max_pairs ? (int)max_pairs : o;
        if (o > max_pairs) counts[7] |= 1;
        counts[6] = counts[3] + R;
    }
}

extern "C" hipError_t lidf_launch_frame_head(const float* valid_mask, const float* miss_mask,
                                             const float* xyz, const float* rgb, const float* intr,
                                             int B, int H, int W, int stride, const GridSpec& g,
                                             int* blk_valid, int* blk_miss, int* counts, int* valid_bid,
                                             int* valid_flat, float* valid_xyz, float* valid_rgb,
                                             int* cell_flag, int* pt_key, int* pt_valid, int* ray_bid,
                                             int* ray_flat, int* ray_pix, float* ray_dir, float* depth,
                                             float* depth2, const int* idx_bid, const int* idx_flat,
                                             long long n_list, hipStream_t st) {
    const long long npix = (long long)B * H * W;
    if (npix <= 0) return hipSuccess;
    const int nb = (int)((npix + FRAME_ITEMS - 1) / FRAME_ITEMS);
    hipLaunchKernelGGL(lidf_frame_count_kernel, dim3(nb), dim3(256), 0, st, valid_mask, miss_mask, npix,
                       blk_valid, blk_miss);
    hipLaunchKernelGGL(lidf_frame_offsets_kernel, dim3(1), dim3(256), 0, st, blk_valid, blk_miss, nb, stride,
                       (int)n_list, counts);
    hipLaunchKernelGGL(lidf_frame_fill_kernel, dim3(nb), dim3(256), 0, st, valid_mask, miss_mask, xyz, rgb,
                       intr, npix, H, W, stride, g, blk_valid, blk_miss, valid_bid, valid_flat, valid_xyz,
                       valid_rgb, cell_flag, pt_key, pt_valid, ray_bid, ray_flat, ray_pix, ray_dir, depth,
                       depth2, n_list > 0 ? 1 : 0);
    if (n_list > 0)
        hipLaunchKernelGGL(lidf_frame_valid_list_kernel, dim3((unsigned)((n_list + 255) / 256)), dim3(256), 0,
                           st, idx_bid, idx_flat, n_list, xyz, rgb, B, (long long)H * W, g, valid_bid, valid_flat,
                           valid_xyz, valid_rgb, cell_flag, pt_key, pt_valid);
    return hipGetLastError();
}

extern "C" size_t lidf_frame_head_blocks(long long npix) { return (size_t)((npix + FRAME_ITEMS - 1) / FRAME_ITEMS); }

extern "C" hipError_t lidf_launch_frame_points(const float* valid_xyz, const float* valid_rgb,
                                               const int* pt_key, const int* pt_rank,
                                               const int* cell_rank, const GridSpec& g, long long cap,
                                               const int* counts, int* pid, int* revidx, float* rel,
                                               float* pnet_inp, float* pnet_abs, hipStream_t st) {
    if (cap <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_frame_points_kernel, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, st,
                       valid_xyz, valid_rgb, pt_key, pt_rank, cell_rank, g, counts, pid, revidx, rel, pnet_inp,
                       pnet_abs);
    return hipGetLastError();
}

extern "C" hipError_t lidf_launch_frame_pairs(int* pair_off, int* counts, long long R_cap,
                                              long long max_pairs, hipStream_t st) {
    hipLaunchKernelGGL(lidf_frame_pairs_kernel, dim3((unsigned)((R_cap + 1 + 255) / 256)), dim3(256), 0, st,
                       pair_off, counts, max_pairs);
    return hipGetLastError();
}

// mask_type 'all' with refine.use_all_pix == False (models/pipeline.py:987-996): a ray's predicted
// point joins the stage-2 PointNet only where the input depth was zero (inp_zero_mask = 1 - valid_mask)
__global__ void lidf_frame_select_kernel(const float* __restrict__ valid_mask,
                                         const int* __restrict__ ray_bid,
                                         const int* __restrict__ ray_flat, long long hw,
                                         const int* __restrict__ counts,
                                         unsigned char* __restrict__ sel) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= counts[0]) return;
    sel[r] = valid_mask[(size_t)ray_bid[r] * hw + ray_flat[r]] == 0.f ? 1 : 0;
}

extern "C" hipError_t lidf_launch_frame_select(const float* valid_mask, const int* ray_bid,
                                               const int* ray_flat, long long hw, long long R_cap,
                                               const int* counts, unsigned char* sel, hipStream_t st) {
    if (R_cap <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_frame_select_kernel, dim3((unsigned)((R_cap + 255) / 256)), dim3(256), 0, st,
                       valid_mask, ray_bid, ray_flat, hw, counts, sel);
    return hipGetLastError();
}
