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

// Selected valid point j = pixel `rem` of image b: its list entries and its cell of the voxel grid
// (batch_get_occupied_idx, utils/point_utils.py:12-76, pass A). Returns whether the point lies in the grid.
__device__ __forceinline__ bool frame_valid_point(int j, int b, int rem, const float* px, const float* pc,
                                                  long long hw, const GridSpec& g, int* valid_bid,
                                                  int* valid_flat, float* valid_xyz, float* valid_rgb,
                                                  int* cell_flag, int* pt_key) {
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
    return ok;
}

// The head of a frame in ONE sweep over the pixels (round 4; rounds 2-3: count -> scan -> fill -> scan of
// the in-grid flags = 3 + 3 launches). A workgroup counts the valid / queried pixels of its 1024 pixels,
// obtains the counts of the pixels before it by a decoupled look-back over the workgroups (lb_* in
// lidf_device.h) and writes every list of the frame head:
//   valid pixel of rank i (image-major pixel order = torch.nonzero's), i % stride == 0, j = i / stride:
//     valid_bid / valid_flat [j], valid_xyz [j,3] = xyz_corrupt[b,:,pix], valid_rgb [j,3] = rgb[b,:,pix]
//     (pipeline.py:144-158) and the cell of the point: pt_key [j] (-1 outside the grid), cell_flag[key] = 1
//     (batch_get_occupied_idx, utils/point_utils.py:12-76, pass A); pt_rank [j] = number of in-grid points
//     before j (a second look-back chain over the same workgroups) = the point's row of the PointNet input
//   queried pixel of rank r: ray_bid / ray_flat / ray_pix / ray_dir [r] (pipeline.py:208-269)
//   every pixel: depth[pix] = xyz_corrupt z (the map the predictions are written into, :593-596)
// n_list > 0: the valid points come as an explicit list (LidfFrameArgs.valid_idx_*: the reference's sampled
// valid_idx, pipeline.py:143-158) — workgroups nb.. of the same launch take 256 list entries each, in the
// list's order, with a look-back chain of their own for pt_rank.
// The workgroup that holds the last pixel (the last list entry) leaves the list lengths in `counts`.
struct FrameHeadArgs {
    const float *valid_mask, *miss_mask, *xyz, *rgb, *intr;
    long long npix;
    int B, H, W, stride, nb;
    GridSpec g;
    int* tickets;                       // [2] zeroed: pixel workgroups, list workgroups
    unsigned long long *stA, *stB, *stL;   // look-back status words (zeroed): [nb], [nb], [list workgroups]
    int* counts;
    int *valid_bid, *valid_flat;
    float *valid_xyz, *valid_rgb;
    int *cell_flag, *pt_key, *pt_rank;
    int *ray_bid, *ray_flat, *ray_pix;
    float *ray_dir, *depth, *depth2;
    const int *idx_bid, *idx_flat;
    long long n_list;
};

__global__ void __launch_bounds__(256) lidf_frame_head_kernel(FrameHeadArgs a) {
    __shared__ int s_tmp[4];
    __shared__ int s_bid;
    __shared__ unsigned long long s_pre;
    const int lane = threadIdx.x & 63;
    const long long hw = (long long)a.H * a.W;
    if ((int)blockIdx.x >= a.nb) {   // ---- a workgroup of the explicit valid-point list
        const int nl = (int)gridDim.x - a.nb;
        const int bid = lb_ticket(a.tickets + 1, &s_bid);
        const long long j = (long long)bid * 256 + threadIdx.x;
        bool ok = false;
        if (j < a.n_list) {
            int b = a.idx_bid[j], rem = a.idx_flat[j];
            // (indices outside the batch are clamped: the reference would raise in index_select)
            b = b < 0 ? 0 : (b >= a.B ? a.B - 1 : b);
            rem = rem < 0 ? 0 : (rem >= hw ? (int)hw - 1 : rem);
            ok = frame_valid_point((int)j, b, rem, a.xyz + (size_t)b * 3 * hw + rem,
                                   a.rgb + (size_t)b * 3 * hw + rem, hw, a.g, a.valid_bid, a.valid_flat,
                                   a.valid_xyz, a.valid_rgb, a.cell_flag, a.pt_key);
        }
        int tot;
        const int ex = block_scan_256(ok ? 1 : 0, s_tmp, tot);
        if (threadIdx.x == 0) lb_store(a.stL, bid, bid == 0 ? LIDF_LB_INC : LIDF_LB_AGG, (unsigned)tot);
        if (threadIdx.x < 64) {
            const unsigned long long pre = bid > 0 ? lb_exclusive(a.stL, bid, lane) : 0ull;
            if (lane == 0) {
                if (bid > 0) lb_store(a.stL, bid, LIDF_LB_INC, pre + (unsigned)tot);
                s_pre = pre;
            }
        }
        __syncthreads();
        const int cg = (int)s_pre;
        if (j < a.n_list) a.pt_rank[j] = cg + ex;
        if (bid == nl - 1 && threadIdx.x == 0) a.counts[3] = cg + tot;   // NV
        return;
    }
    // ---- a workgroup of 1024 pixels
    const int bid = lb_ticket(a.tickets, &s_bid);
    const bool list_mode = a.n_list > 0;
    const long long b0 = (long long)bid * FRAME_ITEMS + threadIdx.x * 4;
    bool fv[4], fm[4];
    int nv = 0, nm = 0;
    for (int k = 0; k < 4; ++k) {
        const bool in = b0 + k < a.npix;
        fv[k] = in && !list_mode && a.valid_mask[b0 + k] != 0.f;   // torch.nonzero: NaN counts, -0.0 does not
        fm[k] = in && (!a.miss_mask || a.miss_mask[b0 + k] != 0.f);
        nv += fv[k] ? 1 : 0;
        nm += fm[k] ? 1 : 0;
    }
    int tv, tm;
    const int ev = block_scan_256(nv, s_tmp, tv);
    const int em = block_scan_256(nm, s_tmp, tm);
    const unsigned long long own = ((unsigned long long)tv << 31) | (unsigned long long)tm;
    if (threadIdx.x == 0) lb_store(a.stA, bid, bid == 0 ? LIDF_LB_INC : LIDF_LB_AGG, own);
    if (threadIdx.x < 64) {
        const unsigned long long pre = bid > 0 ? lb_exclusive(a.stA, bid, lane) : 0ull;
        if (lane == 0) {
            if (bid > 0) lb_store(a.stA, bid, LIDF_LB_INC, pre + own);
            s_pre = pre;
        }
    }
    __syncthreads();
    const int cv = (int)(s_pre >> 31), cm = (int)(s_pre & 0x7fffffffull);
    int iv = cv + ev, im = cm + em;
    int jsel[4] = {-1, -1, -1, -1}, ng = 0;
    bool okk[4] = {false, false, false, false};
    for (int k = 0; k < 4; ++k) {
        const long long i = b0 + k;
        if (i >= a.npix) break;
        const int b = (int)(i / hw);
        const int rem = (int)(i % hw);
        const float* px = a.xyz + (size_t)b * 3 * hw + rem;
        const float z = px[2 * hw];
        a.depth[i] = z;
        if (a.depth2) a.depth2[i] = z;
        if (fv[k]) {
            if (iv % a.stride == 0) {
                jsel[k] = iv / a.stride;
                okk[k] = frame_valid_point(jsel[k], b, rem, px, a.rgb + (size_t)b * 3 * hw + rem, hw, a.g,
                                           a.valid_bid, a.valid_flat, a.valid_xyz, a.valid_rgb, a.cell_flag,
                                           a.pt_key);
                ng += okk[k] ? 1 : 0;
            }
            ++iv;
        }
        if (fm[k]) {
            const int y = rem / a.W, x = rem % a.W;
            a.ray_bid[im] = b;
            a.ray_flat[im] = rem;
            a.ray_pix[2 * im] = x;
            a.ray_pix[2 * im + 1] = y;
            const float fx = a.intr[4 * b], fy = a.intr[4 * b + 1], cx = a.intr[4 * b + 2], cy = a.intr[4 * b + 3];
            const float vx = (float)x - cx;            // pipeline.py:215-219, as lidf_ray_dirs_kernel
            const float vy = ((float)y - cy) * fx / fy;
            const float vz = fx;
            const float nrm = sqrtf(vx * vx + vy * vy + vz * vz);
            a.ray_dir[3 * im] = vx / nrm;
            a.ray_dir[3 * im + 1] = vy / nrm;
            a.ray_dir[3 * im + 2] = vz / nrm;
            ++im;
        }
    }
    int cg = 0, tg = 0;
    if (!list_mode) {   // rank of the in-grid points among the selected valid points
        const int eg = block_scan_256(ng, s_tmp, tg);
        if (threadIdx.x == 0) lb_store(a.stB, bid, bid == 0 ? LIDF_LB_INC : LIDF_LB_AGG, (unsigned)tg);
        if (threadIdx.x < 64) {
            const unsigned long long pre = bid > 0 ? lb_exclusive(a.stB, bid, lane) : 0ull;
            if (lane == 0) {
                if (bid > 0) lb_store(a.stB, bid, LIDF_LB_INC, pre + (unsigned)tg);
                s_pre = pre;
            }
        }
        __syncthreads();
        cg = (int)s_pre;
        int run = cg + eg;
        for (int k = 0; k < 4; ++k) {
            if (jsel[k] < 0) continue;
            a.pt_rank[jsel[k]] = run;
            run += okk[k] ? 1 : 0;
        }
    }
    if (bid == a.nb - 1 && threadIdx.x == 0) {
        const int nv0 = cv + tv;
        a.counts[0] = cm + tm;                                                     // R
        a.counts[4] = list_mode ? (int)a.n_list : nv0;                             // NV0
        a.counts[5] = list_mode ? (int)a.n_list : (nv0 + a.stride - 1) / a.stride; // NVS: valid_idx[::stride]
        a.counts[7] = 0;
        if (!list_mode) a.counts[3] = cg + tg;                                     // NV
    }
}

// Occupied cells -> voxels (batch_get_occupied_idx's torch.unique over (bid, x, y, z) rows = the cells in
// key order; LIDF.get_occ_vox_bound, models/pipeline.py:162-201): one workgroup scans the B x rx x ry x rz
// cell marks (a frame has 729 cells) and writes cell_rank [ncell + 1], the voxel tables and V.
__global__ void __launch_bounds__(1024) lidf_frame_cells_kernel(const int* __restrict__ cell_flag,
                                                                 long long ncell, GridSpec g,
                                                                 int* __restrict__ cell_rank,
                                                                 int* __restrict__ occ, float* __restrict__ vbound,
                                                                 int* __restrict__ vox_bid,
                                                                 float* __restrict__ vox_center,
                                                                 int* __restrict__ counts,
                                                                 int* __restrict__ vox_start) {
    __shared__ int s_w[16];
    const long long cpi = (long long)g.r[0] * g.r[1] * g.r[2];   // cells per image
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int carry = 0;
    for (long long b = 0; b < ncell; b += 1024) {
        const long long k = b + threadIdx.x;
        const int f = (k < ncell && cell_flag[k] != 0) ? 1 : 0;
        int inc = f;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const int o = __shfl_up(inc, s);
            if (lane >= s) inc += o;
        }
        __syncthreads();
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        int wpre = 0, total = 0;
        for (int w = 0; w < 16; ++w) {
            const int t = s_w[w];
            wpre += w < wave ? t : 0;
            total += t;
        }
        const int v = carry + wpre + inc - f;
        if (k < ncell) {
            // the voxels are in cell order, image-major: image i owns voxels [vox_start[i], vox_start[i + 1])
            if (vox_start && k % cpi == 0) vox_start[k / cpi] = v;
            cell_rank[k] = f ? v : -1;   // cell -> voxel table (-1: not occupied)
            if (f) {
                int rem = (int)k;
                const int cz = rem % g.r[2]; rem /= g.r[2];
                const int cy = rem % g.r[1]; rem /= g.r[1];
                const int cx = rem % g.r[0]; rem /= g.r[0];
                occ[4 * v + 0] = rem;
                vox_bid[v] = rem;   // the image index on its own ([V] i32: what the box tests take)
                occ[4 * v + 1] = cx;
                occ[4 * v + 2] = cy;
                occ[4 * v + 3] = cz;
                const int c[3] = {cx, cy, cz};
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float lo = g.xmin[a] + (float)c[a] * g.crop;  // pipeline.py:186
                    vbound[6 * v + a] = lo;
                    vbound[6 * v + 3 + a] = lo + g.crop;               // :187
                    // intersect_pos_type 'rel': the voxel centre (bound_min + bound_max) / 2 (pipeline.py:355-360)
                    if (vox_center) vox_center[3 * v + a] = (lo + (lo + g.crop)) / 2.f;
                }
            }
        }
        carry += total;
    }
    if (threadIdx.x == 0) {
        cell_rank[ncell] = carry;
        counts[2] = carry;   // V
        if (vox_start) vox_start[ncell / cpi] = carry;
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

extern "C" size_t lidf_frame_head_blocks(long long npix) { return (size_t)((npix + FRAME_ITEMS - 1) / FRAME_ITEMS); }

// lb: zeroed scratch of lidf_frame_head_lb_bytes(npix) bytes (two tickets + three status arrays)
extern "C" size_t lidf_frame_head_lb_bytes(long long npix) {
    const size_t nb = lidf_frame_head_blocks(npix), nl = (size_t)((npix + 255) / 256);
    return 64 + (2 * nb + nl) * 8;
}

extern "C" hipError_t lidf_launch_frame_head(const float* valid_mask, const float* miss_mask,
                                             const float* xyz, const float* rgb, const float* intr,
                                             int B, int H, int W, int stride, const GridSpec& g,
                                             void* lb, int* counts, int* valid_bid,
                                             int* valid_flat, float* valid_xyz, float* valid_rgb,
                                             int* cell_flag, int* pt_key, int* pt_rank, int* ray_bid,
                                             int* ray_flat, int* ray_pix, float* ray_dir, float* depth,
                                             float* depth2, const int* idx_bid, const int* idx_flat,
                                             long long n_list, hipStream_t st) {
    const long long npix = (long long)B * H * W;
    if (npix <= 0) return hipSuccess;
    const int nb = (int)lidf_frame_head_blocks(npix);
    FrameHeadArgs a;
    a.valid_mask = valid_mask; a.miss_mask = miss_mask; a.xyz = xyz; a.rgb = rgb; a.intr = intr;
    a.npix = npix; a.B = B; a.H = H; a.W = W; a.stride = stride; a.nb = nb; a.g = g;
    a.tickets = (int*)lb;
    a.stA = (unsigned long long*)((char*)lb + 64);
    a.stB = a.stA + nb;
    a.stL = a.stB + nb;
    a.counts = counts; a.valid_bid = valid_bid; a.valid_flat = valid_flat; a.valid_xyz = valid_xyz;
    a.valid_rgb = valid_rgb; a.cell_flag = cell_flag; a.pt_key = pt_key; a.pt_rank = pt_rank;
    a.ray_bid = ray_bid; a.ray_flat = ray_flat; a.ray_pix = ray_pix; a.ray_dir = ray_dir;
    a.depth = depth; a.depth2 = depth2; a.idx_bid = idx_bid; a.idx_flat = idx_flat;
    a.n_list = n_list > 0 ? n_list : 0;
    const int nl = (int)((a.n_list + 255) / 256);
    hipLaunchKernelGGL(lidf_frame_head_kernel, dim3(nb + nl), dim3(256), 0, st, a);
    return hipGetLastError();
}

extern "C" hipError_t lidf_launch_frame_cells(const int* cell_flag, long long ncell, const GridSpec& g,
                                              int* cell_rank, int* occ, float* vbound, int* vox_bid,
                                              float* vox_center, int* counts, int* vox_start,
                                              hipStream_t st) {
    if (ncell <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_frame_cells_kernel, dim3(1), dim3(1024), 0, st, cell_flag, ncell, g, cell_rank, occ,
                       vbound, vox_bid, vox_center, counts, vox_start);
    return hipGetLastError();
}

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
