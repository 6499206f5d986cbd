// lidf_refine.hip — per-ray kernels of the stage-2 refinement query
// (RefineNet.get_pred_refine, models/pipeline.py:922-1030), built with -ffp-contract=off.
#include "lidf_device.h"

// inside test of extensions/pcl_aabb/pcl_aabb_cuda_kernel.cu:23-44 (inclusive bounds)
__device__ __forceinline__ bool inside_box(float x, float y, float z, const float* vb) {
    if ((x < vb[0]) || (x > vb[3])) return false;
    if ((y < vb[1]) || (y > vb[4])) return false;
    if ((z < vb[2]) || (z > vb[5])) return false;
    return true;
}

extern "C" hipError_t lidf_launch_refine_prep_dev(const float*, const long long*, const int*, long long,
                                                  const float*, const int*, long long, const int*,
                                                  const int*, const float*, long long, int, long long,
                                                  float*, int*, int*, const unsigned char*, const int*,
                                                  const int*, hipStream_t, const CellLookup*);
extern "C" hipError_t lidf_launch_refine_rows_dev(const float*, const int*, const float*, const float*, int,
                                                  int, int, int, long long, const int*, float*, int, int,
                                                  hipStream_t);
extern "C" hipError_t lidf_launch_refine_finish_dev(const float*, const float*, const float*, float, float,
                                                    long long, const int*, float*, const int*, const int*,
                                                    long long, float*, hipStream_t);
extern "C" hipError_t lidf_launch_zero_segments(float* const* ptrs, const long long* counts, int n,
                                                hipStream_t st);

// One thread per ray:
//   end_voxel = max( voxel of the arg-max pair (0 for a ray without pairs: the dummy row,
//                    pipeline.py:941-943), largest occupied voxel of the same image containing
//                    pred_pos (pcl_aabb + scatter max, :939-944) )
//   pnet_inp  = [pred_pos - centre(end_voxel) | rgb(pixel)]   (:975-986, pnet_pos_type 'rel')
//   inp_embed[:, 128:] = [ROI feature | embed(pos) | embed(dir)]   (:947-969, :1019-1026);
//   columns 0..127 (voxel feature) are filled after the PointNet pass.
// Pass 1 (grid.y = slices of the voxel list): end_voxel[r] (zeroed) takes the maximum of the
// arg-max pair's voxel and the containing voxels of this slice. Every ray is tested against every
// voxel (the reference's pcl_aabb + scatter max); the voxel index is wave-uniform, so the bounds
// come through the scalar cache as SGPR operands of the compares (eight voxels per batch of
// scalar loads), and the slices keep every SIMD busy at 76,800 rays x 729 voxels (one long loop per
// ray: 150 us; now ~35). Same predicate as inside_box (a NaN coordinate fails no comparison).
__global__ void __launch_bounds__(256) lidf_refine_endvox_kernel(
    const float* __restrict__ pred_pos, const long long* __restrict__ max_pair_id,
    const int* __restrict__ pair_vox, long long P, const float* __restrict__ vbound,
    const int* __restrict__ vox_bid, long long V, long long per_slice,
    const int* __restrict__ ray_bid, long long R, int* __restrict__ end_voxel,
    const int* __restrict__ dims) {
    if (dims) {   // device-side counts {R, P, V} (the sync-free frame path): R, P, V above = capacities
        R = dims[0];
        P = dims[1];
        V = dims[2];
        per_slice = (V + gridDim.y - 1) / gridDim.y;
    }
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = r < R;
    float x = 0.f, y = 0.f, z = 0.f;
    int bid = -1, ev = 0;
    if (live) {
        x = pred_pos[3 * r];
        y = pred_pos[3 * r + 1];
        z = pred_pos[3 * r + 2];
        bid = ray_bid[r];
        if (blockIdx.y == 0) {
            const long long m = max_pair_id[r];
            ev = (m >= 0 && m < P) ? pair_vox[m] : 0;
        }
    }
    const long long j0 = blockIdx.y * per_slice;
    const long long j1 = j0 + per_slice < V ? j0 + per_slice : V;
#pragma unroll 8
    for (long long j = j0; j < j1; ++j) {
        const float* vb = vbound + 6 * j;
        const bool in = (vox_bid[j] == bid) & !(x < vb[0]) & !(x > vb[3]) & !(y < vb[1]) & !(y > vb[4]) &
                        !(z < vb[2]) & !(z > vb[5]);
        ev = in ? max(ev, (int)j) : ev;
    }
    if (live && ev > 0) atomicMax(end_voxel + r, ev);
}

// The same end voxel through a cell table (stepwise API, LidfRefineArgs.voxel_coord): O(1) per ray.
// cell_table[cell] = largest voxel index occupying the cell (-1 = empty), scattered from the caller's
// voxel_coord / voxel_bid by the kernel below. The cell of a point is estimated from (p - xmin) / crop; the
// voxels of the 27 cells around it are tested with the reference's inclusive predicate on their STORED
// bounds, so the decision is inside_box's and the estimate only has to be right to within one cell — true
// for boxes that are the grid's cells up to rounding. A NaN coordinate fails no comparison (the point is
// "inside" every voxel of its image): such a ray walks the list, as the every-voxel kernel decides.
__global__ void lidf_cell_table_kernel(const int* __restrict__ voxel_coord, const int* __restrict__ vox_bid,
                                       long long V, GridSpec g, int* __restrict__ cell_table) {
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const int b = vox_bid[v], cx = voxel_coord[3 * v], cy = voxel_coord[3 * v + 1], cz = voxel_coord[3 * v + 2];
    if (b < 0 || b >= g.B || cx < 0 || cx >= g.r[0] || cy < 0 || cy >= g.r[1] || cz < 0 || cz >= g.r[2]) return;
    atomicMax(cell_table + (((long long)b * g.r[0] + cx) * g.r[1] + cy) * g.r[2] + cz, (int)v);
}

__global__ void __launch_bounds__(256) lidf_refine_endvox_cells_kernel(
    const float* __restrict__ pred_pos, const long long* __restrict__ max_pair_id,
    const int* __restrict__ pair_vox, long long P, const float* __restrict__ vbound,
    const int* __restrict__ vox_bid, long long V, GridSpec g, const int* __restrict__ cell_table,
    const int* __restrict__ ray_bid, long long R, int* __restrict__ end_voxel) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float x = pred_pos[3 * r], y = pred_pos[3 * r + 1], z = pred_pos[3 * r + 2];
    const int bid = ray_bid[r];
    const long long m = max_pair_id[r];
    int ev = (m >= 0 && m < P) ? pair_vox[m] : 0;
    const float pp[3] = {x, y, z};
    int q[3];
    bool near = bid >= 0 && bid < g.B;
    bool nan = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v = (pp[k] - g.xmin[k]) / g.crop;
        nan = nan || (pp[k] != pp[k]);
        near = near && v > -2.f && v < (float)g.r[k] + 2.f;   // (false for NaN)
        q[k] = near ? (int)floorf(v) : 0;
    }
    if (nan) {
        for (long long j = 0; j < V; ++j)
            if (vox_bid[j] == bid && inside_box(x, y, z, vbound + 6 * j)) ev = max(ev, (int)j);
    } else if (near) {
        int cv[27];   // the 27 cell -> voxel entries are requested together (independent loads)
#pragma unroll
        for (int i = 0; i < 27; ++i) {
            const int cx = q[0] + i / 9 - 1, cy = q[1] + (i / 3) % 3 - 1, cz = q[2] + i % 3 - 1;
            const bool ok = cx >= 0 && cx < g.r[0] && cy >= 0 && cy < g.r[1] && cz >= 0 && cz < g.r[2];
            cv[i] = ok ? cell_table[(((long long)bid * g.r[0] + cx) * g.r[1] + cy) * g.r[2] + cz] : -1;
        }
#pragma unroll
        for (int i = 0; i < 27; ++i)
            if (cv[i] > ev && inside_box(x, y, z, vbound + 6 * (size_t)cv[i])) ev = cv[i];
    }
    end_voxel[r] = ev;
}

// Pass 2, one thread per ray: pnet_vox, pnet_inp from the end voxel.
__global__ void lidf_refine_prep_kernel(const float* __restrict__ pred_pos,
                                        const float* __restrict__ vbound,
                                        const int* __restrict__ ray_bid,
                                        const int* __restrict__ ray_flat,
                                        const float* __restrict__ rgb, long long hw, int pnet_rel,
                                        long long R, float* __restrict__ pnet_inp,
                                        int* __restrict__ pnet_vox,
                                        const int* __restrict__ end_voxel,
                                        const unsigned char* __restrict__ pnet_select,
                                        const int* __restrict__ R_dev,
                                        const int* __restrict__ row0_dev) {
    if (R_dev) R = *R_dev;
    if (row0_dev) {   // frame path: the predicted points follow the *row0_dev valid points
        pnet_inp += 6 * (size_t)*row0_dev;
        pnet_vox += *row0_dev;
    }
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float x = pred_pos[3 * r], y = pred_pos[3 * r + 1], z = pred_pos[3 * r + 2];
    const int bid = ray_bid[r], ev = end_voxel[r];
    const float* vb = vbound + 6 * (size_t)ev;
    const float cx = (vb[0] + vb[3]) / 2.f, cy = (vb[1] + vb[4]) / 2.f, cz = (vb[2] + vb[5]) / 2.f;
    // use_all_pix == False (pipeline.py:987-996): an unselected ray's point stays out of the
    // PointNet — voxel index -1 is skipped by the pooling and gather epilogues
    pnet_vox[r] = (!pnet_select || pnet_select[r]) ? ev : -1;
    float* pi = pnet_inp + 6 * r;
    pi[0] = pnet_rel ? x - cx : x;
    pi[1] = pnet_rel ? y - cy : y;
    pi[2] = pnet_rel ? z - cz : z;
    const float* px = rgb + (size_t)bid * 3 * hw + ray_flat[r];
    pi[3] = px[0];
    pi[4] = px[hw];
    pi[5] = px[2 * hw];
}

// inp_embed[r, 128:] = [ROI feature | embed(pos) | embed(dir)]  (pipeline.py:947-969, :1019-1026):
// one thread per (ray, column) so that the rows leave as coalesced segments; columns 0..127 (voxel
// feature) are filled after the PointNet pass. Same expressions as the per-ray loop it replaces.
#define ROWS_PER_WG 32
__global__ void lidf_refine_rows_kernel(const float* __restrict__ pred_pos,
                                        const int* __restrict__ end_voxel,
                                        const float* __restrict__ vbound,
                                        const float* __restrict__ rayfeat, int ld_rf, int Lv, int L,
                                        int pos_rel, long long R, float* __restrict__ inp_embed,
                                        int ld_e, const int* __restrict__ R_dev, int pos_only, long long r_base) {
    if (R_dev) R = *R_dev;
    const int E = 3 + 6 * L, Ed = 3 + 6 * Lv;
    // pos_only: a later iteration on the same rays — only embed(pos) changed (columns 128 .. 128+E)
    const int c_lo = pos_only ? 128 : 0, ncol = pos_only ? E : 128 + E + Ed;
    // (a 64 x 4 workgroup = 64 consecutive columns of 4 rays at a time: no index division)
    const int cc = blockIdx.x * 64 + threadIdx.x;
    if (cc >= ncol) return;
    const int c = c_lo + cc;
    // (a workgroup walks ROWS_PER_WG rays, four at a time: a workgroup per four rays was bound by its own
    // dispatch — 160,000 workgroups of a few instructions each)
#pragma unroll 2
    for (int it = 0; it < ROWS_PER_WG / 4; ++it) {
        const long long r = r_base + (long long)blockIdx.y * ROWS_PER_WG + 4 * it + threadIdx.y;
        if (r >= R) return;
        const float* rf = rayfeat + (size_t)r * ld_rf;
        float v;
        if (c < 128) {
            v = rf[c];
        } else if (c < 128 + E) {
            const int k = c - 128;                    // index inside embed(pos)
            const int a = k < 3 ? k : (k - 3) % 3;    // coordinate
            float q = pred_pos[3 * r + a];
            if (pos_rel) {
                const float* vb = vbound + 6 * (size_t)end_voxel[r];
                q = q - (vb[a] + vb[3 + a]) / 2.f;
            }
            if (k < 3) {
                v = q;
            } else {
                v = pe_value(q, k);   // (5 VALU; |err| <= 4.2e-7 — lidf_device.h: to_rev / rev_sincos)
            }
        } else {
            v = rf[128 + (c - 128 - E)];
        }
        inp_embed[(size_t)r * ld_e + 128 + c] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// One refine iteration's per-ray work in ONE launch (the sync-free frame path, round 4; rounds 2-3:
// zero -> end voxel -> PointNet rows -> embed(pos) rows, and the previous iteration's finish = 5
// launches): a workgroup owns 128 rays.
//   phase 1 (one thread per ray): the position this iteration starts from — pred_pos of stage 1, or
//     prev + (off * rs + r0) * dir of the previous iteration (pipeline.py:1028-1029; the expressions
//     of lidf_refine_finish_kernel) —, the end voxel, the PointNet row (lidf_refine_prep_kernel);
//   phase 2 (all threads): the 128 x E block embed(pos) of the decoder rows, written as row segments
//     (lidf_refine_rows_kernel's expressions).
// End voxel = the largest occupied voxel of the ray's image whose box contains the point (the
// reference's pcl_aabb + scatter max, pipeline.py:939-944). The voxels are cells of the frame's grid:
// the cell of the point is estimated from (p - xmin) / crop and the 27 cells around it are tested with
// the reference's own inclusive predicate on the voxels' stored bounds — the estimate only has to be
// right to within one cell, the decision is inside_box's. Voxel indices ascend with the cell key. A
// NaN coordinate fails no comparison of inside_box (the point is "inside" every voxel of its image):
// such a ray walks the voxel list as the reference does.
// The workgroups also zero the PointNet's max-pool tables of this iteration (zero0 / zero1).
// ------------------------------------------------------------------------------------------------

#define STEP_RAYS 128   // rays per workgroup: phase 1 on threads 0..127, phase 2 on all 256
__global__ void __launch_bounds__(256) lidf_refine_step_kernel(RefineStepArgs a) {
    __shared__ float s_q[STEP_RAYS * 3];
    {   // this iteration's zero-initialised scratch
        const long long tot = a.nzero0 + a.nzero1;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < tot; i += (long long)gridDim.x * 256) {
            if (i < a.nzero0) a.zero0[i] = 0.f;
            else a.zero1[i - a.nzero0] = 0.f;
        }
    }
    const long long R = a.dims[0], P = a.dims[1], V = a.dims[2];
    const long long r0w = (long long)blockIdx.x * STEP_RAYS;
    if (r0w >= R) return;
    if (threadIdx.x < STEP_RAYS) {
        const long long r = r0w + threadIdx.x;
        if (r < R) {
            float x = a.prev_pos[3 * r], y = a.prev_pos[3 * r + 1], z = a.prev_pos[3 * r + 2];
            if (a.prev_off) {
                const float s = a.prev_off[r] * a.rs + a.r0;
                x = x + s * a.ray_dir[3 * r];
                y = y + s * a.ray_dir[3 * r + 1];
                z = z + s * a.ray_dir[3 * r + 2];
                a.cur_pos[3 * r] = x;
                a.cur_pos[3 * r + 1] = y;
                a.cur_pos[3 * r + 2] = z;
            }
            const int bid = a.ray_bid[r];
            const long long m = a.max_pair_id[r];
            const int rflat = a.ray_flat[r];
            int ev = (m >= 0 && m < P) ? a.pair_vox[m] : 0;
            const float* px = a.rgb + (size_t)bid * 3 * a.hw + rflat;
            const float c0r = px[0], c1r = px[a.hw], c2r = px[2 * a.hw];
            const float pp[3] = {x, y, z};
            if (x != x || y != y || z != z) {
                for (long long j = 0; j < V; ++j)
                    if (a.vox_bid[j] == bid && inside_box(x, y, z, a.vbound + 6 * j)) ev = max(ev, (int)j);
            } else {
                // per axis: the estimated cell q and, for the cells q-1, q, q+1, the reference's inclusive
                // test on the bounds of that cell — bound_min = xmin + c * crop, bound_max = bound_min + crop,
                // the expressions (and so the bits) lidf_frame_cells_kernel stored in voxel_bound
                int q[3];
                bool in[3][3];
                bool any = true;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float v = (pp[k] - a.g.xmin[k]) / a.g.crop;
                    const bool near = v > -2.f && v < (float)a.g.r[k] + 2.f;
                    any = any && near;
                    q[k] = near ? (int)floorf(v) : 0;
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        const int c = q[k] + d - 1;
                        const float lo = a.g.xmin[k] + (float)c * a.g.crop;
                        const float hi = lo + a.g.crop;
                        in[k][d] = near && c >= 0 && c < a.g.r[k] && !(pp[k] < lo) && !(pp[k] > hi);
                    }
                }
                if (any) {
                    // the 27 cell -> voxel entries are requested together (independent loads)
                    int cv[27];
#pragma unroll
                    for (int i = 0; i < 27; ++i) {
                        const int dx = i / 9, dy = (i / 3) % 3, dz = i % 3;
                        const bool ok = in[0][dx] && in[1][dy] && in[2][dz];
                        const int cx = q[0] + dx - 1, cy = q[1] + dy - 1, cz = q[2] + dz - 1;
                        const int key = ((bid * a.g.r[0] + cx) * a.g.r[1] + cy) * a.g.r[2] + cz;
                        cv[i] = ok ? a.cell_rank[key] : -1;   // (cell -> voxel, -1 = not occupied)
                    }
#pragma unroll
                    for (int i = 0; i < 27; ++i) ev = max(ev, cv[i]);
                }
            }
            a.end_voxel[r] = ev;
            const float* vb = a.vbound + 6 * (size_t)ev;
            const float cxx = (vb[0] + vb[3]) / 2.f, cyy = (vb[1] + vb[4]) / 2.f, czz = (vb[2] + vb[5]) / 2.f;
            const long long row = *a.row0_dev + r;
            a.pnet_vox[row] = (!a.sel || a.sel[r]) ? ev : -1;
            float* pi = a.pnet_inp + 6 * row;
            pi[0] = a.pnet_rel ? x - cxx : x;
            pi[1] = a.pnet_rel ? y - cyy : y;
            pi[2] = a.pnet_rel ? z - czz : z;
            pi[3] = c0r;
            pi[4] = c1r;
            pi[5] = c2r;
            s_q[3 * threadIdx.x] = a.pos_rel ? x - cxx : x;
            s_q[3 * threadIdx.x + 1] = a.pos_rel ? y - cyy : y;
            s_q[3 * threadIdx.x + 2] = a.pos_rel ? z - czz : z;
        }
    }
    __syncthreads();
    const int E = 3 + 6 * a.L;
    const int nr = (int)min((long long)STEP_RAYS, R - r0w);
    for (int i = threadIdx.x; i < nr * E; i += 256) {
        const int rr = i / E, k = i % E;
        const int ax = k < 3 ? k : (k - 3) % 3;    // coordinate
        const float q = s_q[3 * rr + ax];
        float v;
        if (k < 3) {
            v = q;
        } else {
            v = pe_value(q, k);   // (5 VALU; |err| <= 4.2e-7 — lidf_device.h: to_rev / rev_sincos)
        }
        a.inp_embed[(size_t)(r0w + rr) * a.ld_e + 256 + k] = v;
    }
}

extern "C" hipError_t lidf_launch_refine_step(const RefineStepArgs& a, long long R_cap, hipStream_t st) {
    if (R_cap <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_refine_step_kernel, dim3((unsigned)((R_cap + STEP_RAYS - 1) / STEP_RAYS)), dim3(256), 0, st, a);
    return hipGetLastError();
}

extern "C" hipError_t lidf_launch_refine_prep(const float* pred_pos, const long long* max_pair_id,
                                              const int* pair_vox, long long P, const float* vbound,
                                              const int* vox_bid, long long V, const int* ray_bid,
                                              const int* ray_flat, const float* rgb, long long hw,
                                              const float* rayfeat, int ld_rf, int Lv, int L,
                                              int pnet_rel, int pos_rel, long long R,
                                              float* pnet_inp, int* pnet_vox, float* inp_embed,
                                              int ld_e, int* end_voxel,
                                              const unsigned char* pnet_select, hipStream_t st,
                                              const CellLookup* cells) {
    return lidf_launch_refine_prep_dev(pred_pos, max_pair_id, pair_vox, P, vbound, vox_bid, V, ray_bid, ray_flat,
                                       rgb, hw, pnet_rel, R, pnet_inp, pnet_vox, end_voxel, pnet_select,
                                       nullptr, nullptr, st, cells);
    (void)rayfeat; (void)ld_rf; (void)Lv; (void)L; (void)pos_rel; (void)inp_embed; (void)ld_e;
}

// dims (optional): device int32 {R, P, V} overriding the host counts (which then bound the launches);
// row0_dev (optional): device row offset of pnet_inp / pnet_vox (the number of valid points)
extern "C" hipError_t lidf_launch_refine_prep_dev(const float* pred_pos, const long long* max_pair_id,
                                                  const int* pair_vox, long long P, const float* vbound,
                                                  const int* vox_bid, long long V, const int* ray_bid,
                                                  const int* ray_flat, const float* rgb, long long hw,
                                                  int pnet_rel, long long R, float* pnet_inp,
                                                  int* pnet_vox, int* end_voxel,
                                                  const unsigned char* pnet_select, const int* dims,
                                                  const int* row0_dev, hipStream_t st,
                                                  const CellLookup* cells) {
    if (R <= 0) return hipSuccess;
    if (cells && cells->table) {   // end voxel through the cell table (stepwise API with a grid)
        const long long nc = (long long)cells->g.B * cells->g.r[0] * cells->g.r[1] * cells->g.r[2];
        if (!cells->ready) {
            hipError_t e = hipMemsetAsync(cells->table, 0xff, (size_t)nc * 4, st);
            if (e != hipSuccess) return e;
            if (V > 0)
                hipLaunchKernelGGL(lidf_cell_table_kernel, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st,
                                   cells->coord, vox_bid, V, cells->g, cells->table);
        }
        hipLaunchKernelGGL(lidf_refine_endvox_cells_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st,
                           pred_pos, max_pair_id, pair_vox, P, vbound, vox_bid, V, cells->g, cells->table, ray_bid,
                           R, end_voxel);
        hipLaunchKernelGGL(lidf_refine_prep_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st,
                           pred_pos, vbound, ray_bid, ray_flat, rgb, hw, pnet_rel, R, pnet_inp, pnet_vox,
                           end_voxel, pnet_select, dims, row0_dev);
        return hipGetLastError();
    }
    // (the frame path — dims — zeroes end_voxel with its other scratch of the iteration, in one launch)
    if (!dims) {
        hipError_t e = hipMemsetAsync(end_voxel, 0, (size_t)R * 4, st);
        if (e != hipSuccess) return e;
    }
    // slices of at least 64 voxels, enough of them for ~8 wavefronts per SIMD
    const long long waves = (R + 63) / 64;
    long long slices = (8 * 1024 + waves - 1) / waves;
    const long long max_slices = (V + 63) / 64;
    if (slices > max_slices) slices = max_slices;
    if (slices < 1) slices = 1;
    const long long per_slice = (V + slices - 1) / slices;
    slices = V > 0 ? (V + per_slice - 1) / per_slice : 1;
    hipLaunchKernelGGL(lidf_refine_endvox_kernel, dim3((unsigned)((R + 255) / 256), (unsigned)slices),
                       dim3(256), 0, st, pred_pos, max_pair_id, pair_vox, P, vbound, vox_bid, V,
                       per_slice, ray_bid, R, end_voxel, dims);
    hipLaunchKernelGGL(lidf_refine_prep_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st,
                       pred_pos, vbound, ray_bid, ray_flat, rgb, hw, pnet_rel, R, pnet_inp, pnet_vox,
                       end_voxel, pnet_select, dims, row0_dev);
    return hipGetLastError();
}

extern "C" hipError_t lidf_launch_refine_rows(const float* pred_pos, const int* end_voxel,
                                              const float* vbound, const float* rayfeat, int ld_rf,
                                              int Lv, int L, int pos_rel, long long R,
                                              float* inp_embed, int ld_e, hipStream_t st) {
    return lidf_launch_refine_rows_dev(pred_pos, end_voxel, vbound, rayfeat, ld_rf, Lv, L, pos_rel, R, nullptr,
                                       inp_embed, ld_e, 0, st);
}
extern "C" hipError_t lidf_launch_refine_rows_dev(const float* pred_pos, const int* end_voxel,
                                                  const float* vbound, const float* rayfeat, int ld_rf,
                                                  int Lv, int L, int pos_rel, long long R,
                                                  const int* R_dev, float* inp_embed, int ld_e,
                                                  int pos_only, hipStream_t st) {
    if (R <= 0) return hipSuccess;
    const int ncol = pos_only ? 3 + 6 * L : 128 + 3 + 6 * L + 3 + 6 * Lv;
    // grid.y holds the rays in groups of ROWS_PER_WG, in slabs of at most 65,535 groups
    const long long groups = (R + ROWS_PER_WG - 1) / ROWS_PER_WG;
    for (long long g0 = 0; g0 < groups; g0 += 65535) {
        const long long ng = groups - g0 < 65535 ? groups - g0 : 65535;
        hipLaunchKernelGGL(lidf_refine_rows_kernel, dim3((unsigned)((ncol + 63) / 64), (unsigned)ng), dim3(64, 4), 0,
                           st, pred_pos, end_voxel, vbound, rayfeat, ld_rf, Lv, L, pos_rel, R, inp_embed, ld_e, R_dev,
                           pos_only, g0 * ROWS_PER_WG);
    }
    return hipGetLastError();
}

// inp_embed[r, 0:128] = occ_voxel_feat[end_voxel[r]]  (pipeline.py:1016); one thread per float4
__global__ void lidf_refine_gather_kernel(const float* __restrict__ vox_feat,
                                          const int* __restrict__ end_voxel, long long R,
                                          float* __restrict__ inp_embed, int ld_e,
                                          const int* __restrict__ R_dev) {
    if (R_dev) R = *R_dev;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * 32) return;
    const long long r = i / 32;
    const int c = (int)(i % 32) * 4;
    const f32x4 v = *(const f32x4*)(vox_feat + (size_t)end_voxel[r] * 128 + c);
    float* e = inp_embed + (size_t)r * ld_e + c;
    e[0] = v[0];
    e[1] = v[1];
    e[2] = v[2];
    e[3] = v[3];
}

// pred_pos_refine = pred_pos + (off*(r1-r0) + r0) * ray_dir  (pipeline.py:1028-1029)
__global__ void lidf_refine_finish_kernel(const float* __restrict__ pred_pos,
                                          const float* __restrict__ off,
                                          const float* __restrict__ ray_dir, float r0, float rs,
                                          long long R, float* __restrict__ out,
                                          const int* __restrict__ R_dev,
                                          const int* __restrict__ ray_bid,
                                          const int* __restrict__ ray_flat, long long hw,
                                          float* __restrict__ depth) {
    if (R_dev) R = *R_dev;
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    // optional: the refined depth map (pred_xyz[miss_bid, miss_flat_img_id] = pred_pos_refine, z)
    if (depth) depth[(size_t)ray_bid[r] * hw + ray_flat[r]] = pred_pos[3 * r + 2] + (off[r] * rs + r0) * ray_dir[3 * r + 2];
    const float s = off[r] * rs + r0;
    out[3 * r] = pred_pos[3 * r] + s * ray_dir[3 * r];
    out[3 * r + 1] = pred_pos[3 * r + 1] + s * ray_dir[3 * r + 1];
    out[3 * r + 2] = pred_pos[3 * r + 2] + s * ray_dir[3 * r + 2];
}

extern "C" hipError_t lidf_launch_refine_gather_dev(const float* vox_feat, const int* end_voxel,
                                                    long long R, const int* R_dev, float* inp_embed,
                                                    int ld_e, hipStream_t st) {
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_refine_gather_kernel, dim3((unsigned)((R * 32 + 255) / 256)),
                       dim3(256), 0, st, vox_feat, end_voxel, R, inp_embed, ld_e, R_dev);
    return hipGetLastError();
}
extern "C" hipError_t lidf_launch_refine_gather(const float* vox_feat, const int* end_voxel,
                                                long long R, float* inp_embed, int ld_e,
                                                hipStream_t st) {
    return lidf_launch_refine_gather_dev(vox_feat, end_voxel, R, nullptr, inp_embed, ld_e, st);
}

extern "C" hipError_t lidf_launch_refine_finish(const float* pred_pos, const float* off,
                                                const float* ray_dir, float r0, float rs,
                                                long long R, float* out, hipStream_t st) {
    return lidf_launch_refine_finish_dev(pred_pos, off, ray_dir, r0, rs, R, nullptr, out, nullptr, nullptr, 0,
                                         nullptr, st);
}
// R_dev (optional): device-side ray count; depth (optional, with ray_bid / ray_flat / hw): the refined
// depth map written alongside
extern "C" hipError_t lidf_launch_refine_finish_dev(const float* pred_pos, const float* off,
                                                    const float* ray_dir, float r0, float rs,
                                                    long long R, const int* R_dev, float* out,
                                                    const int* ray_bid, const int* ray_flat,
                                                    long long hw, float* depth, hipStream_t st) {
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_refine_finish_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0,
                       st, pred_pos, off, ray_dir, r0, rs, R, out, R_dev, ray_bid, ray_flat, hw, depth);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Training path of stage 2 (lidf_refine_train_forward_f32 / _backward_f32): the per-ray pieces of the
// adjoint of one get_pred_refine (models/pipeline.py:1019-1029 under autograd).
//   pred_pos_out = pred_pos + (off * rs + r0) * dir   =>   d off = rs * <g, dir>,  d pred_pos += g
//   pred_pos also enters the iteration as embed(pos [- centre]) (decoder rows) and as the PointNet row
//   [pos - centre | rgb] of its predicted point: d pred_pos += embed'(x)^T d_pe + d_inp[:, 0:3]
//   (the centre belongs to the end voxel, an index: no gradient).
// ------------------------------------------------------------------------------------------------
// (pre != NULL: also through the decoder's output activation, lidf_out_act_kernel's derivative — goff is then
// dL/d(the last pass's pre-activation), what the decoder's backward starts from)
__global__ void lidf_refine_train_goff_kernel(const float* __restrict__ g, const float* __restrict__ dir,
                                              float rs, long long R, const float* __restrict__ pre,
                                              int use_sigmoid, float* __restrict__ goff) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float s = g[3 * r] * dir[3 * r] + g[3 * r + 1] * dir[3 * r + 1] + g[3 * r + 2] * dir[3 * r + 2];
    float v = s * rs;
    if (pre) {
        const float y = pre[r];
        float d;
        if (use_sigmoid) {
            const float o = 1.f / (1.f + expf(-y));
            d = o * (1.f - o);
        } else {
            d = (y >= 0.f && y <= 1.f) ? 1.f : 0.01f;
        }
        v = v * d;
    }
    goff[r] = v;
}

// out[r, c] = g[r, c] (+ sum_o 2^o (cos(2^o x) d_sin[o, c] - sin(2^o x) d_cos[o, c]) + d_pe[r, c]) (+ d_inp[r, c])
// with x = cur[r, c] (- centre of the end voxel when pos_rel); d_pe [R, 3 + 6 L] / d_inp (row stride 6) may be NULL
__global__ void lidf_refine_train_dcur_kernel(const float* __restrict__ g, const float* __restrict__ cur,
                                              const int* __restrict__ end_voxel, const float* __restrict__ vbound,
                                              int pos_rel, const float* __restrict__ d_pe, int L,
                                              const float* __restrict__ d_inp, long long R,
                                              float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * 3) return;
    const long long r = i / 3;
    const int c = (int)(i % 3);
    float acc = g[i];
    if (d_pe) {
        float v = cur[i];
        if (pos_rel) {
            const float* vb = vbound + 6 * (size_t)end_voxel[r];
            v = v - (vb[c] + vb[3 + c]) / 2.f;
        }
        const float* gr = d_pe + (size_t)r * (3 + 6 * L);
        float e = gr[c];
        float f = 1.f;
        for (int o = 0; o < L; ++o) {
            const float a = v * f;
            e += f * (cosf(a) * gr[3 + 6 * o + c] - sinf(a) * gr[3 + 6 * o + 3 + c]);
            f *= 2.f;
        }
        acc += e;
    }
    if (d_inp) acc += d_inp[6 * r + c];
    out[i] = acc;
}

__global__ void lidf_add_inplace_kernel(float* __restrict__ a, const float* __restrict__ b, long long n4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    f32x4 x = ((const f32x4*)a)[i];
    const f32x4 y = ((const f32x4*)b)[i];
    x[0] += y[0]; x[1] += y[1]; x[2] += y[2]; x[3] += y[3];
    ((f32x4*)a)[i] = x;
}

__global__ void lidf_iota_kernel(int* __restrict__ p, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (int)i;
}

extern "C" hipError_t lidf_launch_refine_train_goff(const float* g, const float* dir, float rs, long long R,
                                                    const float* pre, int use_sigmoid, float* goff,
                                                    hipStream_t st) {
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_refine_train_goff_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, g, dir, rs,
                       R, pre, use_sigmoid, goff);
    return hipGetLastError();
}
extern "C" hipError_t lidf_launch_refine_train_dcur(const float* g, const float* cur, const int* end_voxel,
                                                    const float* vbound, int pos_rel, const float* d_pe, int L,
                                                    const float* d_inp, long long R, float* out, hipStream_t st) {
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_refine_train_dcur_kernel, dim3((unsigned)((3 * R + 255) / 256)), dim3(256), 0, st, g, cur,
                       end_voxel, vbound, pos_rel, d_pe, L, d_inp, R, out);
    return hipGetLastError();
}
// a[0:n] += b[0:n], n a multiple of 4, both 16-byte aligned
extern "C" hipError_t lidf_launch_add_inplace(float* a, const float* b, long long n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_add_inplace_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, a, b, n / 4);
    return hipGetLastError();
}
extern "C" hipError_t lidf_launch_iota(int* p, long long n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(lidf_iota_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n);
    return hipGetLastError();
}
