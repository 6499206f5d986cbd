/*
 * lidf_hip.h — C ABI of liblidf_hip.so: the MI355X (gfx950) implementation of the
 * LIDF per-point implicit-depth query path of NVlabs/implicit_depth.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer borrowed for the call unless marked "host";
 *   - nothing is allocated, nothing is synchronised: work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the null stream);
 *   - return value 0 = LIDF_OK, negative = lidf_status (see lidf_strerror); never throws;
 *   - scratch memory comes from the caller: ask lidf_*_workspace_bytes first;
 *   - re-entrant, no global mutable state (safe for 8 processes x 1 GPU, many streams).
 *
 * Each entry point cites the reference interface it replaces (paths relative to
 * /root/reference/src).
 */
#ifndef LIDF_HIP_H
#define LIDF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* lidf_stream_t; /* hipStream_t */

enum lidf_status {
    LIDF_OK = 0,
    LIDF_ERR_BAD_ARG = -1,      /* NULL pointer / negative size / inconsistent arguments   */
    LIDF_ERR_UNSUPPORTED = -2,  /* dimension outside what the kernels are built for        */
    LIDF_ERR_WORKSPACE = -3,    /* workspace too small                                     */
    LIDF_ERR_HIP = -4           /* a HIP runtime call failed (launch, attribute query)     */
};

/* ABI version, bumped on any signature or struct-layout change and on added entry points. lidf_version() returns the value the
 * library was BUILT with; a binding compiled / written against this header must refuse a library that
 * answers anything else (implicit_depth_amd/_lib.py and csrc/lidf_torch_ext.cpp do, at load). */
#define LIDF_ABI_VERSION 12
int lidf_version(void);
/* Static string for a status code. */
const char* lidf_strerror(int status);

/*
 * One implicit decoder: IMNet (models/implicit_net.py:60-98) or IEF (:100-152).
 * Weights are nn.Linear storage, row-major [out,in], borrowed (never cached across calls:
 * parameters change every optimizer step). Hidden widths are fixed to gf_dim=64
 * (256 -> 128 -> 64 -> 1), the value of every shipped config, for every entry point but
 * lidf_decoder_chain_f32 (gf_dim 32 / 64 / 128: the array shapes below scale with gf_dim); any
 * other width runs layer by layer through lidf_linear_f32 (below).
 */
typedef struct LidfDecoder {
    const float* w1; /* [256, d_in]  (IEF: d_in = D + 16; IMNet: d_in = D) */
    const float* b1; /* [256] */
    const float* w2; /* [128, 256] */
    const float* b2; /* [128] */
    const float* w3; /* [64, 128] */
    const float* b3; /* [64] */
    const float* w4; /* [1, 64] */
    const float* b4; /* [1] */
    const float* wenc; /* IEF offset_enc.weight [16,1]; NULL for IMNet */
    const float* benc; /* IEF offset_enc.bias   [16];   NULL for IMNet */
    int32_t is_ief;    /* 0 = IMNet, 1 = IEF */
    int32_t n_iter;    /* IEF iterations (implicit_net.py:133); ignored for IMNet */
    float init_offset; /* IEF.init_offset (implicit_net.py:104) = 0.001 */
    int32_t use_sigmoid; /* output activation: 0 = leaky clamp (implicit_net.py:96,151), 1 = sigmoid */
} LidfDecoder;

/* ---- Positional encoding -------------------------------------------------------------
 * Replaces Embedder.embed / get_embedder(multires)[0] (models/implicit_net.py:9-57):
 * out[i] = cat(x[i], sin(2^0 x[i]), cos(2^0 x[i]), ..., sin(2^(L-1) x[i]), cos(2^(L-1) x[i])).
 * x: [n,3] f32, out: [n, 3+6*multires] f32. multires in [0,16].                          */
int lidf_embed_f32(const float* x, int64_t n, int multires, float* out, lidf_stream_t stream);

/* ---- Decoders on a materialised input --------------------------------------------------
 * Replaces IMNet.forward / IEF.forward (models/implicit_net.py:81-98, 129-152) and the pair
 * of calls at models/pipeline.py:434-435 (offset_dec + prob_dec on the same inp_embed).
 * inp: [n, d] f32 row-major (row stride ld_inp floats). prob/off may each be NULL
 * (then the matching output is not written). out_*: [n] f32 (= [n,1]).                    */
size_t lidf_decoders_workspace_bytes(int64_t n, int d);
int lidf_decoders_f32(const float* inp, int64_t n, int d, int64_t ld_inp,
                      const LidfDecoder* prob, const LidfDecoder* off,
                      float* out_prob, float* out_off,
                      void* workspace, size_t workspace_bytes, lidf_stream_t stream);
/* Same call with the matrix products of layers 1-3 evaluated as three f16-piece products per term,
 * f32 accumulation (LIDF_PRECISION_F16X3 of LidfQueryArgs): |activations| < 65504 required. */
int lidf_decoders_split_f32(const float* inp, int64_t n, int d, int64_t ld_inp,
                      const LidfDecoder* prob, const LidfDecoder* off,
                      float* out_prob, float* out_off,
                      void* workspace, size_t workspace_bytes, lidf_stream_t stream);

/* ---- Fused per-point query -------------------------------------------------------------
 * Replaces LIDF.get_embedding (positional encoding + ROIAlign gather + voxel feature gather,
 * models/pipeline.py:338-425) + LIDF.get_pred (decoders, scaling, per-ray softmax / argmax /
 * select, models/pipeline.py:427-466) + the depth write-back (models/pipeline.py:593-596).
 * Pairs ("points") are (ray, occupied voxel) intersections.  The per-ray reduction needs the
 * pairs grouped by ray: pair_off is the CSR row pointer over rays (pairs of ray r are
 * [pair_off[r], pair_off[r+1]) ), which is what lidf_ray_aabb_compact_* emits.              */
#define LIDF_PRECISION_F32 0
#define LIDF_PRECISION_F16X3 1

typedef struct LidfQueryArgs {
    /* rays (models/pipeline.py:203-269 outputs) */
    int64_t n_rays;            /* R */
    const float* ray_dir;      /* [R,3] unit directions (miss_ray_dir)            */
    const int32_t* ray_pix;    /* [R,2] integer pixel (x,y) (miss_img_ind)        */
    const int32_t* ray_bid;    /* [R]   image index in the batch (miss_bid)       */
    const int32_t* ray_flat;   /* [R]   y*w+x (miss_flat_img_id); may be NULL if depth==NULL */
    /* pairs, ray-major CSR (models/pipeline.py:271-296 outputs, re-ordered)                  */
    int64_t n_pairs;           /* P */
    const int32_t* pair_off;   /* [R+1] */
    const int32_t* pair_ray;   /* [P] ray of each pair  (miss_ray_intersect_idx)  */
    const int32_t* pair_vox;   /* [P] voxel of each pair (occ_vox_intersect_idx)  */
    const float* pair_t;       /* [P,2] (t_enter, t_leave) (dist[vox,ray])        */
    /* feature sources */
    int32_t batch, height, width; /* feat_grid [B,32,h,w] NCHW (full_rgb_feat)    */
    const float* feat_grid;
    int64_t n_vox;             /* V */
    const float* vox_feat;     /* [V,128] occ_voxel_feat (models/pipeline.py:408) */
    const float* vox_center;   /* [V,3] voxel centres; only read if pos_rel != 0  */
    /* model */
    const LidfDecoder* prob;   /* prob_dec  (IMNet)                               */
    const LidfDecoder* off;    /* offset_dec (IEF or IMNet)                       */
    int32_t multires;          /* opt.model.multires (8); 0 = identity (pos_encode False); <= 16 */
    int32_t multires_views;    /* opt.model.multires_views (4); 0 = identity      */
    int32_t roi_inp_bbox;      /* opt.model.roi_inp_bbox (8)                      */
    int32_t pos_rel;           /* opt.model.intersect_pos_type == 'rel'           */
    float offset_range0, offset_range1; /* opt.grid.offset_range                  */
    float part_size;           /* data_dict['part_size'] (0.25)                   */
    /* outputs (any may be NULL except pred_offset/pred_prob/pair_pred_pos)       */
    float* pred_offset;        /* [P]   offset_dec output, before scaling         */
    float* pred_prob;          /* [P]   pred_prob_end                             */
    float* pair_pred_pos;      /* [P,3]                                           */
    float* pred_prob_softmax;  /* [P]   pred_prob_end_softmax                     */
    int64_t* max_pair_id;      /* [R]   argmax pair per ray, P for an empty ray   */
    float* pred_pos;           /* [R,3] (0,0,0) for an empty ray                  */
    float* depth;              /* [B,h,w] depth[bid, flat] = pred_pos.z; untouched elsewhere */
    /* scratch */
    void* workspace;
    size_t workspace_bytes;
    /* optional output: the per-ray [ROI feature | embed(dir)] rows, [R, 128 + 3+6*multires_views]
     * (what lidf_ray_features_f32 computes) so that stage 2 (lidf_refine_f32) can re-use them. */
    float* rayfeat_out;
    /* arithmetic of the decoders' matrix products (layers 1-3 of the per-point kernel):
     *   LIDF_PRECISION_F32    (0, default) f32 inputs on v_mfma_f32_32x32x2_f32
     *   LIDF_PRECISION_F16X3  (1) every f32 operand split into two f16 pieces, three
     *       v_mfma_f32_32x32x16_f16 products per term, f32 accumulation: f32-level accuracy
     *       (relative error of a product <= 2^-22) as long as |activations| < 65504        */
    int32_t precision;
    /* optional: the decoders' weights already packed by lidf_query_pack_f32 for the SAME multires,
     * multires_views and precision (device memory, read-only here). NULL = pack inside this call
     * (3 small launches); eval loops pack once per checkpoint, training loops once per optimizer
     * step. The LidfDecoder pointers are then only consulted for n_iter / init_offset / sigmoid. */
    const void* packed;
    /* Opt-in (0 = off, the reference's data flow; f32 only): run the offset decoder on the SELECTED pair
     * of every ray only. get_pred (models/pipeline.py:427-466) evaluates offset_dec on every pair but
     * everything downstream — pred_pos (:453-454), the depth map, the losses and statistics of
     * compute_loss, stage 2 — reads pair_pred_pos through max_pair_id alone. With this flag prob_dec runs
     * on all pairs, the per-ray softmax / arg-max follows, and offset_dec (two of the three decoder passes)
     * runs on one pair per ray: pred_prob, softmax, max_pair_id, pred_pos and depth are bit-identical to
     * the default; pred_offset / pair_pred_pos hold values ONLY at the selected pairs (other entries are
     * not written). About half the matrix work on a real frame (3.6 pairs per ray). (ABI 7)            */
    int32_t offsets_selected;
} LidfQueryArgs;

/* Packed weights of the fused query: the parameters of prob_dec / offset_dec re-ordered into the
 * streams the kernels consume (the reference re-reads nn.Linear storage every call; here the
 * re-ordering is hoisted out of the per-frame path). Valid until a parameter changes.           */
size_t lidf_query_pack_bytes(void);
int lidf_query_pack_f32(const LidfDecoder* prob, const LidfDecoder* off, int multires,
                        int multires_views, int precision, void* packed, size_t packed_bytes,
                        lidf_stream_t stream);

/* Guarded packing — the safe default for callers that keep `packed` across calls. nn.Parameter storage
 * can be rewritten without any host-visible trace (`p.data.mul_()` does not bump torch's version
 * counter; SURVEY §8b "Ownership": no cache keyed on a pointer without a content check), so the check
 * is made on the device: one launch forms a 64-bit fingerprint of every parameter buffer (plus the
 * scalar fields of the structs), the last block compares it with the fingerprint `packed` was built
 * from, and the pack kernels that follow on the stream return at once when nothing changed. No host
 * synchronisation, a few microseconds per call. `guard`: lidf_pack_guard_bytes() bytes of device
 * memory, zero-filled ONCE when `packed` is allocated and then owned by these calls (one guard per
 * packed blob; calls that share a guard must be ordered on the device).
 * Same for lidf_pointnet_pack_guarded_f32 / lidf_refine_pack_guarded_f32 below.                   */
size_t lidf_pack_guard_bytes(void);
int lidf_query_pack_guarded_f32(const LidfDecoder* prob, const LidfDecoder* off, int multires,
                                int multires_views, int precision, void* packed, size_t packed_bytes,
                                void* guard, lidf_stream_t stream);

/* grid_floats = batch*32*height*width makes room for the optional 4x4 box-sum image that turns the
 * ROIAlign of unclamped boxes into 4 gathers per channel; 0 = minimal workspace (general path). */
size_t lidf_query_workspace_bytes(int64_t n_rays, int64_t n_vox, int64_t grid_floats);
int lidf_query_f32(const LidfQueryArgs* args, lidf_stream_t stream);
/* Instrumented variant for benchmarks: the same call, with two hipEvent_t recorded on `stream`
 * immediately before and after the per-point decoder kernel (the dominant launch); either may be
 * NULL. Not part of the reference's interface.                                                */
int lidf_query_profile_f32(const LidfQueryArgs* args, void* ev_points_begin, void* ev_points_end,
                           lidf_stream_t stream);

/* Per-ray ROIAlign feature (torchvision.ops.roi_align, output 2x2, aligned=True, called at
 * models/pipeline.py:374-387 and :954-967) + direction embedding, exposed on its own because
 * stage 2 (RefineNet.get_pred_refine) re-uses it. rayfeat: [R, 128 + 3 + 6*multires_views],
 * row = [c*4 + ph*2 + pw for 32 channels | embed(dir)]. workspace is optional (NULL / 0): with
 * lidf_ray_features_workspace_bytes the unclamped boxes are pooled from a 4x4 box-sum image.   */
size_t lidf_ray_features_workspace_bytes(int batch, int height, int width, int64_t n_rays);
int lidf_ray_features_f32(const float* feat_grid, int batch, int height, int width,
                          const float* ray_dir, const int32_t* ray_pix, const int32_t* ray_bid,
                          int64_t n_rays, int roi_inp_bbox, int multires_views,
                          float* rayfeat, void* workspace, size_t workspace_bytes,
                          lidf_stream_t stream);

/* Per-ray softmax / argmax / select on its own (torch_scatter.scatter_softmax + scatter_max at
 * models/pipeline.py:442-454). Ties: lowest pair index. Empty ray: id = P, pos = 0.          */
int lidf_ray_reduce_f32(const float* pred_prob, const float* pair_pred_pos,
                        const int32_t* pair_off, int64_t n_rays, int64_t n_pairs,
                        const int32_t* ray_bid, const int32_t* ray_flat, int64_t hw,
                        float* softmax, int64_t* max_pair_id, float* pred_pos, float* depth,
                        lidf_stream_t stream);

/* ---- Ray generation ----------------------------------------------------------------------
 * Replaces the dense part of LIDF.get_miss_ray (models/pipeline.py:208-220):
 * ray_dir[b,y,x] = normalize(x-cx, (y-cy)*fx/fy, fx). intr: [B,4] = (fx,fy,cx,cy) f32.       */
int lidf_ray_dirs_f32(const float* intr, int batch, int height, int width, float* ray_dir,
                      lidf_stream_t stream);

/* ---- Miss-ray selection -------------------------------------------------------------------
 * Replaces LIDF.get_miss_ray (models/pipeline.py:203-269), eval flavour (the train-only random
 * window of :232-254 is the caller's slice of the outputs): miss_idx = nonzero(mask.view(bs,-1))
 * in (image, pixel) order, then per selected pixel the image index, the flat pixel index y*w+x,
 * the unit ray direction (pipeline.py:215-219) and the integer pixel (x, y).
 * mask: [batch*height*width] elements of mask_dtype (the reference passes float masks; NaN is
 * non-zero, as for torch.nonzero). Two calls, because the caller sizes the outputs:
 *   lidf_miss_ray_count  -> n_rays (device int32[1]); keeps block offsets in the workspace
 *   lidf_miss_ray_fill   -> same mask + the SAME workspace; every output may be NULL:
 *       ray_bid/ray_flat [R] i32, ray_pix [R,2] i32, ray_dir [R,3] f32 (what lidf_query_f32 takes)
 *       miss_bid/miss_flat_img_id [R] i64, miss_img_ind [R,2] i64 (the reference's data_dict dtypes) */
#define LIDF_MASK_F32 0
#define LIDF_MASK_U8 1
#define LIDF_MASK_I32 2
#define LIDF_MASK_I64 3
size_t lidf_miss_ray_workspace_bytes(int64_t n_pixels);
int lidf_miss_ray_count(const void* mask, int mask_dtype, int64_t n_pixels, int32_t* n_rays,
                        void* workspace, size_t workspace_bytes, lidf_stream_t stream);
int lidf_miss_ray_fill_f32(const void* mask, int mask_dtype, const float* intr, int batch, int height,
                           int width, const void* workspace, size_t workspace_bytes,
                           int32_t* ray_bid, int32_t* ray_flat, int32_t* ray_pix, float* ray_dir,
                           int64_t* miss_bid, int64_t* miss_flat_img_id, int64_t* miss_img_ind,
                           lidf_stream_t stream);

/* ---- Ray / voxel slab test ---------------------------------------------------------------
 * Dense drop-in for extensions/ray_aabb (ray_aabb_cuda_kernel.cu:10-126): mask [V,R] i32 and
 * dist [V,R,2] f32 must be zero-filled by the caller (reference: torch::zeros).             */
int lidf_ray_aabb_dense_f32(const float* ray_dir, const float* voxel_bound,
                            const int32_t* ray_bid, const int32_t* voxel_bid,
                            int64_t n_rays, int64_t n_vox, int32_t* mask, float* dist,
                            lidf_stream_t stream);
/* Compact ray-major form of the same test: pass 1 writes the hit count per ray, the caller
 * turns counts into pair_off (exclusive scan, lidf_exclusive_scan_i32), pass 2 fills the
 * pairs of each ray in ascending voxel order (= the reference's nonzero() order within a ray,
 * models/pipeline.py:283).                                                                  */
int lidf_ray_aabb_count_f32(const float* ray_dir, const float* voxel_bound,
                            const int32_t* ray_bid, const int32_t* voxel_bid,
                            int64_t n_rays, int64_t n_vox, int32_t* count,
                            lidf_stream_t stream);
int lidf_ray_aabb_fill_f32(const float* ray_dir, const float* voxel_bound,
                           const int32_t* ray_bid, const int32_t* voxel_bid,
                           int64_t n_rays, int64_t n_vox, const int32_t* pair_off,
                           int32_t* pair_ray, int32_t* pair_vox, float* pair_t,
                           lidf_stream_t stream);
/* The same compact list for voxels that are cells of a regular grid — what
 * LIDF.get_occ_vox_bound builds (models/pipeline.py:162-201: bound_min = xmin + coord*part_size,
 * coord = occ_vox_global_coord), replacing the dense V x R test of pipeline.py:277-285 by a walk
 * over the cells a ray can meet. voxel_coord [V,3] i32 = the cell index of every voxel
 * (0 <= coord < (rx,ry,rz); voxels outside are ignored), voxel_bid < batch; all voxels that share
 * a frame and an axis index must carry the same bounds on that axis (true by construction).
 * `build` fills `grid` (lidf_ray_aabb_grid_workspace_bytes) with a cell -> voxel table and the
 * per-axis bound tables; `count` / `fill` are the two passes of the compact form above. Hits,
 * t_enter and t_leave are bit-identical to lidf_ray_aabb_count/fill_f32 (the same products of the
 * voxels' own bounds, compared in the same order); within a ray the pairs ascend in (x,y,z) cell
 * order = ascending voxel index for a list sorted like torch.unique sorts it.
 * PRECONDITIONS (not checked; lidf_voxelize_f32 / get_occ_vox_bound guarantee them): (frame, cell)
 * is unique over the voxel list — the cell table keeps ONE voxel per cell, a duplicate would be
 * dropped silently — and the list is sorted by (frame, x, y, z), otherwise pair_vox does not ascend
 * within a ray. For an arbitrary voxel list use lidf_ray_aabb_count/fill_f32.                     */
size_t lidf_ray_aabb_grid_workspace_bytes(int32_t batch, int32_t rx, int32_t ry, int32_t rz);
int lidf_ray_aabb_grid_build_f32(const float* voxel_bound, const int32_t* voxel_bid,
                                 const int32_t* voxel_coord, int64_t n_vox, int32_t batch,
                                 int32_t rx, int32_t ry, int32_t rz, void* grid, size_t grid_bytes,
                                 lidf_stream_t stream);
int lidf_ray_aabb_grid_count_f32(const float* ray_dir, const int32_t* ray_bid, int64_t n_rays,
                                 int32_t batch, int32_t rx, int32_t ry, int32_t rz,
                                 const void* grid, size_t grid_bytes, int32_t* count,
                                 lidf_stream_t stream);
int lidf_ray_aabb_grid_fill_f32(const float* ray_dir, const int32_t* ray_bid, int64_t n_rays,
                                int32_t batch, int32_t rx, int32_t ry, int32_t rz,
                                const void* grid, size_t grid_bytes, const int32_t* pair_off,
                                int32_t* pair_ray, int32_t* pair_vox, float* pair_t,
                                lidf_stream_t stream);
/* out[0]=0, out[i+1]=out[i]+in[i]; out has n+1 entries. Single-launch (n up to 2^31-1).      */
size_t lidf_exclusive_scan_workspace_bytes(int64_t n);
int lidf_exclusive_scan_i32(const int32_t* in, int64_t n, int32_t* out,
                            void* workspace, size_t workspace_bytes, lidf_stream_t stream);

/* ---- Point / voxel inside test -------------------------------------------------------------
 * Dense drop-in for extensions/pcl_aabb (pcl_aabb_cuda_kernel.cu:10-80): mask [V,Np] i32,
 * zero-filled by the caller.                                                                 */
int lidf_pcl_aabb_dense_f32(const float* pcl_pos, const float* voxel_bound,
                            const int32_t* pcl_bid, const int32_t* voxel_bid,
                            int64_t n_pts, int64_t n_vox, int32_t* mask, lidf_stream_t stream);
/* Compact form used by stage 2 (models/pipeline.py:939-944): for every point the LARGEST index
 * of a voxel of the same image that contains it (inclusive bounds), or -1.                  */
int lidf_pcl_aabb_last_f32(const float* pcl_pos, const float* voxel_bound,
                           const int32_t* pcl_bid, const int32_t* voxel_bid,
                           int64_t n_pts, int64_t n_vox, int32_t* last_vox,
                           lidf_stream_t stream);

/* ---- Occupied-voxel build -------------------------------------------------------------------
 * Replaces utils/point_utils.py:12-76 batch_get_occupied_idx(overlap=False) and
 * LIDF.get_occ_vox_bound (models/pipeline.py:162-201): points [N,3] f32 with image index [N] i32
 * -> occupied voxels in torch.unique's sorted (bid,x,y,z) order and, for the points inside the
 * grid (original order kept), their voxel (revidx), index (valid_v_pid) and voxel-relative
 * coordinate. xmin: host float[3] lower grid corner (already widened by half a voxel,
 * pipeline.py:170), res: host int[3] cells per axis, crop: voxel size.
 * Outputs must be sized for the worst case: occ_bid_coord [batch*res0*res1*res2, 4] i32,
 * voxel_bound [same, 6] f32, valid_pid / revidx [N] i32, rel_coord [N,3] f32.
 * counts (device int32[2]) receives {V, number of points inside the grid}.                     */
size_t lidf_voxelize_workspace_bytes(int64_t n_pts, int64_t n_cells);
int lidf_voxelize_f32(const float* xyz, const int32_t* bid, int64_t n_pts, int batch,
                      const float* xmin, const int32_t* res, float crop, int32_t* occ_bid_coord,
                      float* voxel_bound, int32_t* valid_pid, int32_t* revidx, float* rel_coord,
                      int32_t* counts, void* workspace, size_t workspace_bytes,
                      lidf_stream_t stream);

/* ---- PointNet2Stage ------------------------------------------------------------------------
 * Replaces PointNet2Stage.forward (models/pointnet.py:22-38) incl. its two
 * torch_scatter.scatter(..., reduce='max') poolings, for the shipped dimensions
 * (input_channels 6, gf_dim 32, output_channels 128; other widths: lidf_linear_f32 with its scatter-max
 * epilogue, layer by layer). Weights: nn.Linear storage [out,in].
 * inp [N,6] f32, vox [N] i32 (vox2point_idx: voxel of every point, < n_vox; a negative entry
 * leaves that point out of both poolings) -> out [n_vox,128].                                    */
typedef struct LidfPointNet {
    const float *w_p1, *b_p1; /* point_lin1 [32,6]    */
    const float *w_p2, *b_p2; /* point_lin2 [64,32]   */
    const float *w_v1, *b_v1; /* vox_lin1   [64,64]   */
    const float *w_p3, *b_p3; /* point_lin3 [128,128] */
    const float *w_p4, *b_p4; /* point_lin4 [128,128] */
    const float *w_v2, *b_v2; /* vox_lin2   [128,128] */
    /* optional: the seven weight streams already packed by lidf_pointnet_pack_f32 (device memory);
     * NULL = pack inside every call (7 small launches). Valid until a parameter changes.       */
    const void* packed;
} LidfPointNet;
size_t lidf_pointnet_pack_bytes(void);
int lidf_pointnet_pack_f32(const LidfPointNet* w, void* packed, size_t packed_bytes,
                           lidf_stream_t stream);
int lidf_pointnet_pack_guarded_f32(const LidfPointNet* w, void* packed, size_t packed_bytes,
                                   void* guard, lidf_stream_t stream);
size_t lidf_pointnet_workspace_bytes(int64_t n_pts, int64_t n_vox);
int lidf_pointnet_f32(const LidfPointNet* w, const float* inp, const int32_t* vox, int64_t n_pts,
                      int64_t n_vox, float* out, void* workspace, size_t workspace_bytes,
                      lidf_stream_t stream);

/* ---- Stage-2 refinement query -------------------------------------------------------------
 * One iteration of RefineNet.get_pred_refine (models/pipeline.py:922-1030), eval flavour
 * (no perturbation; refine.use_all_pix True or, with pnet_select, False): end voxel of every ray, PointNet
 * over (valid points + predicted points), [voxel feature | ROI feature | embed(pos) | embed(dir)]
 * -> IEF (D = 256 + 3+6*multires + 3+6*multires_views) -> pred_pos + offset * ray_dir.
 * RefineNet.forward (:1032-1041) calls it refine.forward_times times, feeding pred_pos_out back. */
typedef struct LidfRefineArgs {
    int64_t n_rays;
    const float* ray_dir;        /* [R,3] */
    const int32_t* ray_bid;      /* [R]   */
    const int32_t* ray_flat;     /* [R]  y*w+x */
    const float* pred_pos;       /* [R,3] position to refine (stage-1 pred_pos, or previous output) */
    const int64_t* max_pair_id;  /* [R]  stage-1 arg-max pair, n_pairs for a ray without pairs */
    const int32_t* pair_vox;     /* [P]  */
    int64_t n_pairs;
    int64_t n_vox;
    const float* voxel_bound;    /* [V,6] */
    const int32_t* voxel_bid;    /* [V]   */
    const float* rgb_img;        /* [B,3,h,w] */
    int32_t batch, height, width;
    const float* rayfeat;        /* [R, 128 + 3+6*multires_views] from lidf_ray_features_f32 */
    int64_t n_valid;
    const float* valid_inp;      /* [Nv,6] PointNet input of the valid points (rel coord | rgb) */
    const int32_t* valid_vox;    /* [Nv]   revidx */
    const LidfPointNet* pnet;    /* refine pnet_model */
    const LidfDecoder* off;      /* refine offset_dec (IEF or IMNet), d_in D */
    int32_t multires, multires_views;
    int32_t pos_rel;             /* refine.intersect_pos_type == 'rel' */
    int32_t pnet_pos_rel;        /* refine.pnet_pos_type == 'rel' */
    float offset_range0, offset_range1; /* refine.offset_range */
    float* pred_pos_out;         /* [R,3] pred_pos_refine */
    int32_t* end_voxel_id;       /* [R] optional output */
    void* workspace;
    size_t workspace_bytes;
    int32_t precision;           /* LIDF_PRECISION_F32 (0) / LIDF_PRECISION_F16X3: the refine IEF  */
    /* mask_type 'all' with refine.use_all_pix == False (models/pipeline.py:987-996): [R] bytes, only
     * rays with a non-zero entry (inp_zero_mask = 1 - valid_mask at the ray's pixel) feed their
     * predicted point back into the PointNet; every ray is still refined. NULL = all rays
     * (refine.use_all_pix == True, the shipped configs).                                         */
    const uint8_t* pnet_select;
    /* f32 precision: the IEF's packed weight streams (lidf_refine_pack_f32, built once per
     * parameter version) or NULL = packed inside the call, into the workspace.                  */
    const void* packed;
    /* optional (f32): [R,256] scratch that carries the per-ray part of the decoder's layer 1
     * (W1[:, ROI | direction columns] rayfeat[r]: the same for every iteration on these rays) from one
     * call to the next: the call with ray_l1_ready == 0 fills it, later calls with ray_l1_ready != 0
     * read it. NULL = formed inside every call.                                                     */
    float* ray_l1;
    int32_t ray_l1_ready;
    /* optional (ABI 8): the end voxel through a cell table instead of testing every ray against every voxel
     * (models/pipeline.py:939-944: pcl_aabb + scatter max = the largest voxel of the ray's image whose box
     * contains pred_pos). For voxel lists that are cells of a regular grid — what LIDF.get_occ_vox_bound
     * (:162-201) / lidf_voxelize_f32 build: voxel j of image voxel_bid[j] is cell voxel_coord[j] of the
     * grid_res cells that start at grid_xmin with edge grid_part, and voxel_bound[j] is that cell's box up to
     * rounding. The call scatters the voxels into cell_table (cell -> largest voxel index, -1 = empty),
     * estimates the cell of a point from (p - grid_xmin) / grid_part and applies the reference's inclusive
     * test to the STORED bounds of the voxels in the 27 cells around it: the same end_voxel_id as the
     * every-voxel test (tests/test_refine_gpu.py), O(R) instead of O(R V). A NaN coordinate is "inside" every
     * voxel of its image for the reference's predicate; such a ray walks the list.
     * voxel_coord NULL = test every voxel. cell_table: caller scratch of batch * res0 * res1 * res2 int32;
     * cell_table_ready != 0: it already holds this voxel list (a later iteration on the same frame).   */
    const int32_t* voxel_coord;  /* [V,3] */
    int32_t grid_res[3];
    float grid_xmin[3];
    float grid_part;
    int32_t* cell_table;
    int32_t cell_table_ready;
} LidfRefineArgs;
size_t lidf_refine_workspace_bytes(int64_t n_rays, int64_t n_valid, int64_t n_vox);
size_t lidf_refine_pack_bytes(int32_t multires, int32_t multires_views);
int lidf_refine_pack_f32(const LidfDecoder* off, int32_t multires, int32_t multires_views,
                         void* packed, size_t packed_bytes, lidf_stream_t stream);
int lidf_refine_pack_guarded_f32(const LidfDecoder* off, int32_t multires, int32_t multires_views,
                                 void* packed, size_t packed_bytes, void* guard, lidf_stream_t stream);
int lidf_refine_f32(const LidfRefineArgs* args, lidf_stream_t stream);
/* Instrumented variant for benchmarks: the same call with hipEvent_t recorded on `stream` around the
 * PointNet2Stage pass and around the IEF rows kernel (any may be NULL). Not part of the reference's
 * interface.                                                                                       */
int lidf_refine_profile_f32(const LidfRefineArgs* args, void* ev_pnet_begin, void* ev_pnet_end,
                            void* ev_ief_begin, void* ev_ief_end, lidf_stream_t stream);

/* ---- The evaluation path of a batch of frames in ONE call, without a host round trip --------------
 * LIDF.forward, exp_type 'test' (models/pipeline.py:652-717: prepare_data :91-133, get_valid_points
 * :135-160, get_occ_vox_bound :162-201, get_miss_ray :203-269, compute_ray_aabb :271-296, the
 * PointNet2Stage voxel embedding :399-408, get_embedding + get_pred :338-466, the depth map :593-596)
 * followed, when refine_times > 0, by RefineNet.forward (:1032-1041: refine_times x get_pred_refine).
 *
 * The reference sizes every compacted list on the host (torch.nonzero / torch.unique / .item()):
 * four device -> host round trips per frame, during which the GPU idles. Here every list lives in a
 * caller-provided buffer sized for the worst case and its length stays on the device: `counts`
 * receives the lengths, every launch is sized for the capacity and reads its count on the device.
 * The call only enqueues work with launch parameters that depend on (batch, height, width,
 * max_pairs, lds_voxels) alone — it can be captured in a hipGraph and replayed for every frame.
 *
 * Capacities: rays / valid points <= batch*height*width; voxels <= batch*res0*res1*res2; pairs <=
 * max_pairs (a ray crosses at most res0+res1+res2-2 cells of the grid; when a frame has more pairs
 * than max_pairs the list is cut, counts[7] bit 0 is set and the cut rays' results are invalid).
 * Output arrays must hold the capacities; entries beyond the counts are unspecified.
 * The packed weight streams are mandatory (lidf_*_pack_guarded_f32).                              */
#define LIDF_FRAME_COUNTS 8
#define LIDF_FC_RAYS 0        /* R:   queried pixels (miss rays)                               */
#define LIDF_FC_PAIRS 1       /* P:   (ray, occupied voxel) pairs                              */
#define LIDF_FC_VOX 2         /* V:   occupied voxels                                          */
#define LIDF_FC_VALID_IN 3    /* NV:  valid points inside the grid (rows of pnet_inp)          */
#define LIDF_FC_VALID_PIX 4   /* NV0: valid pixels                                             */
#define LIDF_FC_VALID_SEL 5   /* NVS: valid points kept by valid_stride                        */
#define LIDF_FC_PNET_REFINE 6 /* NV + R: points of the stage-2 PointNet                        */
#define LIDF_FC_OVERFLOW 7    /* bit 0: pairs cut at max_pairs                                 */
typedef struct LidfFrameArgs {
    int32_t batch, height, width;
    /* the batch (datasets/cleargrasp_dataset.py:165-180 keys), device, f32 */
    const float* rgb;          /* [B,3,h,w]                                                       */
    const float* xyz_corrupt;  /* [B,3,h,w]                                                       */
    const float* valid_mask;   /* [B,h,w] non-zero = the pixel's measured point is used (mask_type
                                  'all': depth_corrupt itself — valid <=> depth_corrupt != 0)     */
    const float* miss_mask;    /* [B,h,w] non-zero = query this pixel (pred_mask); NULL = every pixel */
    const float* intr;         /* [B,4] fx, fy, cx, cy                                            */
    const float* feat_grid;    /* [B,32,h,w] full_rgb_feat                                        */
    /* voxel grid (LIDF.get_occ_vox_bound): lower corner already widened by half a voxel          */
    float xmin[3];
    int32_t res[3];
    float part_size;
    int32_t valid_stride;      /* >= 1: every valid_stride-th valid pixel feeds the voxels        */
    /* stage 1 */
    const LidfPointNet* pnet;  /* ->packed mandatory                                              */
    const LidfDecoder* prob;
    const LidfDecoder* off;
    const void* packed_query;  /* lidf_query_pack(_guarded)_f32 blob, f32 precision               */
    int32_t multires, multires_views, roi_inp_bbox, pos_rel;
    float offset_range0, offset_range1;
    /* stage 2 (refine_times == 0: skipped) */
    int32_t refine_times;
    const LidfPointNet* pnet_refine;   /* ->packed mandatory                                      */
    const LidfDecoder* off_refine;
    const void* packed_refine;         /* lidf_refine_pack(_guarded)_f32 blob                     */
    int32_t refine_pos_rel, refine_pnet_pos_rel, refine_use_all_pix;
    float refine_offset_range0, refine_offset_range1;
    int32_t precision;         /* LIDF_PRECISION_F32 (0) / LIDF_PRECISION_F16X3: both decoders and the stage-2
                                  IEF (packed_query must be packed for the same precision; the split-f16
                                  IEF of stage 2 packs inside the call, packed_refine is not read)        */
    /* capacities */
    int64_t max_pairs;
    int32_t lds_voxels;        /* bound of the PointNet's LDS pooling tables (<= 288). More occupied voxels:
                                  batch >= 2 walks the points grouped by voxel (counting sort + windowed
                                  LDS tables), batch == 1 takes per-point global atomic maxima. 0 = 128
                                  (two workgroups per CU); 288 costs 1.5 % and covers any real frame  */
    /* outputs (device; capacity in brackets: N = batch*height*width, C = batch*res0*res1*res2)   */
    int32_t* counts;           /* [LIDF_FRAME_COUNTS]                                             */
    int32_t *valid_bid, *valid_flat;       /* [N]   image / flat pixel of the selected valid points */
    float *valid_xyz, *valid_rgb;          /* [N,3]                                               */
    int32_t* occ_bid_coord;                /* [C,4] (bid, x, y, z) of the occupied voxels          */
    float* voxel_bound;                    /* [C,6]                                               */
    int32_t *valid_v_pid, *revidx;         /* [N + N] in-grid points: index into the selected points,
                                              voxel; revidx has room for the N predicted points of
                                              stage 2 behind the valid ones                        */
    float* valid_v_rel_coord;              /* [N,3]                                               */
    float* pnet_inp;                       /* [N + N, 6] cat(rel coord, rgb) (+ stage-2 rows)      */
    float* occ_voxel_feat;                 /* [C,128]                                             */
    int32_t *ray_bid, *ray_flat, *ray_pix; /* [N], [N], [N,2]                                      */
    float* ray_dir;                        /* [N,3]                                               */
    int32_t* pair_off;                     /* [N+1]                                               */
    int32_t *pair_ray, *pair_vox;          /* [max_pairs]                                         */
    float* pair_t;                         /* [max_pairs,2]                                       */
    float *pred_offset, *pred_prob, *pred_prob_softmax; /* [max_pairs]                            */
    float* pair_pred_pos;                  /* [max_pairs,3]                                       */
    int64_t* max_pair_id;                  /* [N]                                                 */
    float* pred_pos;                       /* [N,3]                                               */
    float* rayfeat;                        /* [N, 128 + 3 + 6*multires_views]                     */
    float* pred_depth;                     /* [B,h,w] xyz_corrupt z with the rays' pixels replaced */
    float* pred_pos_refine;                /* [N,3]   (refine_times > 0)                          */
    int32_t* end_voxel_id;                 /* [N]                                                 */
    float* pred_depth_refine;              /* [B,h,w]                                             */
    void* workspace;
    size_t workspace_bytes;
    /* optional: the valid points as an explicit list instead of every valid_stride-th valid pixel —
     * what LIDF.get_valid_points keeps when grid.valid_sample_num != -1 (models/pipeline.py:143-146:
     * utils/point_utils.py sample_valid_points, a random block sampler that is host code upstream of
     * the path). n_valid_idx > 0: point j is pixel valid_idx_flat[j] of image valid_idx_bid[j], in
     * this order (duplicates allowed, as the sampler produces them for sparse frames);
     * n_valid_idx <= batch*height*width (image / pixel indices outside the batch are clamped into it,
     * where the reference's index_select would raise); valid_mask is then only read by refine_use_all_pix == 0;
     * counts[LIDF_FC_VALID_PIX] = counts[LIDF_FC_VALID_SEL] = n_valid_idx. (ABI 6)                  */
    const int32_t* valid_idx_bid;
    const int32_t* valid_idx_flat;
    int64_t n_valid_idx;
    /* weight streams of ALL modules of the frame kept and validated by the call itself (ABI 7).
     * pack_mode LIDF_FRAME_PACK_CALLER (0): the caller's blobs above (pnet->packed, packed_query,
     *   pnet_refine->packed, packed_refine), each validated by its own lidf_*_pack_guarded_f32 call
     *   (4 fingerprint + 6 pack launches per frame);
     * LIDF_FRAME_PACK_GUARDED (1): pack_blob (lidf_frame_pack_bytes() bytes, owned by the caller, only
     *   ever touched by launches of ONE stream) holds the streams of every module, pack_guard
     *   (lidf_frame_pack_guard_bytes() bytes, zero-filled once) their fingerprints: ONE fingerprint
     *   launch over all parameter buffers + two early-exit pack launches per frame; the caller's blob
     *   pointers above are not read;
     * LIDF_FRAME_PACK_TRUSTED (2): pack_blob as the last GUARDED call left it, no check (parameters known
     *   not to have changed since: eval loops re-validate every so many frames).                        */
    void* pack_blob;
    size_t pack_blob_bytes;
    void* pack_guard;
    int32_t pack_mode;
    int32_t offsets_selected;   /* as LidfQueryArgs.offsets_selected (opt-in; f32) */
    /* optional second stream (ABI 7): launches of a frame that fill a fraction of the device each run side by
     * side. On `aux_stream`: the weight-stream guard (pack_mode GUARDED: fingerprint + early-exit packs), the
     * box sums of the feature map, then — once the rays exist — the per-ray RoIAlign features; on `stream`
     * meanwhile: zeroed scratch, frame head, voxel list, ray / voxel pairs, PointNet rows and (after the
     * guard) PointNet2Stage; with stage 2, the per-ray layer-1 table of its decoder then runs on
     * `aux_stream` beside the per-point kernel instead of inside the query's layer-1 launch. `ev_fork` is
     * recorded on `stream` three times (start of the frame, rays exist, layer-1 launch done) and awaited by
     * `aux_stream` each time; `ev_join` is recorded on `aux_stream` three times (guard done, per-ray
     * features done, stage-2 table done) and awaited by `stream` before the first weight stream is read /
     * before the layer-1 tables / before stage 2. Results are bit-identical. All three NULL = one stream (the default). ev_fork / ev_join: two
     * hipEvent_t of the caller (hipEventDisableTiming is enough), not shared with a frame in flight on
     * another stream. Capturable (the side stream joins the capture through the events); measured on
     * MI355X: eager 1.71 -> 1.66 ms per 240x320 frame, under a replayed graph no gain.                */
    lidf_stream_t aux_stream;
    void* ev_fork;
    void* ev_join;
    /* Error exits with a side stream (ABI 8): once the first fork is recorded, ANY non-zero return of the call
     * first records ev_join on aux_stream and makes `stream` wait for it — the side stream's queued launches
     * are ordered before whatever the caller enqueues next on `stream` (no open fork under capture, no launch
     * still reading feat_grid / writing rayfeat after the caller frees them).
     * fail_after (test hook, 0 = off): the call returns LIDF_ERR_HIP right after enqueuing stage k, exactly as
     * a failed launch there would — 1 frame head (second fork open), 2 pairs / PointNet rows, 3 PointNet,
     * 4 query. tests/test_frame_gpu.py forces mid-frame failures with it. The field is honoured ONLY in a process
     * whose environment has LIDF_TEST_FAULTS=1; everywhere else it is ignored, so a caller compiled against an
     * older, shorter struct that did not zero the tail cannot trip it. (General rule of this header: structs grow
     * at the end between ABI versions — memset the whole struct to zero before filling it.)               */
    int32_t fail_after;
    /* profile_events (ABI 11; benchmarks only, NULL = none): six hipEvent_t recorded on `stream` around the three
     * matrix launches of a frame whose fraction of the peak a record states — [0],[1] the per-point kernel
     * (as lidf_query_profile_f32 does), [2],[3] / [4],[5] the stage-2 decoder of refine iterations 0 / 1 (as
     * lidf_refine_profile_f32 does). Entries may be NULL. HIP-event durations of one-stream frames only: with
     * the side stream the launches beside them stretch what the events bracket.                        */
    void* const* profile_events;
} LidfFrameArgs;
#define LIDF_FRAME_PACK_CALLER 0
#define LIDF_FRAME_PACK_GUARDED 1
#define LIDF_FRAME_PACK_TRUSTED 2
size_t lidf_frame_pack_bytes(void);
size_t lidf_frame_pack_guard_bytes(void);
size_t lidf_frame_workspace_bytes(int32_t batch, int32_t height, int32_t width, const int32_t* res,
                                  int64_t max_pairs, int32_t lds_voxels, int32_t refine_times);
int lidf_frame_f32(const LidfFrameArgs* args, lidf_stream_t stream);
/* The two events of LidfFrameArgs.ev_fork / ev_join, created and destroyed by the runtime this library is
 * linked against (a binding that opened its own copy of the HIP runtime would hand over foreign handles).
 * hipEventDisableTiming. Host calls; *out receives the hipEvent_t. (ABI 8)                               */
int lidf_event_create(void** out);
int lidf_event_destroy(void* event);

/* ---- One linear layer of any width ------------------------------------------------------------
 * out[r, 0:nout] = act( x[r, 0:k] . w[0:nout, 0:k]^T + b  (+ addrows[addidx[r], 0:nout]) ), torch.nn.Linear's
 * weight layout ([nout, k], row stride ldw), through the same f32 matrix-instruction kernel the fixed-width
 * paths use (lidf_linear.hip; 256 output columns per launch). It is what the Python modules are built from
 * when a width differs from the shipped configuration (models/implicit_net.py IMNet / IEF with
 * gf_dim != 64 or out_dim != 1, models/pointnet.py PointNet2Stage with gf_dim != 32 or
 * output_channels != 128): those run layer by layer instead of as one register-chained launch.
 *   act     0 = none, 1 = max(v, slope * v)  (slope 0: ReLU; 0.02: the decoders' leaky ReLU)
 *   addrows optional gathered term (row stride ld_add >= nout), added before the activation
 *   out     optional [n, nout] (row stride ld_out)
 *   pool    optional [*, nout] (row stride ld_pool), pool[poolidx[r], c] = max(pool[...], value): the
 *           scatter-max of models/pointnet.py:27,35; needs act = 1, slope = 0, nout % 32 == 0 and a
 *           zero-initialised table (values are >= 0, a voxel without points keeps 0 as torch_scatter does)
 *   workspace  lidf_linear_workspace_bytes(k)                                                        */
size_t lidf_linear_workspace_bytes(int32_t k);
int lidf_linear_f32(const float* x, int64_t ldx, int64_t n, int32_t k, const float* w, int64_t ldw,
                    const float* b, int32_t nout, int32_t act, float slope, const float* addrows,
                    const int32_t* addidx, int64_t ld_add, float* out, int64_t ld_out, float* pool,
                    const int32_t* poolidx, int64_t ld_pool, void* workspace, size_t workspace_bytes,
                    lidf_stream_t stream);

/* The same layer with TWO gathered terms, out = act(x w^T + b + addrows[addidx[r]] + addrows2[addidx2[r]]): layer 1 of
 * a decoder at any width in its factorised form (models/pipeline.py:431-433 concatenates per-voxel, per-ray and
 * per-pair columns; W1 x = W1[:, voxel columns] vox_feat[v] + W1[:, ray columns] rayfeat[r] + W1[:, pair columns]
 * PE(p)), so that only the pair columns are multiplied per pair and no [P, D] row is formed. Both terms are
 * required (one term: lidf_linear_f32); no pooling epilogue. (ABI 10)                                          */
int lidf_linear_gather2_f32(const float* x, int64_t ldx, int64_t n, int32_t k, const float* w, int64_t ldw,
                            const float* b, int32_t nout, int32_t act, float slope, const float* addrows,
                            const int32_t* addidx, int64_t ld_add, const float* addrows2,
                            const int32_t* addidx2, int64_t ld_add2, float* out, int64_t ld_out,
                            void* workspace, size_t workspace_bytes, lidf_stream_t stream);

/* Weight gradient of such a layer: c[i, j] += sum_r a[r, i] * b[r, j] (a = dL/d(pre-activation) [n, m],
 * b = the layer's input rows [n, n_cols]: c is dL/dW in nn.Linear's [out, in] layout), db[i] += sum_r a[r, i]
 * (optional). c / db are accumulated into: zero them for a fresh gradient. With workspace
 * (lidf_wgrad_workspace_bytes() bytes) the partial sums of the row slices are reduced in a fixed order
 * (run-to-run identical) — for m >= 32 and n_cols >= 4, the block path. Narrower layers (m < 32: a 1-wide
 * output layer; n_cols < 4: the IEF's offset_enc = Linear(1, 16)) and calls whose partial blocks do not
 * fit the workspace take the column kernel, which adds its row slices with float atomics once n > 512:
 * their gradients are correct to rounding but NOT bit-reproducible run to run. workspace NULL: float
 * atomics throughout.                                                                                   */
size_t lidf_wgrad_workspace_bytes(void);
int lidf_wgrad_f32(const float* a, int64_t lda, int32_t m, const float* b, int64_t ldb, int32_t n_cols,
                   int64_t n, float* c, int64_t ldc, float* db, void* workspace, size_t workspace_bytes,
                   lidf_stream_t stream);

/* RoIAlign of the per-ray boxes at any channel count and output size (models/pipeline.py:374-391:
 * box = pixel +- roi_inp_bbox / 2, corners clamped to the image, torchvision.ops.roi_align with
 * output_size = roi_out_bbox, spatial_scale 1, sampling_ratio -1, aligned = True). out[r, (c*S + ph)*S + pw]
 * (the reference's reshape of [K, C, S, S]), row stride ld_out >= C*S*S. The shipped rgb_out = 32 /
 * roi_out_bbox = 2 run inside lidf_query_f32 / lidf_ray_features_f32 (box sums); this is the kernel for
 * the other settings (implicit_depth_amd/generic.py). Invalid ray_bid / pixel values are not checked. */
int lidf_roi_align_f32(const float* feat_grid, int32_t batch, int32_t channels, int32_t height, int32_t width,
                       const int32_t* ray_pix, const int32_t* ray_bid, int64_t n_rays, int32_t roi_inp_bbox,
                       int32_t roi_out_bbox, float* out, int64_t ld_out, lidf_stream_t stream);

/* ---- Eval depth metrics ----------------------------------------------------------------------
 * Replaces the bs == 1 evaluation branch of LIDF.compute_loss (models/pipeline.py:577-627): the
 * predicted depth map, the ground-truth depth map and the segmentation mask ([src_h, src_w],
 * device; seg_dtype 0 = no mask, 1 = uint8 / bool, 2 = float32 — the reference's corrupt_mask, cast
 * as its `astype(np.uint8)` casts it, :588) are resized to dst_h x dst_w (144 x 256 in the
 * reference) with cv2.resize's INTER_NEAREST rule, non-finite ground truth counts as 0, valid =
 * gt > 0 and mask != 0, and out (device float[10]) receives
 *   a1, a2, a3 (thresholds 1.05, 1.10, 1.25), rmse, rmse_log, log10 (natural log, as the
 *   reference computes it), abs_rel, mae, sq_rel, number of valid pixels
 * — no .cpu() round trip. dst == src gives the plain masked statistics. 0 valid pixels: NaN, as
 * torch's mean of an empty tensor.
 * workspace: lidf_depth_metrics_workspace_bytes() bytes, ZERO-FILLED ONCE by the caller; a call leaves
 * it as it found it, so one buffer serves every later call of the same stream (the statistics image is
 * summed by several workgroups, the one that arrives last adds their partial sums in a fixed order:
 * run-to-run identical). (ABI 7)                                                                 */
size_t lidf_depth_metrics_workspace_bytes(void);
int lidf_depth_metrics_f32(const float* pred_depth, const float* gt_depth, const void* seg_mask,
                           int32_t seg_dtype, int32_t src_h, int32_t src_w, int32_t dst_h, int32_t dst_w,
                           float* out, void* workspace, size_t workspace_bytes, lidf_stream_t stream);

/* ---- A whole decoder at gf_dim 32 / 64 / 128 as ONE launch (ABI 12) ------------------------------------------
 * IMNet / IEF (models/implicit_net.py:60-152: gf_dim is a constructor argument) with layer 1 factorised as the
 * fixed-width kernels have it: x [n, k] (row stride ldx) are the per-row operand columns — they multiply
 * W1[:, w1_col0 : w1_col0 + k] — and the columns that depend on a voxel / a ray alone arrive as rows of two tables
 * of 4 gf_dim floats: voxpart[vox_idx[i]] (which carries b1, and for an IEF its constant W1[:, enc columns] benc;
 * vox_idx NULL: row 0 for every row) and raypart[ray_idx[i]] (ray_idx NULL: row i); either table may be NULL.
 * inp_dim = the decoder's input width (W1 has inp_dim (+ 16 for an IEF) columns). out [n].
 * x must be readable up to column 16 ceil(k / 16) of its LAST row (the kernel reads whole 16-column groups and
 * masks what lies beyond k). prepacked != 0: the workspace still holds the weight stream an earlier call packed
 * for this decoder, k and w1_col0 (the slabs of one query; parameters unchanged in between) — no pack launch.
 * Other widths: LIDF_ERR_UNSUPPORTED (the layers run one by one, lidf_linear_f32).                             */
size_t lidf_decoder_chain_workspace_bytes(int32_t gf_dim, int32_t k);
int lidf_decoder_chain_f32(const LidfDecoder* dec, int32_t gf_dim, int32_t inp_dim, const float* x, int64_t ldx,
                           int32_t k, int32_t w1_col0, int64_t n, const float* voxpart, const int32_t* vox_idx,
                           const float* raypart, const int32_t* ray_idx, float* out, int32_t prepacked,
                           void* workspace, size_t workspace_bytes, lidf_stream_t stream);

/* ---- Decoders, training path (first step) -------------------------------------
 * What autograd does for models/implicit_net.py IMNet / IEF on [n, d] rows: a forward that keeps
 * the activations of every layer and pass, and a backward that returns the gradient of the input
 * rows and of every parameter. Same arithmetic as lidf_decoders_f32, layer by layer.
 *   act        caller-owned, lidf_decoder_train_act_floats(n, n_pass) floats, written by the
 *              forward and read by the backward (n_pass = n_iter for an IEF, 1 for an IMNet)
 *   g_out      [n] dL/d(output)
 *   d_inp      [n, d] (row stride ld_dinp) or NULL
 *   grads      device buffers shaped like the parameters (w1 [256, d(+16)], ...); overwritten.
 * Weight gradients are reduced slab by slab in a fixed order (run-to-run identical).            */
typedef struct LidfDecoderGrads {
    float *w1, *b1, *w2, *b2, *w3, *b3, *w4, *b4;
    float *wenc, *benc; /* IEF only */
} LidfDecoderGrads;
size_t lidf_decoder_train_act_floats(int64_t n, int32_t n_pass);
size_t lidf_decoder_train_workspace_bytes(int64_t n, int32_t d);
int lidf_decoder_forward_train_f32(const float* inp, int64_t n, int32_t d, int64_t ld_inp,
                                   const LidfDecoder* dec, float* out, float* act, void* workspace,
                                   size_t workspace_bytes, lidf_stream_t stream);
int lidf_decoder_backward_f32(const float* inp, int64_t n, int32_t d, int64_t ld_inp,
                              const LidfDecoder* dec, const float* act, const float* g_out,
                              float* d_inp, int64_t ld_dinp, const LidfDecoderGrads* grads,
                              void* workspace, size_t workspace_bytes, lidf_stream_t stream);
/* Both decoders on the SAME rows (models/pipeline.py:434-435: self.prob_dec(inp), self.offset_dec(inp)) as one
 * backward (ABI 12): parameter gradients as lidf_decoder_backward_f32 computes them for each, and the rows'
 * gradient d_inp = d_inp(prob) + d_inp(off) as ONE product over [S_prob | S_off] (K = 512), stored once —
 * what two autograd nodes do with two products, two [n, d] stores and an accumulation launch. The forwards are
 * two lidf_decoder_forward_train_f32 calls whose workspaces are the pair workspace at
 * lidf_decoder_pair_workspace_offset(n, d, 0 | 1) (each lidf_decoder_train_workspace_bytes(n, d) long).   */
size_t lidf_decoder_pair_workspace_bytes(int64_t n, int32_t d);
size_t lidf_decoder_pair_workspace_offset(int64_t n, int32_t d, int32_t which);
int lidf_decoder_pair_backward_f32(const float* inp, int64_t n, int32_t d, int64_t ld_inp,
                                   const LidfDecoder* prob, const LidfDecoder* off,
                                   const float* act_prob, const float* act_off,
                                   const float* g_prob, const float* g_off, float* d_inp,
                                   int64_t ld_dinp, const LidfDecoderGrads* grads_prob,
                                   const LidfDecoderGrads* grads_off, void* workspace,
                                   size_t workspace_bytes, lidf_stream_t stream);

/* ---- Query, training path -----------------------------------------------------------------------
 * LIDF.get_embedding (models/pipeline.py:338-420) with gradients, in the reference's own
 * formulation: the decoder input rows [P, 256 + 2(3+6L) + (3+6Lv)] are materialised
 *   [ vox_feat[pair_vox] | rayfeat[pair_ray][0:128] | embed(enter) | embed(leave) | embed(dir) ]
 * (rayfeat = lidf_ray_features_f32) and go through the decoders' training path; the gradient of the
 * rows is reduced to d vox_feat [V,128] (float atomics) and d rayfeat [R, 128+(3+6Lv)] (per-ray
 * sums over the contiguous pairs, pair_off = CSR), and the ROI columns of d rayfeat pass through
 * RoIAlign backward to d feat_grid [B,32,h,w]. Without a workspace every ray adds its samples'
 * shares with float atomics. With batch*128*height*width floats of workspace the rays with an
 * unclamped box are parked in a [B,128,h,w] image and gathered per pixel (no atomics); with
 * batch*height*width ints more (a rays-per-pixel table) a pixel named by one ray is stored instead
 * of added, and the boxes clamped at the border are parked too and accumulated tile by tile through
 * LDS (one add per touched pixel and tile). Outputs are overwritten.                            */
int lidf_build_rows_f32(const int32_t* pair_ray, const int32_t* pair_vox, const float* pair_t,
                        const float* ray_dir, const float* vox_center, int32_t pos_rel,
                        const float* vox_feat, const float* rayfeat, int32_t multires,
                        int32_t multires_views, int64_t n_pairs, float* rows, lidf_stream_t stream);
int lidf_rows_backward_f32(const float* d_rows, const int32_t* pair_off, const int32_t* pair_vox,
                           int64_t n_rays, int64_t n_pairs, int64_t n_vox, int32_t multires,
                           int32_t multires_views, float* d_vox_feat, float* d_rayfeat,
                           lidf_stream_t stream);
int lidf_ray_features_backward_f32(const float* d_rayfeat, const int32_t* ray_pix,
                                   const int32_t* ray_bid, int64_t n_rays, int32_t batch,
                                   int32_t height, int32_t width, int32_t roi_inp_bbox,
                                   int32_t multires_views, float* d_feat_grid, void* workspace,
                                   size_t workspace_bytes, lidf_stream_t stream);

/* ---- Query decoders, factorised training path ------------------------------------------------
 * The layer-1 rewrite of the inference kernel carried through training: per decoder
 *   layer 1 = W1[:, enter|leave] pe[p] + voxpart[pair_vox[p]] + raypart[pair_ray[p]] (+ u*off)
 * with voxpart = W1[:, 0:128] vox_feat + b1 (+c) per voxel and raypart = W1[:, rgb|dir] rayfeat per
 * ray, so only the positional encodings pe [P, 2(3+6L)] (lidf_pe_rows_f32) are per-pair rows and
 * the backward reduces the layer-1 gradient to per-voxel and per-ray sums before any product
 * with a weight: no [P, 385] row or row gradient exists. d_vox_feat [V,128] / d_rayfeat
 * [R,128+(3+6Lv)] are overwritten, or added to when accumulate_inputs != 0 (second decoder).    */
typedef struct LidfQueryTrainArgs {
    int64_t n_pairs, n_rays, n_vox;
    const int32_t *pair_off, *pair_ray, *pair_vox; /* ray-major CSR pairs, as LidfQueryArgs */
    const float* pe;        /* [P, 2(3+6*multires)] */
    int32_t multires, multires_views;
    const float* vox_feat;  /* [V,128] */
    const float* rayfeat;   /* [R,128+3+6*multires_views] (lidf_ray_features_f32) */
    const LidfDecoder* dec;
} LidfQueryTrainArgs;
int lidf_pe_rows_f32(const int32_t* pair_ray, const int32_t* pair_vox, const float* pair_t,
                     const float* ray_dir, const float* vox_center, int32_t pos_rel,
                     int32_t multires, int64_t n_pairs, float* pe, lidf_stream_t stream);
size_t lidf_query_decoder_act_floats(int64_t n_pairs, int64_t n_rays, int64_t n_vox, int32_t n_pass);
size_t lidf_query_decoder_workspace_bytes(int64_t n_pairs, int64_t n_rays, int64_t n_vox);
int lidf_query_decoder_forward_train_f32(const LidfQueryTrainArgs* args, float* out, float* act,
                                         void* workspace, size_t workspace_bytes,
                                         lidf_stream_t stream);
/* Both decoders' training forward in ONE launch of the per-point kernel of lidf_query_f32 with the
 * activations kept (the positional encodings are formed in registers, `pe` of `args` is not read;
 * args->dec = prob_dec). pair_t [P,2], ray_dir [R,3], vox_center [V,3] / pos_rel as LidfQueryArgs.
 * act_prob / act_off: lidf_query_decoder_act_floats(..., n_pass of that decoder) floats each, in the
 * layout lidf_query_decoder_backward_f32 reads. Same results as two
 * lidf_query_decoder_forward_train_f32 calls up to f32 re-association.                           */
size_t lidf_query_forward_train_workspace_bytes(int64_t n_rays, int64_t n_vox);
int lidf_query_forward_train_f32(const LidfQueryTrainArgs* args, const LidfDecoder* offset_dec,
                                 const float* pair_t, const float* ray_dir, const float* vox_center,
                                 int32_t pos_rel, float* out_prob, float* out_off, float* act_prob,
                                 float* act_off, void* workspace, size_t workspace_bytes,
                                 lidf_stream_t stream);
int lidf_query_decoder_backward_f32(const LidfQueryTrainArgs* args, const float* act,
                                    const float* g_out, float* d_vox_feat, float* d_rayfeat,
                                    int32_t accumulate_inputs, const LidfDecoderGrads* grads,
                                    void* workspace, size_t workspace_bytes, lidf_stream_t stream);
/* The same backward when the output gradient is non-zero at ONE pair per ray only — what the reference's
 * losses give offset_dec: every loss term reaches it through pred_pos = pair_pred_pos[max_pair_id]
 * (models/pipeline.py:437-454, 468-476), so dL/d pred_offset = scale * (ray_dir[r] . g_pred_pos[r]) at pair
 * rows[r] of ray r (scale = (offset_range1 - offset_range0) * sqrt(3) * part_size) and exactly zero elsewhere.
 * Rows with a zero output gradient add exactly zero to every sum of the backward, so the gradients equal those
 * of lidf_query_decoder_backward_f32 for that g_out up to f32 summation order, from n_rays rows instead of
 * n_pairs. rows [n_rays] int64: the selected pair of every ray, any value outside [0, n_pairs) for a ray
 * without one (LidfQueryArgs.max_pair_id). g_pred_pos, ray_dir [n_rays,3]; g_offset_rows [n_rays] (optional):
 * a gradient on pred_offset[rows[r]] itself, added. act_is_rows = 0: `act` as the forward wrote it for ALL pairs
 * (lidf_query_forward_train_f32), the selected rows are gathered; 1: `act` is the one-pair-per-ray list's own
 * (act_off_rows of lidf_query_forward_train_selected_f32).
 * Workspace: lidf_query_decoder_rows_workspace_bytes(n_rays, n_vox, multires, n_pass). (ABI 9)              */
size_t lidf_query_decoder_rows_workspace_bytes(int64_t n_rays, int64_t n_vox, int32_t multires,
                                               int32_t n_pass);
int lidf_query_decoder_backward_rows_f32(const LidfQueryTrainArgs* args, const float* act, int32_t act_is_rows,
                                         const int64_t* rows, const float* g_pred_pos,
                                         const float* g_offset_rows, const float* ray_dir, float scale,
                                         float* d_vox_feat, float* d_rayfeat, int32_t accumulate_inputs,
                                         const LidfDecoderGrads* grads, void* workspace,
                                         size_t workspace_bytes, lidf_stream_t stream);
/* The training forward with offset_dec on the selected pair of every ray only — the training counterpart of
 * LidfQueryArgs.offsets_selected, opt-in: nothing in the reference reads an offset of another pair
 * (pred_offset is a local of get_pred, pair_pred_pos is stored at pipeline.py:461 and never read). prob_dec on
 * every pair (activations kept in act_prob, lidf_query_decoder_act_floats(n_pairs, n_rays, n_vox, 1)), per-ray
 * softmax and arg-max (max_pair_id [R]: the arg-max also when max_pair_id_in overrides the selection,
 * pipeline.py:444-446), offset_dec on the one-pair-per-ray list (activations kept in act_off_rows,
 * lidf_query_decoder_act_floats(n_rays, n_rays, n_vox, n_iter)), pred_pos [R,3]; pred_offset [P] and
 * pair_pred_pos [P,3] receive the selected pairs' values, their other entries are left as the caller set them.
 * Workspace: lidf_query_forward_train_workspace_bytes. (ABI 9)                                                */
int lidf_query_forward_train_selected_f32(const LidfQueryTrainArgs* args, const LidfDecoder* offset_dec,
                                          const float* pair_t, const float* ray_dir, const float* vox_center,
                                          int32_t pos_rel, float offset_range0, float offset_range1,
                                          float part_size, const int64_t* max_pair_id_in, float* out_prob,
                                          float* softmax, int64_t* max_pair_id, float* pred_offset,
                                          float* pair_pred_pos, float* pred_pos, float* act_prob,
                                          float* act_off_rows, void* workspace, size_t workspace_bytes,
                                          lidf_stream_t stream);

/* ---- Positional encoding and PointNet2Stage, training path ----------------------------------------
 * What autograd derives for Embedder.embed (models/implicit_net.py:9-39) and for
 * PointNet2Stage.forward (models/pointnet.py:22-38, torch_scatter max pooling: the gradient of a
 * pooled entry goes to one source row — here the lowest row index attaining the maximum).
 *   lidf_embed_backward_f32: d x [n,3] from g_out [n, 3+6*multires]
 *   lidf_pointnet_forward_train_f32: the forward, keeping every layer's rows, the pooled tables and
 *     their arg rows in `act` (lidf_pointnet_train_act_floats floats)
 *   lidf_pointnet_backward_f32: g_out [n_vox,128] -> d_inp [n,6] (optional) and the gradient of every
 *     parameter (buffers shaped like the parameters, overwritten; float atomics inside).        */
typedef struct LidfPointNetGrads {
    float *w_p1, *b_p1, *w_p2, *b_p2, *w_v1, *b_v1, *w_p3, *b_p3, *w_p4, *b_p4, *w_v2, *b_v2;
} LidfPointNetGrads;
int lidf_embed_backward_f32(const float* x, const float* g_out, int64_t n, int multires, float* d_x,
                            lidf_stream_t stream);
size_t lidf_pointnet_train_act_floats(int64_t n_pts, int64_t n_vox);
size_t lidf_pointnet_train_workspace_bytes(int64_t n_pts, int64_t n_vox);
int lidf_pointnet_forward_train_f32(const LidfPointNet* w, const float* inp, const int32_t* vox,
                                    int64_t n_pts, int64_t n_vox, float* out, float* act,
                                    void* workspace, size_t workspace_bytes, lidf_stream_t stream);
int lidf_pointnet_backward_f32(const LidfPointNet* w, const float* inp, const int32_t* vox,
                               int64_t n_pts, int64_t n_vox, const float* act, const float* g_out,
                               float* d_inp, const LidfPointNetGrads* grads, void* workspace,
                               size_t workspace_bytes, lidf_stream_t stream);

/* ---- Stage 2, one training step in two calls (ABI 8) --------------------------------------------
 * RefineNet.forward with exp_type 'train' (models/pipeline.py:1032-1041: forward_times x get_pred_refine,
 * :922-1030) and what loss.backward() derives for it (trainers/train_refine.py:393-399): gradients of every
 * PointNet2Stage (pnet_model_refine) and decoder (offset_dec_refine, IEF or IMNet) parameter, of the
 * per-ray features (RoIAlign of full_rgb_feat | embed(dir): optional) and of the incoming position
 * (optional; stage 1 is frozen in train_refine.py:73, so it normally arrives detached).
 *
 * `args` is the argument block of lidf_refine_f32 (the train-only perturbation of :925-937 is the
 * caller's: pred_pos = stage-1 pred_pos + noise * ray_dir); precision must be LIDF_PRECISION_F32,
 * pnet_select NULL; packed / ray_l1 are not read; voxel_coord (optional) selects the cell-table end-voxel
 * lookup; workspace >= lidf_refine_train_workspace_bytes; pred_pos_out [R,3] receives the refined
 * position, end_voxel_id (optional) the last iteration's end voxels.
 *
 * What runs: one guarded-free multi-pack of every weight stream of both modules per step (forward AND
 * backward launches read the streams the forward call left in `act`); per iteration the end voxel, the
 * PointNet2Stage training forward, the decoder in its factorised form — layer 1 = W1[:, 0:128]
 * vox_feat[end voxel] (one row per voxel) + W1[:, ROI | dir] rayfeat[r] (one row per ray, formed ONCE per
 * step: the same in every iteration) + W1[:, embed(pos)] embed(pos) inside the register-chained kernel that
 * keeps the activations — and the position update. The backward walks the iterations in reverse: chained
 * input gradients, per-voxel segment sums of the layer-1 gradient (no [R,334] input gradient is formed:
 * only its embed(pos) columns and the per-voxel sums are needed), PointNet2Stage backward, the adjoint of
 * embed(pos) and of the PointNet rows of the predicted points; parameter gradients are accumulated over the
 * iterations inside the call (buffers shaped like the parameters, overwritten). No torch op in between.
 *
 * act: lidf_refine_train_act_bytes(...) bytes, written by the forward, read by the backward (the
 * parameters must not change in between). n_pass = off->n_iter for an IEF, 1 for an IMNet.
 * backward: g_pos [R,3] = dL/d pred_pos_out; d_pred_pos [R,3] / d_rayfeat [R, 128 + 3+6*multires_views]
 * optional outputs (NULL: not formed). The backward takes the forward's argument block; its workspace is pure
 * scratch (it may be another buffer of the same size) and pred_pos_out / end_voxel_id may be NULL there.  */
size_t lidf_refine_train_act_bytes(int64_t n_rays, int64_t n_valid, int64_t n_vox, int32_t multires,
                                   int32_t multires_views, int32_t n_pass, int32_t forward_times);
size_t lidf_refine_train_workspace_bytes(int64_t n_rays, int64_t n_valid, int64_t n_vox, int32_t multires);
int lidf_refine_train_forward_f32(const LidfRefineArgs* args, int32_t forward_times, void* act,
                                  size_t act_bytes, lidf_stream_t stream);
int lidf_refine_train_backward_f32(const LidfRefineArgs* args, int32_t forward_times, const void* act,
                                   size_t act_bytes, const float* g_pos, float* d_pred_pos, float* d_rayfeat,
                                   const LidfPointNetGrads* pnet_grads, const LidfDecoderGrads* dec_grads,
                                   lidf_stream_t stream);

/* ---- Per-pair / per-ray tail of get_pred with its adjoint --------------------------------------
 * models/pipeline.py:437-454 for the training path (the inference path has it inside
 * lidf_query_f32): pair_pred_pos = dir t_enter + ((off (r1-r0) + r0) sqrt(3) part_size) dir, the
 * per-ray softmax of the (detached, :442) logits, arg-max (or max_pair_id_in: the ground-truth
 * selection of :444-446) and pred_pos = pair_pred_pos[max_pair_id] with the dummy row. The adjoint
 * returns d pred_offset [P] from the gradients of pair_pred_pos [P,3] and pred_pos [R,3] (either may
 * be NULL); nothing flows into the logits.                                                        */
int lidf_query_tail_f32(const float* pred_offset, const float* pred_prob, const int32_t* pair_off,
                        const int32_t* pair_ray, const float* pair_t, const float* ray_dir,
                        int64_t n_rays, int64_t n_pairs, float offset_range0, float offset_range1,
                        float part_size, const int64_t* max_pair_id_in, float* pair_pred_pos,
                        float* softmax, int64_t* max_pair_id, float* pred_pos, lidf_stream_t stream);
int lidf_query_tail_backward_f32(const float* g_pair_pred_pos, const float* g_pred_pos,
                                 const int64_t* max_pair_id, const int32_t* pair_ray,
                                 const float* ray_dir, int64_t n_rays, int64_t n_pairs,
                                 float offset_range0, float offset_range1, float part_size,
                                 float* d_pred_offset, lidf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LIDF_HIP_H */
