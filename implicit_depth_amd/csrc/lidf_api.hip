// lidf_api.hip — the extern "C" surface declared in include/lidf_hip.h: argument checks,
// workspace carving, kernel sequencing. No allocation, no synchronisation, no global state.
#include <string.h>
#include <stdlib.h>

#include "lidf_device.h"
#include "lidf_hip.h"

extern "C" {
hipError_t lidf_launch_pack(const StreamLayout&, const NetW&, const NetW&, const L1Map&, float*,
                            float*, hipStream_t);
hipError_t lidf_launch_points(int mode, const PointsArgs&, int grid, hipStream_t);
hipError_t lidf_launch_l1only_pair(const PointsArgs&, const PointsArgs&, const PointsArgs*, int cus, hipStream_t);
hipError_t lidf_launch_refine_step(const RefineStepArgs&, long long, hipStream_t);
hipError_t lidf_launch_pack_h(const StreamLayout&, const NetW&, const NetW&, const L1Map&, float*,
                              float*, hipStream_t);
hipError_t lidf_launch_points_h(const PointsArgs&, int cus, hipStream_t);
StreamLayout lidf_make_layout_rows_h(int nets, int D, int l1only);
hipError_t lidf_launch_pack_rows_h(const StreamLayout&, const NetW&, const NetW&, const L1Map&, float*,
                                   float*, hipStream_t);
hipError_t lidf_launch_rows_h(const PointsArgs&, int grid, hipStream_t);
hipError_t lidf_launch_embed(const float*, long long, int, float*, hipStream_t);
hipError_t lidf_launch_rayfeat(const float*, float*, int, int, int, const float*, const int*,
                               const int*, long long, int, int, float*, int, hipStream_t);
hipError_t lidf_launch_ray_reduce(const float*, const float*, const int*, long long, long long,
                                  const int*, const int*, long long, float*, long long*, float*,
                                  float*, hipStream_t);
hipError_t lidf_launch_ray_dirs(const float*, int, int, int, float*, hipStream_t);
hipError_t lidf_launch_ray_aabb_dense(const float*, const float*, const int*, const int*,
                                      long long, long long, int*, float*, hipStream_t);
hipError_t lidf_launch_ray_aabb_grid_build(const float*, const int*, const int*, long long, int, int, int,
                                           int, int*, unsigned*, float*, hipStream_t);
hipError_t lidf_launch_ray_aabb_grid(bool, const float*, const int*, long long, int, int, int, int,
                                     const int*, const unsigned*, const float*, int*, const int*, int*,
                                     int*, float*, hipStream_t);
hipError_t lidf_launch_ray_aabb_compact(bool, const float*, const float*, const int*, const int*,
                                        long long, long long, int*, const int*, int*, int*, float*,
                                        hipStream_t);
hipError_t lidf_launch_pcl_aabb_dense(const float*, const float*, const int*, const int*,
                                      long long, long long, int*, hipStream_t);
hipError_t lidf_launch_pcl_aabb_last(const float*, const float*, const int*, const int*,
                                     long long, long long, int*, hipStream_t);
hipError_t lidf_launch_scan(const int*, long long, int*, int*, hipStream_t);
hipError_t lidf_launch_miss_count(const void*, int, long long, int*, int*, int*, int*, hipStream_t);
hipError_t lidf_launch_miss_fill(const void*, int, long long, const int*, const float*, int, int, int*,
                                 int*, int*, float*, long long*, long long*, long long*, hipStream_t);
hipError_t lidf_launch_linear(int nt, const LinearArgs&, int grid, hipStream_t);
hipError_t lidf_launch_chain16(int gf, const Chain16Args&, int cus, hipStream_t);
hipError_t lidf_launch_wgrad(const float*, long long, int, const float*, long long, int, long long,
                             float*, int, float*, float*, size_t, hipStream_t);
hipError_t lidf_launch_pack_pointnet(const float*, const float*, const float*, const float*, const float*,
                                     const float*, const float*, float*, const LidfPackGuardState*,
                                     hipStream_t);
hipError_t lidf_launch_rayfeat_dev(const float*, float*, int, int, int, const float*, const int*,
                                   const int*, long long, const int*, int, int, float*, int, hipStream_t);
hipError_t lidf_launch_rayfeat_phase(const float*, float*, int, int, int, const float*, const int*,
                                     const int*, long long, const int*, int, int, float*, int, int, hipStream_t);
hipError_t lidf_launch_ray_reduce_dev(const float*, const float*, const int*, long long, long long,
                                      const int*, const int*, const int*, const int*, long long, float*,
                                      long long*, float*, float*, hipStream_t, const int*, const float*, int*,
                                      int*, float*);
hipError_t lidf_launch_selected_finish(const long long*, const float*, const float*, long long, long long,
                                       const int*, const int*, const int*, const int*, long long, float*,
                                       float*, float*, float*, hipStream_t);
hipError_t lidf_launch_scan_dev(const int*, long long, const int*, int*, int*, int*, hipStream_t);
hipError_t lidf_launch_ray_aabb_compact_dev(bool, const float*, const float*, const int*, const int*,
                                            long long, long long, const int*, const int*, int*,
                                            const int*, int*, int*, float*, long long, hipStream_t);
hipError_t lidf_launch_pointnet_chain_dev(int, const float*, const float*, const int*, const float*,
                                          float*, long long, int, long long, const int*,
                                          const int*, const int*, const int*, int, hipStream_t);
hipError_t lidf_launch_vox2(const Vox2Args&, hipStream_t);
size_t lidf_group_idx_bytes(long long, long long);
hipError_t lidf_launch_group_idx(const int*, long long, const int*, long long, void*, const int**, const int**,
                                 hipStream_t);
hipError_t lidf_launch_ief16(const Ief16Args&, int, hipStream_t);
int lidf_pointnet_lds_max_voxels(void);
size_t lidf_pointnet_sort_bytes(long long, long long);
hipError_t lidf_launch_sort_idx(const int*, long long, const int*, long long, void*, const int**,
                                const int**, hipStream_t);
hipError_t lidf_launch_pointnet_chain_sorted(int, const float*, const float*, const int*, const float*,
                                             float*, long long, long long, const int*, const int*, int,
                                             hipStream_t);
hipError_t lidf_launch_refine_prep_dev(const float*, const long long*, const int*, long long, const float*,
                                       const int*, long long, const int*, const int*, const float*,
                                       long long, int, long long, float*, int*, int*,
                                       const unsigned char*, const int*, const int*, hipStream_t,
                                       const CellLookup*);
hipError_t lidf_launch_refine_rows_dev(const float*, const int*, const float*, const float*, int, int, int,
                                       int, long long, const int*, float*, int, int, hipStream_t);
hipError_t lidf_launch_refine_finish_dev(const float*, const float*, const float*, float, float, long long,
                                         const int*, float*, const int*, const int*, long long, float*,
                                         hipStream_t);
hipError_t lidf_launch_frame_head(const float*, const float*, const float*, const float*, const float*, int,
                                  int, int, int, const GridSpec&, void*, int*, int*, int*, float*,
                                  float*, int*, int*, int*, int*, int*, int*, float*, float*, float*,
                                  const int*, const int*, long long, hipStream_t);
size_t lidf_frame_head_blocks(long long);
size_t lidf_frame_head_lb_bytes(long long);
hipError_t lidf_launch_frame_cells(const int*, long long, const GridSpec&, int*, int*, float*, int*, float*,
                                   int*, int*, hipStream_t);
size_t lidf_ray_aabb_onepass_lb_bytes(long long);
hipError_t lidf_launch_ray_aabb_onepass(const float*, const float*, const int*, const int*, long long, int*,
                                        void*, int*, int*, int*, float*, long long, const int*, hipStream_t);
hipError_t lidf_launch_roi_align(const float*, int, int, int, const int*, const int*, long long, int, int,
                                 float*, long long, hipStream_t);
hipError_t lidf_launch_frame_points(const float*, const float*, const int*, const int*, const int*,
                                    const GridSpec&, long long, const int*, int*, int*, float*, float*,
                                    float*, hipStream_t);
hipError_t lidf_launch_vox_cells_bid(const int*, const int*, long long, const GridSpec&, int*, float*, int*,
                                     hipStream_t);
hipError_t lidf_launch_frame_select(const float*, const int*, const int*, long long, long long, const int*,
                                    unsigned char*, hipStream_t);
hipError_t lidf_launch_fingerprint(const float* const*, const long long*, int, unsigned long long,
                                   LidfPackGuardState*, hipStream_t);
hipError_t lidf_launch_fingerprint_multi(const float* const*, const long long*, const int*, int,
                                         const unsigned long long*, int, void*, int, hipStream_t);
size_t lidf_pointnet_chain_stream_bytes(void);
hipError_t lidf_launch_pointnet_chain(int, const float*, const float*, const int*, const float*, float*,
                                      float*, long long, long long, int, hipStream_t);
size_t lidf_pointnet_pool_scratch_bytes(long long);
hipError_t lidf_launch_dgrad_chain(const float*, const float*, const float*, const unsigned*,
                                   const unsigned*, long long, float, float*, float*, int, float*, int, hipStream_t);
size_t lidf_dgrad_stream_bytes(void);
hipError_t lidf_launch_pack_dgrad(const float*, const float*, float*, hipStream_t);
hipError_t lidf_launch_pnet_gather_segsum(const float*, const int*, long long, const int*, const int*, long long,
                                          long long, float*, float*, hipStream_t);
hipError_t lidf_launch_l4_backward(const float*, const float*, const float*, float, long long, float*,
                                   float*, float*, float*, hipStream_t);
hipError_t lidf_launch_ief_tail(const float*, const float*, const float*, int, const float*,
                                const float*, long long, int, float*, float*, float*, float*, float*,
                                float*, float*, hipStream_t);
hipError_t lidf_launch_ief_first_pass(const float*, float*, float, const float*, int, const float*,
                                      const float*, float*, float*, float*, hipStream_t);
hipError_t lidf_launch_out_act(const float*, long long, int, float*, const float*, float*,
                               hipStream_t);
hipError_t lidf_launch_build_rows(const int*, const int*, const float*, const float*, const float*, int,
                                  const float*, const float*, int, int, int, long long, float*, int,
                                  hipStream_t);
hipError_t lidf_launch_rows_backward(const float*, int, int, const int*, const int*, long long,
                                     long long, int, float*, float*, int, hipStream_t);
hipError_t lidf_launch_rayfeat_backward(const float*, int, const int*, const int*, long long, int, int,
                                        int, int, float*, float*, int*, hipStream_t);
hipError_t lidf_launch_pe_rows(const int*, const int*, const float*, const float*, const float*, int, int,
                               long long, float*, hipStream_t);
hipError_t lidf_launch_seg_sum_ray(const float*, int, const int*, long long, float*, hipStream_t);
hipError_t lidf_launch_seg_sum_idx(const float*, const int*, long long, long long, float*, void*, size_t,
                                   hipStream_t);
size_t lidf_seg_sum_idx_ws_bytes(long long, long long);
hipError_t lidf_launch_pair_pos(const float*, const int*, const float*, const float*, long long, float,
                                float, float, float, float*, hipStream_t);
hipError_t lidf_launch_ray_select(const float*, const long long*, long long, long long, float*,
                                  hipStream_t);
hipError_t lidf_launch_pair_pos_backward(const float*, const float*, const long long*, const int*,
                                         const float*, long long, float, float*, hipStream_t);
hipError_t lidf_launch_relu_mask(const float*, const float*, long long, float*, hipStream_t);
hipError_t lidf_launch_segmax_arg(const float*, const int*, const float*, long long, int, int*, hipStream_t);
hipError_t lidf_launch_segmax_backward(const float*, const int*, const int*, const float*, long long, int,
                                       int, float*, hipStream_t);
hipError_t lidf_launch_seg_sum_rows(const float*, const int*, long long, int, float*, hipStream_t);
hipError_t lidf_launch_embed_backward(const float*, const float*, long long, int, float*, hipStream_t);
hipError_t lidf_launch_depth_metrics(const float*, const float*, const void*, int, int, int, int, int,
                                     float*, void*, hipStream_t);
size_t lidf_depth_metrics_ws_bytes(void);
hipError_t lidf_launch_vox_mark(const float*, const int*, long long, const GridSpec&, int*, int*,
                                int*, hipStream_t);
hipError_t lidf_launch_vox_cells(const int*, const int*, long long, const GridSpec&, int*, float*,
                                 hipStream_t);
hipError_t lidf_launch_vox_points(const float*, const int*, const int*, const int*, long long,
                                  const GridSpec&, int*, int*, float*, hipStream_t);
hipError_t lidf_launch_refine_prep(const float*, const long long*, const int*, long long,
                                   const float*, const int*, long long, const int*, const int*,
                                   const float*, long long, const float*, int, int, int, int, int,
                                   long long, float*, int*, float*, int, int*, const unsigned char*,
                                   hipStream_t, const CellLookup*);
hipError_t lidf_launch_refine_gather(const float*, const int*, long long, float*, int, hipStream_t);
hipError_t lidf_launch_refine_gather_dev(const float*, const int*, long long, const int*, float*, int,
                                         hipStream_t);
hipError_t lidf_launch_refine_rows(const float*, const int*, const float*, const float*, int, int, int, int,
                                   long long, float*, int, hipStream_t);
hipError_t lidf_launch_refine_finish(const float*, const float*, const float*, float, float,
                                     long long, float*, hipStream_t);
}

#define LIDF_API extern "C" __attribute__((visibility("default")))
#define CHECK_HIP(x)                       \
    do {                                   \
        if ((x) != hipSuccess) return LIDF_ERR_HIP; \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static size_t linex_stream_bytes(int k);

extern "C" hipError_t lidf_launch_zero_segments(float* const* ptrs, const long long* counts, int n,
                                                hipStream_t st);
// Zero up to two buffers of 4-byte words with ONE kernel launch (the frame path: a kernel node of a
// captured graph instead of memset nodes, and one launch where hipMemsetAsync would be one per buffer).
static hipError_t zero_words(void* a, size_t words_a, void* b, size_t words_b, hipStream_t st) {
    float* ptrs[2] = {(float*)a, (float*)b};
    const long long cnt[2] = {(long long)words_a, (long long)words_b};
    return lidf_launch_zero_segments(ptrs, cnt, 2, st);
}

// Guarded packing (lidf_*_pack_guarded_f32): the guard of the API call in progress on this thread;
// every pack launch below it carries the pointer and returns at once when the fingerprint of the
// parameters did not change. Set and cleared inside one API call (no state survives the call).
static thread_local const LidfPackGuardState* tl_guard = nullptr;
struct GuardScope {
    explicit GuardScope(const LidfPackGuardState* g) { tl_guard = g; }
    ~GuardScope() { tl_guard = nullptr; }
};
static inline StreamLayout guarded(StreamLayout lay) {
    lay.guard = tl_guard;
    return lay;
}
// Pack-only API calls collect the streams of a module and pack them with one launch per
// LIDF_PACK_JOBS streams (lidf_pack_multi_kernel) instead of one launch per stream.
extern "C" hipError_t lidf_launch_pack_multi(const PackJobs&, hipStream_t);
extern "C" hipError_t lidf_launch_pack(const StreamLayout&, const NetW&, const NetW&, const L1Map&, float*,
                                       float*, hipStream_t);
static thread_local PackJobs* tl_jobs = nullptr;
static hipError_t flush_jobs(PackJobs& j, hipStream_t st) {
    const hipError_t e = lidf_launch_pack_multi(j, st);
    j.n = 0;
    return e;
}
static hipError_t pack_stream(const StreamLayout& lay, const NetW& n0, const NetW& n1, const L1Map& m,
                              float* stream, float* aux, hipStream_t st) {
    if (!tl_jobs) return lidf_launch_pack(guarded(lay), n0, n1, m, stream, aux, st);
    if (tl_jobs->n == LIDF_PACK_JOBS) {
        const hipError_t e = flush_jobs(*tl_jobs, st);
        if (e != hipSuccess) return e;
    }
    PackJob& k = tl_jobs->job[tl_jobs->n++];
    k.lay = guarded(lay); k.n0 = n0; k.n1 = n1; k.m = m; k.stream = stream; k.aux = aux;
    return hipSuccess;
}
struct JobScope {   // collects until destroyed; the owner flushes explicitly (flush_jobs) before leaving
    PackJobs jobs;
    JobScope() { jobs.n = 0; tl_jobs = &jobs; }
    ~JobScope() { tl_jobs = nullptr; }
};
static inline unsigned long long salt_mix(unsigned long long h, unsigned long long v) {
    h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    h = (h ^ (h >> 30)) * 0xBF58476D1CE4E5B9ull;
    return h ^ (h >> 27);
}

static int cu_count(int* out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return LIDF_ERR_HIP;
    if (hipDeviceGetAttribute(out, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return LIDF_ERR_HIP;
    if (*out <= 0) *out = 256;
    return LIDF_OK;
}

extern "C" hipError_t lidf_launch_zero_segments(float* const*, const long long*, int, hipStream_t);
// the gradient buffers of one decoder start at zero: one launch over the ten arrays
static hipError_t zero_decoder_grads(const LidfDecoderGrads* g, int ld1, int is_ief, hipStream_t st) {
    float* const p[10] = {g->w1, g->b1, g->w2, g->b2, g->w3, g->b3, g->w4, g->b4,
                          is_ief ? g->wenc : nullptr, is_ief ? g->benc : nullptr};
    const long long c[10] = {(long long)LIDF_H1 * ld1, LIDF_H1, (long long)LIDF_H2 * LIDF_H1, LIDF_H2,
                             (long long)LIDF_H3 * LIDF_H2, LIDF_H3, LIDF_H3, 1, 16, 16};
    return lidf_launch_zero_segments(p, c, 10, st);
}

static int check_decoder(const LidfDecoder* d) {
    if (!d) return LIDF_OK;
    if (!d->w1 || !d->b1 || !d->w2 || !d->b2 || !d->w3 || !d->b3 || !d->w4 || !d->b4)
        return LIDF_ERR_BAD_ARG;
    if (d->is_ief && (!d->wenc || !d->benc || d->n_iter < 1 || d->n_iter > 64))
        return LIDF_ERR_BAD_ARG;
    return LIDF_OK;
}

static NetW to_netw(const LidfDecoder* d, int dcore) {
    NetW n;
    n.w1 = d->w1; n.b1 = d->b1; n.w2 = d->w2; n.b2 = d->b2;
    n.w3 = d->w3; n.b3 = d->b3; n.w4 = d->w4; n.b4 = d->b4;
    n.wenc = d->wenc; n.benc = d->benc;
    n.is_ief = d->is_ief ? 1 : 0;
    n.dcore = dcore;
    n.ld1 = dcore + (d->is_ief ? 16 : 0);
    return n;
}

static L1Map rows_map(int n0, int c0, int n1, int c1, int add_bias) {
    L1Map m = {};
    m.n0 = n0; m.c0 = c0; m.n1 = n1; m.c1 = c1;
    m.D = n0 + n1;
    m.KQ1 = (m.D + 1 + 7) / 8;
    m.add_bias = add_bias;
    m.nt = 8;
    m.nout = 256;
    return m;
}

static void fill_net_args(PointsArgs& a, int i, const LidfDecoder* d, float* out, int is_offset) {
    a.npass[i] = d->is_ief ? d->n_iter : 1;
    a.init[i] = d->is_ief ? d->init_offset : 0.f;
    a.sigmoid[i] = d->use_sigmoid;
    a.out[i] = out;
    a.is_offset[i] = is_offset;
}

LIDF_API int lidf_version(void) { return LIDF_ABI_VERSION; }

LIDF_API const char* lidf_strerror(int status) {
    switch (status) {
        case LIDF_OK: return "ok";
        case LIDF_ERR_BAD_ARG: return "lidf_hip: bad argument (null pointer, negative size or inconsistent sizes)";
        case LIDF_ERR_UNSUPPORTED: return "lidf_hip: unsupported dimension";
        case LIDF_ERR_WORKSPACE: return "lidf_hip: workspace too small";
        case LIDF_ERR_HIP: return "lidf_hip: HIP runtime error";
        default: return "lidf_hip: unknown status";
    }
}

LIDF_API int lidf_embed_f32(const float* x, int64_t n, int multires, float* out,
                              lidf_stream_t stream) {
    if (n < 0 || multires < 0 || multires > 16) return LIDF_ERR_BAD_ARG;
    if (n == 0) return LIDF_OK;
    if (!x || !out) return LIDF_ERR_BAD_ARG;
    CHECK_HIP(lidf_launch_embed(x, n, multires, out, (hipStream_t)stream));
    return LIDF_OK;
}

// ---- decoders on materialised rows ---------------------------------------------------------------
LIDF_API size_t lidf_decoders_workspace_bytes(int64_t n, int d) {
    (void)n;
    if (d <= 0) return 0;
    L1Map m = rows_map(d, 0, 0, 0, 1);
    StreamLayout lay = lidf_make_layout(2, LIDF_MODE_ROWS, m);
    const size_t hs = (size_t)lidf_make_layout_rows_h(2, d, 0).total * 4;  // split-f16 stream
    const size_t fs = (size_t)lay.total * 4;
    return align_up(fs > hs ? fs : hs, 256) + align_up(2 * LIDF_AUX_FLOATS * 4, 256);
}

static int decoders_impl(const float* inp, int64_t n, int d, int64_t ld_inp, const LidfDecoder* prob,
                         const LidfDecoder* off, float* out_prob, float* out_off, void* workspace,
                         size_t workspace_bytes, int precision, lidf_stream_t stream,
                         const int* n_dev = nullptr);

LIDF_API int lidf_decoders_f32(const float* inp, int64_t n, int d, int64_t ld_inp,
                                 const LidfDecoder* prob, const LidfDecoder* off, float* out_prob,
                                 float* out_off, void* workspace, size_t workspace_bytes,
                                 lidf_stream_t stream) {
    return decoders_impl(inp, n, d, ld_inp, prob, off, out_prob, out_off, workspace, workspace_bytes,
                         LIDF_PRECISION_F32, stream);
}

LIDF_API int lidf_decoders_split_f32(const float* inp, int64_t n, int d, int64_t ld_inp,
                                       const LidfDecoder* prob, const LidfDecoder* off,
                                       float* out_prob, float* out_off, void* workspace,
                                       size_t workspace_bytes, lidf_stream_t stream) {
    return decoders_impl(inp, n, d, ld_inp, prob, off, out_prob, out_off, workspace, workspace_bytes,
                         LIDF_PRECISION_F16X3, stream);
}

static int decoders_impl(const float* inp, int64_t n, int d, int64_t ld_inp, const LidfDecoder* prob,
                         const LidfDecoder* off, float* out_prob, float* out_off, void* workspace,
                         size_t workspace_bytes, int precision, lidf_stream_t stream,
                         const int* n_dev) {
    if (n < 0 || d <= 0 || ld_inp < d) return LIDF_ERR_BAD_ARG;
    if (d > (1 << 20)) return LIDF_ERR_UNSUPPORTED;
    if (!prob && !off) return LIDF_ERR_BAD_ARG;
    int rc;
    if ((rc = check_decoder(prob)) || (rc = check_decoder(off))) return rc;
    if (n == 0) return LIDF_OK;  // empty input: nothing to write (reference returns [0,1])
    if ((prob && !out_prob) || (off && !out_off)) return LIDF_ERR_BAD_ARG;
    if (!inp) return LIDF_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < lidf_decoders_workspace_bytes(n, d))
        return LIDF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;

    const int nets = (prob ? 1 : 0) + (off ? 1 : 0);
    const LidfDecoder* ds[2] = {prob ? prob : off, off};
    float* outs[2] = {prob ? out_prob : out_off, out_off};
    L1Map m = rows_map(d, 0, 0, 0, 1);
    const bool split = precision == LIDF_PRECISION_F16X3;
    StreamLayout lay = split ? lidf_make_layout_rows_h(nets, d, 0) : lidf_make_layout(nets, LIDF_MODE_ROWS, m);
    float* stream_buf = (float*)workspace;
    float* aux = (float*)((char*)workspace + lidf_decoders_workspace_bytes(n, d) -
                          align_up(2 * LIDF_AUX_FLOATS * 4, 256));
    NetW n0 = to_netw(ds[0], d);
    NetW n1 = nets == 2 ? to_netw(ds[1], d) : n0;
    if (split)
        CHECK_HIP(lidf_launch_pack_rows_h(lay, n0, n1, m, stream_buf, aux, st));
    else
        CHECK_HIP(lidf_launch_pack(lay, n0, n1, m, stream_buf, aux, st));

    PointsArgs a = {};
    a.stream = stream_buf;
    a.aux = aux;
    a.nets = nets;
    a.l1_quads = lay.l1_quads;
    a.net_quads = lay.net_quads;
    a.n = n;
    a.n_dev = n_dev;   // optional device-side row count (the frame path; n = capacity)
    for (int i = 0; i < nets; ++i) fill_net_args(a, i, ds[i], outs[i], 0);
    a.X = inp;
    a.ldx = ld_inp;
    a.D = m.D; a.KQ1 = m.KQ1; a.has_bias = 1;
    int cus;
    if ((rc = cu_count(&cus))) return rc;
    long long ntile = (n + 127) / 128;
    int grid = (int)(ntile < cus ? ntile : cus);
    if (split)
        CHECK_HIP(lidf_launch_rows_h(a, grid, st));
    else
        CHECK_HIP(lidf_launch_points(LIDF_MODE_ROWS, a, grid, st));
    return LIDF_OK;
}

// ---- per-ray features ---------------------------------------------------------------------------
LIDF_API size_t lidf_ray_features_workspace_bytes(int batch, int height, int width, int64_t n_rays) {
    if (batch <= 0 || height <= 0 || width <= 0 || n_rays < 0) return 0;
    return align_up(((size_t)batch * 32 * height * width + (size_t)n_rays + 1) * 4, 256);
}

LIDF_API int lidf_ray_features_f32(const float* feat_grid, int batch, int height, int width,
                                     const float* ray_dir, const int32_t* ray_pix,
                                     const int32_t* ray_bid, int64_t n_rays, int roi_inp_bbox,
                                     int multires_views, float* rayfeat, void* workspace,
                                     size_t workspace_bytes, lidf_stream_t stream) {
    if (n_rays < 0 || batch <= 0 || height <= 0 || width <= 0 || roi_inp_bbox < 0 ||
        multires_views < 0 || multires_views > 16)
        return LIDF_ERR_BAD_ARG;
    if (n_rays == 0) return LIDF_OK;
    if (!feat_grid || !ray_dir || !ray_pix || !ray_bid || !rayfeat) return LIDF_ERR_BAD_ARG;
    const int ld = 128 + 3 + 6 * multires_views;
    // with room for the 4x4 box-sum image (+ the list of clamped-box rays) the unclamped boxes take
    // 4 gathers per channel; without it every ray takes the general bilinear path
    const bool use_box = workspace &&
                         workspace_bytes >= lidf_ray_features_workspace_bytes(batch, height, width, n_rays);
    CHECK_HIP(lidf_launch_rayfeat(feat_grid, use_box ? (float*)workspace : nullptr, batch, height,
                                  width, ray_dir, ray_pix, ray_bid, n_rays, roi_inp_bbox / 2,
                                  multires_views, rayfeat, ld, (hipStream_t)stream));
    return LIDF_OK;
}

LIDF_API int lidf_ray_reduce_f32(const float* pred_prob, const float* pair_pred_pos,
                                   const int32_t* pair_off, int64_t n_rays, int64_t n_pairs,
                                   const int32_t* ray_bid, const int32_t* ray_flat, int64_t hw,
                                   float* softmax, int64_t* max_pair_id, float* pred_pos,
                                   float* depth, lidf_stream_t stream) {
    if (n_rays < 0 || n_pairs < 0) return LIDF_ERR_BAD_ARG;
    if (n_rays == 0) return LIDF_OK;
    if (!pair_off || (n_pairs > 0 && !pred_prob)) return LIDF_ERR_BAD_ARG;
    if (pred_pos && n_pairs > 0 && !pair_pred_pos) return LIDF_ERR_BAD_ARG;
    if (depth && (!ray_bid || !ray_flat || !pair_pred_pos)) return LIDF_ERR_BAD_ARG;
    CHECK_HIP(lidf_launch_ray_reduce(pred_prob, pair_pred_pos, pair_off, n_rays, n_pairs, ray_bid,
                                     ray_flat, hw, softmax, (long long*)max_pair_id, pred_pos,
                                     depth, (hipStream_t)stream));
    return LIDF_OK;
}

// ---- fused query ---------------------------------------------------------------------------------
struct QueryWs {
    // the first four slots hold the packed weights (lidf_query_pack_f32 writes exactly this prefix)
    size_t stream_pts, aux_pts, stream_vox, stream_ray, packed_end, counter, voxpart, raypart, rayfeat,
        sel, box, total;
};

static QueryWs query_ws(int64_t R, int64_t V, int L, int Lv, int64_t grid_floats = 0) {
    QueryWs w;
    L1Map mf = {};
    mf.L = L;
    const int Ed = 3 + 6 * Lv;
    size_t o = 0;
    {
        const size_t f32 = (size_t)lidf_make_layout(2, LIDF_MODE_FUSED, mf).total * 4;
        const size_t f16 = (size_t)lidf_make_layout_h(2, mf).total * 4;
        w.stream_pts = o; o += align_up(f32 > f16 ? f32 : f16, 256);
    }
    w.aux_pts = o;    o += align_up(2 * LIDF_AUX_FLOATS * 4, 256);
    w.stream_vox = o; o += align_up((size_t)lidf_make_layout(2, LIDF_MODE_L1ONLY, rows_map(128, 0, 0, 0, 1)).total * 4, 256);
    {
        const size_t f32 = (size_t)lidf_make_layout(2, LIDF_MODE_L1ONLY, rows_map(128, 0, Ed, 0, 0)).total * 4;
        const size_t f16 = (size_t)lidf_make_layout_rows_h(2, 128 + Ed, 1).total * 4;
        w.stream_ray = o; o += align_up(f32 > f16 ? f32 : f16, 256);
    }
    w.packed_end = o;
    w.counter = o;    o += 256;  // tile hand-out counter of the split-f16 kernel
    w.voxpart = o;    o += align_up((size_t)(V > 0 ? V : 1) * 512 * 4, 256);
    w.raypart = o;    o += align_up((size_t)(R > 0 ? R : 1) * 512 * 4, 256);
    w.rayfeat = o;    o += align_up((size_t)(R > 0 ? R : 1) * (128 + Ed) * 4, 256);
    // offsets_selected: the one-pair-per-ray list [ray | vox | t (2) | offset | position (3)]
    w.sel = o;        o += align_up((size_t)(R > 0 ? R : 1) * 32, 256);
    // optional box-sum image + the list of clamped-box rays (last)
    w.box = o;        o += grid_floats > 0 ? align_up((size_t)(grid_floats + R + 1) * 4, 256) : 0;
    w.total = o;
    return w;
}

LIDF_API size_t lidf_query_workspace_bytes(int64_t n_rays, int64_t n_vox, int64_t grid_floats) {
    // sized for the largest supported embedding (multires = multires_views = 16)
    return query_ws(n_rays, n_vox, LIDF_MAX_L_FUSED, 16, grid_floats > 0 ? grid_floats : 0).total;
}

// The decoders' parameters re-ordered into the streams the query kernels consume: the per-point
// stream (+ layer-4 operands), the per-voxel and the per-ray layer-1 streams.
static int pack_query_weights(const LidfDecoder* prob, const LidfDecoder* off, int L, int Lv,
                              int precision, char* dst, hipStream_t st) {
    const int E = 3 + 6 * L, Ed = 3 + 6 * Lv;
    const int D = 256 + 2 * E + Ed;
    const QueryWs w = query_ws(1, 1, L, Lv);
    float* stream_pts = (float*)(dst + w.stream_pts);
    float* aux_pts = (float*)(dst + w.aux_pts);
    float* stream_vox = (float*)(dst + w.stream_vox);
    float* stream_ray = (float*)(dst + w.stream_ray);
    NetW np = to_netw(prob, D), no = to_netw(off, D);
    L1Map mf = {};
    mf.L = L;
    mf.enter_c0 = 256;
    mf.leave_c0 = 256 + E;
    const bool split = precision == LIDF_PRECISION_F16X3;
    if (split)
        CHECK_HIP(lidf_launch_pack_h(guarded(lidf_make_layout_h(2, mf)), np, no, mf, stream_pts, aux_pts, st));
    else
        CHECK_HIP(pack_stream(lidf_make_layout(2, LIDF_MODE_FUSED, mf), np, no, mf, stream_pts, aux_pts, st));
    L1Map mv = rows_map(128, 0, 0, 0, 1);  // voxel part carries b1 (+ IEF constant)
    CHECK_HIP(pack_stream(lidf_make_layout(2, LIDF_MODE_L1ONLY, mv), np, no, mv, stream_vox, aux_pts, st));
    L1Map mr = rows_map(128, 128, Ed, 256 + 2 * E, 0);  // rgb ROI columns + direction embedding
    if (split)
        CHECK_HIP(lidf_launch_pack_rows_h(guarded(lidf_make_layout_rows_h(2, mr.D, 1)), np, no, mr, stream_ray,
                                          nullptr, st));
    else
        CHECK_HIP(pack_stream(lidf_make_layout(2, LIDF_MODE_L1ONLY, mr), np, no, mr, stream_ray, aux_pts, st));
    return LIDF_OK;
}

static int check_query_model(const LidfDecoder* prob, const LidfDecoder* off, int L, int Lv,
                             int precision) {
    if (L < 0 || L > LIDF_MAX_L_FUSED || Lv < 0 || Lv > 16) return LIDF_ERR_UNSUPPORTED;
    int rc;
    if (!prob || !off) return LIDF_ERR_BAD_ARG;
    if ((rc = check_decoder(prob)) || (rc = check_decoder(off))) return rc;
    if (prob->is_ief) return LIDF_ERR_UNSUPPORTED;  // prob_dec is an IMNet (pipeline.py:82)
    if (precision != LIDF_PRECISION_F32 && precision != LIDF_PRECISION_F16X3) return LIDF_ERR_BAD_ARG;
    // the split-f16 kernel is built for up to 8 octaves (opt.model.multires = 8)
    if (precision == LIDF_PRECISION_F16X3 && L > 8) return LIDF_ERR_UNSUPPORTED;
    return LIDF_OK;
}

LIDF_API size_t lidf_query_pack_bytes(void) {
    return query_ws(1, 1, LIDF_MAX_L_FUSED, 16).packed_end;
}

LIDF_API int lidf_query_pack_f32(const LidfDecoder* prob, const LidfDecoder* off, int multires,
                                   int multires_views, int precision, void* packed,
                                   size_t packed_bytes, lidf_stream_t stream) {
    int rc = check_query_model(prob, off, multires, multires_views, precision);
    if (rc) return rc;
    if (!packed || packed_bytes < query_ws(1, 1, multires, multires_views).packed_end)
        return LIDF_ERR_WORKSPACE;
    JobScope js;
    if ((rc = pack_query_weights(prob, off, multires, multires_views, precision, (char*)packed,
                                 (hipStream_t)stream)))
        return rc;
    CHECK_HIP(flush_jobs(js.jobs, (hipStream_t)stream));
    return LIDF_OK;
}

LIDF_API size_t lidf_pack_guard_bytes(void) { return align_up(sizeof(LidfPackGuardState), 64); }

// the parameter buffers of one decoder as fingerprint segments (d_in = columns of linear_1)
static int decoder_segs(const LidfDecoder* d, int d_in, const float** ptrs, long long* n, int k) {
    const float* p[10] = {d->w1, d->b1, d->w2, d->b2, d->w3, d->b3, d->w4, d->b4, d->wenc, d->benc};
    const long long c[10] = {(long long)LIDF_H1 * d_in, LIDF_H1, (long long)LIDF_H2 * LIDF_H1, LIDF_H2,
                             (long long)LIDF_H3 * LIDF_H2, LIDF_H3, LIDF_H3, 1, 16, 16};
    for (int i = 0; i < (d->is_ief ? 10 : 8); ++i) { ptrs[k] = p[i]; n[k] = c[i]; ++k; }
    return k;
}
static unsigned long long decoder_salt(unsigned long long h, const LidfDecoder* d) {
    unsigned init_bits;
    memcpy(&init_bits, &d->init_offset, 4);
    h = salt_mix(h, (unsigned long long)d->is_ief | ((unsigned long long)d->n_iter << 8) |
                        ((unsigned long long)d->use_sigmoid << 40));
    return salt_mix(h, init_bits);
}

static unsigned long long query_salt(const LidfDecoder* prob, const LidfDecoder* off, int multires,
                                     int multires_views, int precision) {
    unsigned long long salt = salt_mix(0x51ED270B1ull, ((unsigned long long)multires << 32) |
                                                           ((unsigned long long)multires_views << 8) |
                                                           (unsigned long long)precision);
    return decoder_salt(decoder_salt(salt, prob), off);
}
static unsigned long long refine_salt(const LidfDecoder* off, int multires, int multires_views) {
    return decoder_salt(salt_mix(0x2EF19Eull, ((unsigned long long)multires << 32) |
                                                  (unsigned long long)multires_views), off);
}
#define PNET_SALT 0x9071E7ull
static int pnet_segs(const LidfPointNet* w, const float** ptrs, long long* n, int k) {
    const float* p[12] = {w->w_p1, w->b_p1, w->w_p2, w->b_p2, w->w_v1, w->b_v1,
                          w->w_p3, w->b_p3, w->w_p4, w->b_p4, w->w_v2, w->b_v2};
    const long long c[12] = {32 * 6, 32, 64 * 32, 64, 64 * 64, 64, 128 * 128, 128, 128 * 128, 128,
                             128 * 128, 128};
    for (int i = 0; i < 12; ++i) { ptrs[k] = p[i]; n[k] = c[i]; ++k; }
    return k;
}
// a guarded call that failed after its fingerprint launch must not leave a guard that matches the
// parameters beside streams that were not (completely) re-packed: forget the fingerprint
static int guard_fail(void* guard, size_t bytes, hipStream_t st, int rc) {
    (void)hipMemsetAsync(guard, 0, bytes, st);
    return rc;
}

LIDF_API int lidf_query_pack_guarded_f32(const LidfDecoder* prob, const LidfDecoder* off, int multires,
                                           int multires_views, int precision, void* packed,
                                           size_t packed_bytes, void* guard, lidf_stream_t stream) {
    int rc = check_query_model(prob, off, multires, multires_views, precision);
    if (rc) return rc;
    if (!guard) return LIDF_ERR_BAD_ARG;
    if (!packed || packed_bytes < query_ws(1, 1, multires, multires_views).packed_end)
        return LIDF_ERR_WORKSPACE;
    const int D = 256 + 2 * (3 + 6 * multires) + 3 + 6 * multires_views;
    const float* ptrs[LIDF_FP_MAX_SEGS];
    long long cnt[LIDF_FP_MAX_SEGS];
    int k = decoder_segs(prob, D + (prob->is_ief ? 16 : 0), ptrs, cnt, 0);
    k = decoder_segs(off, D + (off->is_ief ? 16 : 0), ptrs, cnt, k);
    const unsigned long long salt = query_salt(prob, off, multires, multires_views, precision);
    CHECK_HIP(lidf_launch_fingerprint(ptrs, cnt, k, salt, (LidfPackGuardState*)guard,
                                      (hipStream_t)stream));
    GuardScope scope((const LidfPackGuardState*)guard);
    JobScope js;
    if ((rc = pack_query_weights(prob, off, multires, multires_views, precision, (char*)packed,
                                 (hipStream_t)stream)))
        return guard_fail(guard, sizeof(LidfPackGuardState), (hipStream_t)stream, rc);
    if (flush_jobs(js.jobs, (hipStream_t)stream) != hipSuccess)
        return guard_fail(guard, sizeof(LidfPackGuardState), (hipStream_t)stream, LIDF_ERR_HIP);
    return LIDF_OK;
}

// dims (optional): device int32 {R, P, V} — the sync-free frame path. n_rays / n_pairs / n_vox of `q`
// are then the capacities the launches (and the workspace) are sized for, and every kernel reads its
// count on the device.
// extra_l1 (optional, f32): a third layer-1 table over the same rayfeat rows, formed in the launch of
// voxpart / raypart (the frame path: the per-ray part of the stage-2 decoder's layer 1; X is set here).
// phases: which launches this call makes — the frame path's side stream takes the per-ray ones (the caller
// launched the per-ray features into q->rayfeat_out with this workspace's box-sum image, then calls with
// QP_RAYTAB on the side stream and with QP_MAIN on the main one; both need q->packed).
enum { QP_RAYFEAT = 1, QP_RAYTAB = 2, QP_MAIN = 4, QP_ALL = 7 };
// ev_l1 (optional): recorded on the stream behind the layer-1 table launch (the frame path's side stream starts
// the stage-2 table there, beside the per-point kernel).
static int query_impl(const LidfQueryArgs* q, void* ev_points_begin, void* ev_points_end,
                      lidf_stream_t stream, const int* dims = nullptr, PointsArgs* extra_l1 = nullptr,
                      int phases = QP_ALL, void* ev_l1 = nullptr) {
    if (!q) return LIDF_ERR_BAD_ARG;
    if (dims && !q->packed) return LIDF_ERR_UNSUPPORTED;
    const int64_t R = q->n_rays, P = q->n_pairs, V = q->n_vox;
    if (R < 0 || P < 0 || V < 0) return LIDF_ERR_BAD_ARG;
    // 32-bit lane offsets in the kernels: 12 R bytes of ray directions must fit
    if (P > 0x7fffffffLL || R > 0x15555555LL) return LIDF_ERR_UNSUPPORTED;
    int rc;
    if ((rc = check_query_model(q->prob, q->off, q->multires, q->multires_views, q->precision)))
        return rc;
    hipStream_t st = (hipStream_t)stream;
    const int L = q->multires, Lv = q->multires_views;
    const int E = 3 + 6 * L, Ed = 3 + 6 * Lv;
    const int D = 256 + 2 * E + Ed;

    if (q->offsets_selected && q->precision != LIDF_PRECISION_F32) return LIDF_ERR_UNSUPPORTED;
    if (R == 0) return LIDF_OK;  // nothing to write (pipeline.py:686-687 early exit)
    if (!q->ray_dir || !q->ray_pix || !q->ray_bid || !q->pair_off) return LIDF_ERR_BAD_ARG;
    if (q->depth && !q->ray_flat) return LIDF_ERR_BAD_ARG;
    if (P > 0) {
        if (V == 0) return LIDF_ERR_BAD_ARG;
        if (!q->pair_ray || !q->pair_vox || !q->pair_t || !q->feat_grid || !q->vox_feat ||
            !q->pred_offset || !q->pred_prob || !q->pair_pred_pos)
            return LIDF_ERR_BAD_ARG;
        if (q->pos_rel && !q->vox_center) return LIDF_ERR_BAD_ARG;
        if (q->batch <= 0 || q->height <= 0 || q->width <= 0) return LIDF_ERR_BAD_ARG;
        // the box-sum image (B*32*h*w floats) is used when the caller's workspace has room for it
        const int64_t grid_floats = (int64_t)q->batch * 32 * q->height * q->width;
        QueryWs w = query_ws(R, V, L, Lv, grid_floats);
        bool use_box = q->workspace_bytes >= w.total;
        if (!use_box) w = query_ws(R, V, L, Lv);
        if (!q->workspace || q->workspace_bytes < w.total) return LIDF_ERR_WORKSPACE;
        char* ws = (char*)q->workspace;
        // 1. weight streams: the caller's packed blob (lidf_query_pack_f32, built once per
        // parameter version) or packed here, into the head of the workspace, for this call
        const char* pk = q->packed ? (const char*)q->packed : ws;
        if (!q->packed && phases != QP_ALL) return LIDF_ERR_BAD_ARG;
        if (!q->packed &&
            (rc = pack_query_weights(q->prob, q->off, L, Lv, q->precision, ws, st)))
            return rc;
        const float* stream_pts = (const float*)(pk + w.stream_pts);
        const float* aux_pts = (const float*)(pk + w.aux_pts);
        const float* stream_vox = (const float*)(pk + w.stream_vox);
        const float* stream_ray = (const float*)(pk + w.stream_ray);
        float* voxpart = (float*)(ws + w.voxpart);
        float* raypart = (float*)(ws + w.raypart);
        float* rayfeat = q->rayfeat_out ? q->rayfeat_out : (float*)(ws + w.rayfeat);
        int cus;
        if ((rc = cu_count(&cus))) return rc;
        L1Map mf = {};
        mf.L = L;
        const bool split = q->precision == LIDF_PRECISION_F16X3;
        StreamLayout lf = split ? lidf_make_layout_h(2, mf) : lidf_make_layout(2, LIDF_MODE_FUSED, mf);
        L1Map mv = rows_map(128, 0, 0, 0, 1);
        StreamLayout lv = lidf_make_layout(2, LIDF_MODE_L1ONLY, mv);
        L1Map mr = rows_map(128, 128, Ed, 256 + 2 * E, 0);
        StreamLayout lr = lidf_make_layout(2, LIDF_MODE_L1ONLY, mr);

        // 2. per-ray features [ROI 2x2 of the feature map | embed(dir)]
        if (phases & QP_RAYFEAT)
            CHECK_HIP(lidf_launch_rayfeat_dev(q->feat_grid, use_box ? (float*)(ws + w.box) : nullptr,
                                              q->batch, q->height, q->width, q->ray_dir, q->ray_pix,
                                              q->ray_bid, R, dims, q->roi_inp_bbox / 2, Lv, rayfeat,
                                              128 + Ed, st));
        // 3. layer-1 partial products: per voxel  voxpart[v] = W1[:, 0:128] vox_feat[v] + b1 (+c),
        //    per ray  raypart[r] = W1[:, rgb|dir] rayfeat[r]
        {
            PointsArgs av = {};
            av.stream = stream_vox; av.aux = aux_pts;
            av.nets = 2; av.l1_quads = lv.l1_quads; av.net_quads = lv.net_quads;
            av.n = V; av.X = q->vox_feat; av.ldx = 128;
            av.n_dev = dims ? dims + 2 : nullptr;
            av.D = mv.D; av.KQ1 = mv.KQ1; av.has_bias = 1;
            av.out_base = voxpart;
            const long long ntv = (V + 127) / 128;
            PointsArgs a = {};
            a.stream = stream_ray; a.aux = aux_pts;
            a.nets = 2; a.l1_quads = lr.l1_quads; a.net_quads = lr.net_quads;
            a.n = R; a.X = rayfeat; a.ldx = 128 + Ed;
            a.n_dev = dims;
            a.D = mr.D; a.KQ1 = mr.KQ1; a.has_bias = 0;
            a.out_base = raypart;
            long long nt = (R + 127) / 128;
            if (extra_l1) { extra_l1->X = rayfeat; extra_l1->ldx = 128 + Ed; }
            if (split) {
                if (phases & QP_MAIN)
                    CHECK_HIP(lidf_launch_points(LIDF_MODE_L1ONLY, av, (int)(ntv < 2 * cus ? ntv : 2 * cus), st));
                // the per-ray partial products with the split-f16 rows kernel (layer 1 only)
                StreamLayout lh = lidf_make_layout_rows_h(2, mr.D, 1);
                a.l1_quads = lh.l1_quads; a.net_quads = lh.net_quads;
                a.npass[0] = a.npass[1] = 0;
                if (phases & QP_RAYTAB) CHECK_HIP(lidf_launch_rows_h(a, (int)(nt < cus ? nt : cus), st));
            } else {
                // one launch over the (tile, net, half) items of the tables (an absent one: stream == NULL)
                const PointsArgs none = {};
                CHECK_HIP(lidf_launch_l1only_pair((phases & QP_RAYTAB) ? a : none, (phases & QP_MAIN) ? av : none,
                                                  (phases & QP_RAYTAB) ? extra_l1 : nullptr, cus, st));
            }
        }
        if (ev_l1) CHECK_HIP(hipEventRecord((hipEvent_t)ev_l1, st));
        if (!(phases & QP_MAIN)) return LIDF_OK;
        // 4. per-point kernel
        {
            PointsArgs a = {};
            a.stream = stream_pts; a.aux = aux_pts;
            a.nets = 2; a.l1_quads = lf.l1_quads; a.net_quads = lf.net_quads;
            a.n = P;
            a.n_dev = dims ? dims + 1 : nullptr;
            fill_net_args(a, 0, q->prob, q->pred_prob, 0);
            fill_net_args(a, 1, q->off, q->pred_offset, 1);
            a.pair_ray = q->pair_ray; a.pair_vox = q->pair_vox; a.pair_t = q->pair_t;
            a.ray_dir = q->ray_dir; a.voxpart = voxpart; a.raypart = raypart;
            a.vox_center = q->vox_center; a.pos_rel = q->pos_rel; a.L = L;
            a.r0 = q->offset_range0;
            a.rscale = q->offset_range1 - q->offset_range0;
            a.sqrt3 = (float)1.7320508075688772;  // np.sqrt(3) rounded to f32 (pipeline.py:438)
            a.part_size = q->part_size;
            a.pair_pred_pos = q->pair_pred_pos;
            a.tile_counter = (int*)(ws + w.counter);   // dynamic tile hand-out of both kernels
            // (the frame path zeroes the counter with its other scratch, in one launch up front)
            if (!dims) CHECK_HIP(hipMemsetAsync(a.tile_counter, 0, 4, st));
#ifdef LIDF_PROFILE
            a.out_base = rayfeat;  // development only: phase timers land in the rayfeat scratch
#endif
            long long nt = (P + 127) / 128;
            if (ev_points_begin) CHECK_HIP(hipEventRecord((hipEvent_t)ev_points_begin, st));
            if (split) {
                CHECK_HIP(lidf_launch_points_h(a, cus, st));
            } else if (!q->offsets_selected) {
                CHECK_HIP(lidf_launch_points(LIDF_MODE_FUSED, a, (int)(nt < cus ? nt : cus), st));
            } else {
                // Offsets for the selected pairs only (LidfQueryArgs.offsets_selected): prob_dec on every pair
                // (ONE net of the two-net stream and tables), the per-ray softmax / arg-max — which also leaves
                // the selected pair of every ray as a one-pair-per-ray list —, offset_dec on that list (R
                // points instead of P), then positions / depth / the selected pairs' slots of the per-pair arrays.
                if (!q->max_pair_id) return LIDF_ERR_BAD_ARG;
                PointsArgs ap = a;
                ap.nets = 1; ap.part_ld = 512; ap.part_off = 0;
                ap.out[1] = nullptr; ap.is_offset[0] = 0; ap.pair_pred_pos = nullptr;
                CHECK_HIP(lidf_launch_points(LIDF_MODE_FUSED, ap, (int)(nt < cus ? nt : cus), st));
                char* sb = ws + w.sel;
                const size_t Rc = (size_t)R;
                int* sel_ray = (int*)sb;
                int* sel_vox = (int*)(sb + Rc * 4);
                float* sel_t = (float*)(sb + Rc * 8);
                float* off_sel = (float*)(sb + Rc * 16);
                float* pos_sel = (float*)(sb + Rc * 20);
                CHECK_HIP(lidf_launch_ray_reduce_dev(q->pred_prob, nullptr, q->pair_off, R, P, dims,
                                                     dims ? dims + 1 : nullptr, q->ray_bid, q->ray_flat,
                                                     (long long)q->height * q->width, q->pred_prob_softmax,
                                                     (long long*)q->max_pair_id, nullptr, nullptr, st, q->pair_vox,
                                                     q->pair_t, sel_ray, sel_vox, sel_t));
                PointsArgs ao = a;
                ao.stream = stream_pts + (size_t)lf.net_quads * 256; ao.aux = aux_pts + LIDF_AUX_FLOATS;
                ao.nets = 1; ao.part_ld = 512; ao.part_off = 256;
                ao.n = R; ao.n_dev = dims;   // (dims[0] = R)
                fill_net_args(ao, 0, q->off, off_sel, 1);
                ao.out[1] = nullptr;
                ao.pair_ray = sel_ray; ao.pair_vox = sel_vox; ao.pair_t = sel_t;
                ao.pair_pred_pos = pos_sel;
                ao.tile_counter = nullptr;   // (static split: R / 32 wave-tiles)
                const long long ntr = (R + 127) / 128;
                CHECK_HIP(lidf_launch_points(LIDF_MODE_FUSED, ao, (int)(ntr < cus ? ntr : cus), st));
                CHECK_HIP(lidf_launch_selected_finish((const long long*)q->max_pair_id, off_sel, pos_sel, R, P, dims,
                                                      dims ? dims + 1 : nullptr, q->ray_bid, q->ray_flat,
                                                      (long long)q->height * q->width, q->pred_offset,
                                                      q->pair_pred_pos, q->pred_pos, q->depth, st));
                if (ev_points_end) CHECK_HIP(hipEventRecord((hipEvent_t)ev_points_end, st));
                return LIDF_OK;
            }
            if (ev_points_end) CHECK_HIP(hipEventRecord((hipEvent_t)ev_points_end, st));
        }
    }
    // 5. per-ray softmax / argmax / select / depth
    if (q->pred_prob_softmax || q->max_pair_id || q->pred_pos || q->depth) {
        CHECK_HIP(lidf_launch_ray_reduce_dev(q->pred_prob, q->pair_pred_pos, q->pair_off, R, P, dims,
                                             dims ? dims + 1 : nullptr, q->ray_bid, q->ray_flat,
                                             (long long)q->height * q->width, q->pred_prob_softmax,
                                             (long long*)q->max_pair_id, q->pred_pos, q->depth, st, nullptr,
                                             nullptr, nullptr, nullptr, nullptr));
    }
    return LIDF_OK;
}

LIDF_API int lidf_query_f32(const LidfQueryArgs* q, lidf_stream_t stream) {
    return query_impl(q, nullptr, nullptr, stream);
}

LIDF_API int lidf_query_profile_f32(const LidfQueryArgs* q, void* ev_points_begin,
                                      void* ev_points_end, lidf_stream_t stream) {
    return query_impl(q, ev_points_begin, ev_points_end, stream);
}

// ---- rays, boxes, scan ---------------------------------------------------------------------------
LIDF_API int lidf_ray_dirs_f32(const float* intr, int batch, int height, int width,
                                 float* ray_dir, lidf_stream_t stream) {
    if (batch < 0 || height < 0 || width < 0) return LIDF_ERR_BAD_ARG;
    if ((long long)batch * height * width == 0) return LIDF_OK;
    if (!intr || !ray_dir) return LIDF_ERR_BAD_ARG;
    CHECK_HIP(lidf_launch_ray_dirs(intr, batch, height, width, ray_dir, (hipStream_t)stream));
    return LIDF_OK;
}

// ---- get_miss_ray (mask -> compacted rays) ------------------------------------------------------
struct MissWs {
    size_t block_cnt, block_off, sums, total;
};
static MissWs miss_ws(int64_t n) {
    if (n < 0) n = 0;
    const size_t nb = (size_t)((n + 1023) / 1024);
    MissWs w;
    size_t o = 0;
    w.block_cnt = o; o += align_up((nb + 1) * 4, 256);
    w.block_off = o; o += align_up((nb + 2) * 4, 256);
    w.sums = o;      o += lidf_exclusive_scan_workspace_bytes((int64_t)nb);
    w.total = o;
    return w;
}

LIDF_API size_t lidf_miss_ray_workspace_bytes(int64_t n_pixels) { return miss_ws(n_pixels).total; }

LIDF_API int lidf_miss_ray_count(const void* mask, int mask_dtype, int64_t n_pixels,
                                   int32_t* n_rays, void* workspace, size_t workspace_bytes,
                                   lidf_stream_t stream) {
    if (n_pixels < 0 || !n_rays) return LIDF_ERR_BAD_ARG;
    if (mask_dtype < LIDF_MASK_F32 || mask_dtype > LIDF_MASK_I64) return LIDF_ERR_BAD_ARG;
    if (n_pixels > 0x7ffffbffLL) return LIDF_ERR_UNSUPPORTED;
    if (n_pixels > 0 && !mask) return LIDF_ERR_BAD_ARG;
    const MissWs w = miss_ws(n_pixels);
    if (!workspace || workspace_bytes < w.total) return LIDF_ERR_WORKSPACE;
    char* ws = (char*)workspace;
    CHECK_HIP(lidf_launch_miss_count(mask, mask_dtype, n_pixels, (int*)(ws + w.block_cnt),
                                     (int*)(ws + w.block_off), (int*)(ws + w.sums), n_rays,
                                     (hipStream_t)stream));
    return LIDF_OK;
}

LIDF_API int lidf_miss_ray_fill_f32(const void* mask, int mask_dtype, const float* intr, int batch,
                                      int height, int width, const void* workspace,
                                      size_t workspace_bytes, int32_t* ray_bid, int32_t* ray_flat,
                                      int32_t* ray_pix, float* ray_dir, int64_t* miss_bid,
                                      int64_t* miss_flat_img_id, int64_t* miss_img_ind,
                                      lidf_stream_t stream) {
    if (batch < 0 || height < 0 || width < 0) return LIDF_ERR_BAD_ARG;
    if (mask_dtype < LIDF_MASK_F32 || mask_dtype > LIDF_MASK_I64) return LIDF_ERR_BAD_ARG;
    const int64_t n = (int64_t)batch * height * width;
    if (n > 0x7ffffbffLL) return LIDF_ERR_UNSUPPORTED;
    if (n == 0) return LIDF_OK;
    if (!mask || (ray_dir && !intr)) return LIDF_ERR_BAD_ARG;
    const MissWs w = miss_ws(n);
    if (!workspace || workspace_bytes < w.total) return LIDF_ERR_WORKSPACE;
    const char* ws = (const char*)workspace;
    CHECK_HIP(lidf_launch_miss_fill(mask, mask_dtype, n, (const int*)(ws + w.block_off), intr,
                                    height, width, ray_bid, ray_flat, ray_pix, ray_dir,
                                    (long long*)miss_bid, (long long*)miss_flat_img_id,
                                    (long long*)miss_img_ind, (hipStream_t)stream));
    return LIDF_OK;
}

static int check_box_args(const void* a, const void* b, const void* c, const void* d, int64_t n,
                          int64_t v) {
    if (n < 0 || v < 0) return LIDF_ERR_BAD_ARG;
    if (n > 0x7fffffffLL || v > 0x7fffffffLL) return LIDF_ERR_UNSUPPORTED;
    if (n > 0 && (!a || !c)) return LIDF_ERR_BAD_ARG;
    if (v > 0 && (!b || !d)) return LIDF_ERR_BAD_ARG;
    return LIDF_OK;
}

LIDF_API int lidf_ray_aabb_dense_f32(const float* ray_dir, const float* voxel_bound,
                                       const int32_t* ray_bid, const int32_t* voxel_bid,
                                       int64_t n_rays, int64_t n_vox, int32_t* mask, float* dist,
                                       lidf_stream_t stream) {
    int rc = check_box_args(ray_dir, voxel_bound, ray_bid, voxel_bid, n_rays, n_vox);
    if (rc) return rc;
    if (n_rays == 0 || n_vox == 0) return LIDF_OK;
    if (!mask || !dist) return LIDF_ERR_BAD_ARG;
    if (n_vox > 65535) return LIDF_ERR_UNSUPPORTED;  // grid.y limit, as in the reference launch
    CHECK_HIP(lidf_launch_ray_aabb_dense(ray_dir, voxel_bound, ray_bid, voxel_bid, n_rays, n_vox,
                                         mask, dist, (hipStream_t)stream));
    return LIDF_OK;
}

LIDF_API int lidf_ray_aabb_count_f32(const float* ray_dir, const float* voxel_bound,
                                       const int32_t* ray_bid, const int32_t* voxel_bid,
                                       int64_t n_rays, int64_t n_vox, int32_t* count,
                                       lidf_stream_t stream) {
    int rc = check_box_args(ray_dir, voxel_bound, ray_bid, voxel_bid, n_rays, n_vox);
    if (rc) return rc;
    if (n_rays == 0) return LIDF_OK;
    if (!count) return LIDF_ERR_BAD_ARG;
    CHECK_HIP(lidf_launch_ray_aabb_compact(false, ray_dir, voxel_bound, ray_bid, voxel_bid, n_rays,
                                           n_vox, count, nullptr, nullptr, nullptr, nullptr,
                                           (hipStream_t)stream));
    return LIDF_OK;
}

LIDF_API int lidf_ray_aabb_fill_f32(const float* ray_dir, const float* voxel_bound,
                                      const int32_t* ray_bid, const int32_t* voxel_bid,
                                      int64_t n_rays, int64_t n_vox, const int32_t* pair_off,
                                      int32_t* pair_ray, int32_t* pair_vox, float* pair_t,
                                      lidf_stream_t stream) {
    int rc = check_box_args(ray_dir, voxel_bound, ray_bid, voxel_bid, n_rays, n_vox);
    if (rc) return rc;
    if (n_rays == 0 || n_vox == 0) return LIDF_OK;
    if (!pair_off) return LIDF_ERR_BAD_ARG;
    // pair_ray/pair_vox/pair_t may be NULL only when there is no pair; not knowable here
    CHECK_HIP(lidf_launch_ray_aabb_compact(true, ray_dir, voxel_bound, ray_bid, voxel_bid, n_rays,
                                           n_vox, nullptr, pair_off, pair_ray, pair_vox, pair_t,
                                           (hipStream_t)stream));
    return LIDF_OK;
}

// ---- the compact list on a regular voxel grid (lidf_aux.hip: per-axis interval tables + cell table)
struct GridWs {
    size_t cell, col, tab, total;
};
static bool grid_dims_ok(int batch, int rx, int ry, int rz) {
    return batch >= 0 && rx > 0 && ry > 0 && rz > 0 && rx <= 1024 && ry <= 1024 && rz <= 1024 &&
           (long long)batch * rx * ry * rz <= 0x7fffffffLL;
}
static GridWs grid_ws(int batch, int rx, int ry, int rz) {
    GridWs w;
    size_t o = 0;
    w.cell = o; o += align_up((size_t)(batch > 0 ? batch : 1) * rx * ry * rz * 4, 256);
    w.col = o;  o += align_up((size_t)(batch > 0 ? batch : 1) * rx * ry * ((rz + 31) / 32) * 4, 256);
    w.tab = o;  o += align_up((size_t)(batch > 0 ? batch : 1) * (rx + ry + rz) * 8, 256);
    w.total = o;
    return w;
}
LIDF_API size_t lidf_ray_aabb_grid_workspace_bytes(int32_t batch, int32_t rx, int32_t ry, int32_t rz) {
    if (!grid_dims_ok(batch, rx, ry, rz)) return 0;
    return grid_ws(batch, rx, ry, rz).total;
}

LIDF_API int lidf_ray_aabb_grid_build_f32(const float* voxel_bound, const int32_t* voxel_bid,
                                            const int32_t* voxel_coord, int64_t n_vox, int32_t batch,
                                            int32_t rx, int32_t ry, int32_t rz, void* grid,
                                            size_t grid_bytes, lidf_stream_t stream) {
    if (n_vox < 0 || !grid_dims_ok(batch, rx, ry, rz)) return LIDF_ERR_BAD_ARG;
    if (n_vox > 0x7fffffffLL) return LIDF_ERR_UNSUPPORTED;
    if (n_vox > 0 && (!voxel_bound || !voxel_bid || !voxel_coord)) return LIDF_ERR_BAD_ARG;
    const GridWs w = grid_ws(batch, rx, ry, rz);
    if (!grid || grid_bytes < w.total) return LIDF_ERR_WORKSPACE;
    if (batch == 0) return LIDF_OK;
    CHECK_HIP(lidf_launch_ray_aabb_grid_build(voxel_bound, voxel_bid, voxel_coord, n_vox, batch, rx, ry,
                                              rz, (int*)((char*)grid + w.cell),
                                              (unsigned*)((char*)grid + w.col),
                                              (float*)((char*)grid + w.tab), (hipStream_t)stream));
    return LIDF_OK;
}

static int grid_rays(bool fill, const float* ray_dir, const int32_t* ray_bid, int64_t n_rays,
                     int32_t batch, int32_t rx, int32_t ry, int32_t rz, const void* grid,
                     size_t grid_bytes, int32_t* count, const int32_t* pair_off, int32_t* pair_ray,
                     int32_t* pair_vox, float* pair_t, lidf_stream_t stream) {
    if (n_rays < 0 || !grid_dims_ok(batch, rx, ry, rz)) return LIDF_ERR_BAD_ARG;
    if (n_rays > 0x7fffffffLL) return LIDF_ERR_UNSUPPORTED;
    if (n_rays == 0) return LIDF_OK;
    if (!ray_dir || !ray_bid || (fill ? !pair_off : !count)) return LIDF_ERR_BAD_ARG;
    const GridWs w = grid_ws(batch, rx, ry, rz);
    if (!grid || grid_bytes < w.total) return LIDF_ERR_WORKSPACE;
    CHECK_HIP(lidf_launch_ray_aabb_grid(fill, ray_dir, ray_bid, n_rays, batch, rx, ry, rz,
                                        (const int*)((const char*)grid + w.cell),
                                        (const unsigned*)((const char*)grid + w.col),
                                        (const float*)((const char*)grid + w.tab), count, pair_off,
                                        pair_ray, pair_vox, pair_t, (hipStream_t)stream));
    return LIDF_OK;
}

LIDF_API int lidf_ray_aabb_grid_count_f32(const float* ray_dir, const int32_t* ray_bid, int64_t n_rays,
                                            int32_t batch, int32_t rx, int32_t ry, int32_t rz,
                                            const void* grid, size_t grid_bytes, int32_t* count,
                                            lidf_stream_t stream) {
    return grid_rays(false, ray_dir, ray_bid, n_rays, batch, rx, ry, rz, grid, grid_bytes, count,
                     nullptr, nullptr, nullptr, nullptr, stream);
}

LIDF_API int lidf_ray_aabb_grid_fill_f32(const float* ray_dir, const int32_t* ray_bid, int64_t n_rays,
                                           int32_t batch, int32_t rx, int32_t ry, int32_t rz,
                                           const void* grid, size_t grid_bytes, const int32_t* pair_off,
                                           int32_t* pair_ray, int32_t* pair_vox, float* pair_t,
                                           lidf_stream_t stream) {
    return grid_rays(true, ray_dir, ray_bid, n_rays, batch, rx, ry, rz, grid, grid_bytes, nullptr,
                     pair_off, pair_ray, pair_vox, pair_t, stream);
}

LIDF_API size_t lidf_exclusive_scan_workspace_bytes(int64_t n) {
    if (n < 0) n = 0;
    return align_up((size_t)((n + 1023) / 1024 + 1) * 4, 256);
}

LIDF_API int lidf_exclusive_scan_i32(const int32_t* in, int64_t n, int32_t* out, void* workspace,
                                       size_t workspace_bytes, lidf_stream_t stream) {
    if (n < 0 || !out) return LIDF_ERR_BAD_ARG;
    if (n > 0x7ffffffeLL) return LIDF_ERR_UNSUPPORTED;
    if (n > 0 && !in) return LIDF_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < lidf_exclusive_scan_workspace_bytes(n))
        return LIDF_ERR_WORKSPACE;
    CHECK_HIP(lidf_launch_scan(in, n, out, (int*)workspace, (hipStream_t)stream));
    return LIDF_OK;
}

LIDF_API int lidf_pcl_aabb_dense_f32(const float* pcl_pos, const float* voxel_bound,
                                       const int32_t* pcl_bid, const int32_t* voxel_bid,
                                       int64_t n_pts, int64_t n_vox, int32_t* mask,
                                       lidf_stream_t stream) {
    int rc = check_box_args(pcl_pos, voxel_bound, pcl_bid, voxel_bid, n_pts, n_vox);
    if (rc) return rc;
    if (n_pts == 0 || n_vox == 0) return LIDF_OK;
    if (!mask) return LIDF_ERR_BAD_ARG;
    if (n_vox > 65535) return LIDF_ERR_UNSUPPORTED;
    CHECK_HIP(lidf_launch_pcl_aabb_dense(pcl_pos, voxel_bound, pcl_bid, voxel_bid, n_pts, n_vox,
                                         mask, (hipStream_t)stream));
    return LIDF_OK;
}

LIDF_API int lidf_pcl_aabb_last_f32(const float* pcl_pos, const float* voxel_bound,
                                      const int32_t* pcl_bid, const int32_t* voxel_bid,
                                      int64_t n_pts, int64_t n_vox, int32_t* last_vox,
                                      lidf_stream_t stream) {
    int rc = check_box_args(pcl_pos, voxel_bound, pcl_bid, voxel_bid, n_pts, n_vox);
    if (rc) return rc;
    if (n_pts == 0) return LIDF_OK;
    if (!last_vox) return LIDF_ERR_BAD_ARG;
    CHECK_HIP(lidf_launch_pcl_aabb_last(pcl_pos, voxel_bound, pcl_bid, voxel_bid, n_pts, n_vox,
                                        last_vox, (hipStream_t)stream));
    return LIDF_OK;
}

// ---- PointNet2Stage --------------------------------------------------------------------------
struct LinSpec {
    const float *w, *b;    // [nout, ldw] weight, bias (nullable)
    int nout, ldw, c0, k;  // uses weight columns [c0, c0+k)
};

static size_t lin_stream_bytes(int k, int nt) {
    L1Map m = rows_map(k, 0, 0, 0, 1);
    return align_up((size_t)m.KQ1 * nt * 1024, 256);
}

// one linear layer through lidf_linear_kernel; `stream_buf` must hold lin_stream_bytes(k, nt)
static int run_linear(const LinSpec& L, const float* X, long long ldx, long long n,
                      const float* addrows, const int* addidx, int relu, float* out,
                      long long ld_out, float* pool, const int* poolidx, float* stream_buf,
                      int cus, hipStream_t st, bool pack_only = false, bool prepacked = false,
                      const int* n_dev = nullptr) {
    if (n <= 0 && !pack_only) return LIDF_OK;
    const int nt = L.nout / 32;
    L1Map m = rows_map(L.k, L.c0, 0, 0, L.b ? 1 : 0);
    m.nt = nt;
    m.nout = L.nout;
    StreamLayout lay = lidf_make_layout(1, LIDF_MODE_LINEAR, m);
    NetW nw = {};
    nw.w1 = L.w; nw.b1 = L.b; nw.ld1 = L.ldw; nw.is_ief = 0; nw.dcore = L.ldw;
    if (!prepacked) CHECK_HIP(pack_stream(lay, nw, nw, m, stream_buf, nullptr, st));
    if (pack_only) return LIDF_OK;
    LinearArgs a = {};
    a.stream = stream_buf; a.kq1 = m.KQ1; a.X = X; a.ldx = ldx; a.n = n;
    a.n_dev = n_dev;
    a.D = m.D; a.has_bias = L.b ? 1 : 0;
    a.addrows = addrows; a.addidx = addidx; a.ld_add = L.nout; a.relu = relu;
    a.out = out; a.ld_out = ld_out; a.pool = pool; a.poolidx = poolidx; a.ld_pool = L.nout;
    long long nt128 = (n + 127) / 128;
    int grid = (int)(nt128 < 4LL * cus ? nt128 : 4LL * cus);
    CHECK_HIP(lidf_launch_linear(nt, a, grid, st));
    return LIDF_OK;
}


// The per-voxel layers of PointNet2Stage in pairs (lidf_vox2_kernel): layer 1 on X [n, k1] through the
// rows-mode stream s1 (nt1 tiles), optionally layer 2 on its activated output through s2. Streams as
// run_linear / run_linex pack them (kq = k-quads per tile).
static int run_vox2(const float* X, int k1, int64_t n, const int* n_dev, const float* s1, int nt1, int relu1,
                    float* out1, const float* s2, int kq2, int nt2, int relu2, float* out2, hipStream_t st) {
    Vox2Args a = {};
    a.X = X; a.ldx = k1; a.n = n; a.n_dev = n_dev;
    a.s1 = s1; a.kq1 = (k1 + 1 + 7) / 8; a.nt1 = nt1; a.D1 = k1; a.bias1 = 1; a.relu1 = relu1;
    a.out1 = out1; a.ld1 = 32 * nt1;
    a.s2 = s2; a.kq2 = kq2; a.nt2 = nt2; a.bias2 = 1; a.relu2 = relu2; a.out2 = out2; a.ld2 = 32 * nt2;
    CHECK_HIP(lidf_launch_vox2(a, st));
    return LIDF_OK;
}

struct PnetWs {
    size_t s[7], chain, f1, f2, pool1, g1, gpart, f4, pool2, part, part_bytes, sort, total;
};
static PnetWs pnet_ws(int64_t n, int64_t v) {
    PnetWs w;
    size_t o = 0;
    const int ks[7] = {6, 32, 64, 64, 64, 128, 128};
    const int nts[7] = {1, 2, 2, 4, 4, 4, 4};
    for (int i = 0; i < 7; ++i) { w.s[i] = o; o += lin_stream_bytes(ks[i], nts[i]); }
    w.chain = o; o += align_up(lidf_pointnet_chain_stream_bytes(), 256);
    const size_t N = (size_t)(n > 0 ? n : 1), V = (size_t)(v > 0 ? v : 1);
    // (no per-point intermediates: the inference chains keep them in registers; f1 marks the end of
    // the stream slots)
    w.f1 = o; w.f2 = o; w.f4 = o;
    (void)N;
    w.pool1 = o; o += align_up(V * 64 * 4, 256);
    w.g1 = o;    o += align_up(V * 64 * 4, 256);
    w.gpart = o; o += align_up(V * 128 * 4, 256);
    w.pool2 = o; o += align_up(V * 128 * 4, 256);
    w.part_bytes = lidf_pointnet_pool_scratch_bytes(v);
    w.part = o;  o += align_up(w.part_bytes, 256);
    // tables beyond the LDS pooling: the points are walked grouped by voxel (counting-sort scratch)
    w.sort = o;  o += v > lidf_pointnet_lds_max_voxels() ? align_up(lidf_pointnet_sort_bytes(n, v), 256) : 0;
    w.total = o;
    return w;
}

LIDF_API size_t lidf_pointnet_workspace_bytes(int64_t n_pts, int64_t n_vox) {
    return pnet_ws(n_pts, n_vox).total;
}

static int check_pointnet_w(const LidfPointNet* w) {
    if (!w || !w->w_p1 || !w->b_p1 || !w->w_p2 || !w->b_p2 || !w->w_v1 || !w->b_v1 || !w->w_p3 ||
        !w->b_p3 || !w->w_p4 || !w->b_p4 || !w->w_v2 || !w->b_v2)
        return LIDF_ERR_BAD_ARG;
    return LIDF_OK;
}

// every intermediate of PointNet2Stage.forward (models/pointnet.py:22-38); f5 may be NULL when the
// pooled layer's rows are not needed (inference)
struct PnetBufs {
    float *f1, *f2, *pool1, *g1, *gpart, *f4, *f5, *pool2;
    float* streams[7];
    float* chain;   // stream of the per-point chains (inference)
    float* part;    // slabs of the LDS pooling path (NULL: global atomic maxima)
    int* sort;      // counting-sort scratch of the voxel-sorted walk (tables beyond the LDS pooling) or NULL
};

// mode 0: pack the weight streams and run; 1: pack only (lidf_pointnet_pack_f32); 2: run on
// streams packed earlier
// tail (inference, optional): one more per-voxel layer on the PointNet's output in the launch of
// vox_lin2 — the voxel columns of the stage-2 decoder's layer 1 (a rows-mode stream of nt tiles, kq
// k-quads per tile, bias column at 128; out [n_vox, 32 nt])
struct VoxTail {
    const float* stream;
    int kq, nt;
    float* out;
};
static int pointnet_impl(const LidfPointNet* w, const float* inp, const int32_t* vox, int64_t n,
                         int64_t n_vox, float* out, const PnetBufs& b, int cus, hipStream_t st,
                         int mode = 0, const VoxTail* tail = nullptr) {
    int rc;
    const bool po = mode == 1, pp = mode == 2;
    if (!po) {
        // torch_scatter fills voxels without points with 0; values are post-ReLU so 0 is the identity
        // (one launch for both tables, where hipMemsetAsync is one fill kernel each)
        CHECK_HIP(zero_words(b.pool1, (size_t)n_vox * 64, b.pool2, (size_t)n_vox * 128, st));
    }
    if (!b.f5) {
        // inference: the per-point layers are two register chains (lidf_pointnet.hip), no per-point
        // intermediate is written; the per-voxel layers stay launches over V rows
        if (!pp) {
            if (tl_jobs) {   // collected with the module's other streams (lidf_pack_multi_kernel)
                StreamLayout lay = {};
                lay.nets = 1; lay.mode = LIDF_MODE_PNET_CHAIN;
                lay.total = (int)(lidf_pointnet_chain_stream_bytes() / 4);
                NetW nw = {};
                nw.w1 = w->w_p1; nw.b1 = w->b_p1; nw.w2 = w->w_p2; nw.b2 = w->b_p2; nw.w3 = w->w_p3;
                nw.b3 = w->w_p4; nw.w4 = w->b_p4;
                L1Map m0 = {};
                CHECK_HIP(pack_stream(lay, nw, nw, m0, b.chain, nullptr, st));
            } else {
                CHECK_HIP(lidf_launch_pack_pointnet(w->w_p1, w->b_p1, w->w_p2, w->b_p2, w->w_p3, w->w_p4,
                                                    w->b_p4, b.chain, tl_guard, st));
            }
        }
        const bool sorted = !po && b.sort && n > 0 && n_vox > lidf_pointnet_lds_max_voxels() &&
                            lidf_pointnet_sort_bytes(n, n_vox) > 0;
        const int *perm = nullptr, *n_perm = nullptr;
        if (sorted) {
            CHECK_HIP(lidf_launch_sort_idx(vox, n, nullptr, n_vox, b.sort, &perm, &n_perm, st));
            CHECK_HIP(lidf_launch_pointnet_chain_sorted(1, b.chain, inp, vox, nullptr, b.pool1, n_vox, n, perm,
                                                        n_perm, cus, st));
        } else if (!po) {
            CHECK_HIP(lidf_launch_pointnet_chain(1, b.chain, inp, vox, nullptr, b.pool1, b.part, n_vox, n, cus, st));
        }
        // the weight streams of the per-voxel layers (pack_only / prepacked as before)
        if ((rc = run_linear({w->w_v1, w->b_v1, 64, 64, 0, 64}, nullptr, 64, 0, nullptr, nullptr, 1,
                             nullptr, 64, nullptr, nullptr, b.streams[2], cus, st, true, pp)))
            return rc;
        if ((rc = run_linear({w->w_p3, w->b_p3, 128, 128, 0, 64}, nullptr, 64, 0, nullptr, nullptr, 0,
                             nullptr, 128, nullptr, nullptr, b.streams[3], cus, st, true, pp)))
            return rc;
        if ((rc = run_linear({w->w_v2, w->b_v2, 128, 128, 0, 128}, nullptr, 128, 0, nullptr, nullptr, 1,
                             nullptr, 128, nullptr, nullptr, b.streams[6], cus, st, true, pp)))
            return rc;
        if (po) return LIDF_OK;
        // g1 = relu(vox_lin1(pool1)); gpart[v] = W3[:, :64] g1[v] + b3 (what the layer-3 accumulators of
        // a point start from): one launch over the voxels
        if ((rc = run_vox2(b.pool1, 64, n_vox, nullptr, b.streams[2], 2, 1, b.g1, b.streams[3], 9, 4, 0,
                           b.gpart, st)))
            return rc;
        if (sorted)
            CHECK_HIP(lidf_launch_pointnet_chain_sorted(2, b.chain, inp, vox, b.gpart, b.pool2, n_vox, n, perm,
                                                        n_perm, cus, st));
        else
            CHECK_HIP(lidf_launch_pointnet_chain(2, b.chain, inp, vox, b.gpart, b.pool2, b.part, n_vox, n, cus, st));
        // out = relu(vox_lin2(pool2)) (+ the caller's per-voxel layer on it, same launch)
        return run_vox2(b.pool2, 128, n_vox, nullptr, b.streams[6], 4, 1, out, tail ? tail->stream : nullptr,
                        tail ? tail->kq : 0, tail ? tail->nt : 0, 0, tail ? tail->out : nullptr, st);
    }
    // with the rows kept (training forward): layer by layer
    // point_feat1 = relu(point_lin1(inp)); point_feat2 = relu(point_lin2(.)); pool per voxel
    if ((rc = run_linear({w->w_p1, w->b_p1, 32, 6, 0, 6}, inp, 6, n, nullptr, nullptr, 1, b.f1, 32,
                         nullptr, nullptr, b.streams[0], cus, st, po, pp)))
        return rc;
    if ((rc = run_linear({w->w_p2, w->b_p2, 64, 32, 0, 32}, b.f1, 32, n, nullptr, nullptr, 1, b.f2, 64,
                         b.pool1, vox, b.streams[1], cus, st, po, pp)))
        return rc;
    // occ_voxel_feat = relu(vox_lin1(pool1))
    if ((rc = run_linear({w->w_v1, w->b_v1, 64, 64, 0, 64}, b.pool1, 64, n_vox, nullptr, nullptr, 1,
                         b.g1, 64, nullptr, nullptr, b.streams[2], cus, st, po, pp)))
        return rc;
    // point_lin3(cat(voxel feat, point_feat2)) = W3[:, :64] g1[vox] + W3[:, 64:] f2 + b3
    if ((rc = run_linear({w->w_p3, nullptr, 128, 128, 0, 64}, b.g1, 64, n_vox, nullptr, nullptr, 0,
                         b.gpart, 128, nullptr, nullptr, b.streams[3], cus, st, po, pp)))
        return rc;
    if ((rc = run_linear({w->w_p3, w->b_p3, 128, 128, 64, 64}, b.f2, 64, n, b.gpart, vox, 1, b.f4, 128,
                         nullptr, nullptr, b.streams[4], cus, st, po, pp)))
        return rc;
    // point_feat5 = relu(point_lin4(.)) pooled per voxel; out = relu(vox_lin2(pool2))
    if ((rc = run_linear({w->w_p4, w->b_p4, 128, 128, 0, 128}, b.f4, 128, n, nullptr, nullptr, 1,
                         b.f5, 128, b.pool2, vox, b.streams[5], cus, st, po, pp)))
        return rc;
    if ((rc = run_linear({w->w_v2, w->b_v2, 128, 128, 0, 128}, b.pool2, 128, n_vox, nullptr, nullptr,
                         1, out, 128, nullptr, nullptr, b.streams[6], cus, st, po, pp)))
        return rc;
    return LIDF_OK;
}

LIDF_API int lidf_pointnet_f32(const LidfPointNet* w, const float* inp, const int32_t* vox,
                               int64_t n, int64_t n_vox, float* out, void* workspace,
                               size_t workspace_bytes, lidf_stream_t stream) {
    if (!w || n < 0 || n_vox < 0) return LIDF_ERR_BAD_ARG;
    if (n_vox == 0) return LIDF_OK;
    if (!out || (n > 0 && (!inp || !vox))) return LIDF_ERR_BAD_ARG;
    int rc;
    if ((rc = check_pointnet_w(w))) return rc;
    PnetWs ws = pnet_ws(n, n_vox);
    if (!workspace || workspace_bytes < ws.total) return LIDF_ERR_WORKSPACE;
    char* base = (char*)workspace;
    PnetBufs b = {};
    b.f1 = (float*)(base + ws.f1); b.f2 = (float*)(base + ws.f2);
    b.pool1 = (float*)(base + ws.pool1); b.g1 = (float*)(base + ws.g1);
    b.gpart = (float*)(base + ws.gpart); b.f4 = (float*)(base + ws.f4);
    b.f5 = nullptr; b.pool2 = (float*)(base + ws.pool2);
    for (int i = 0; i < 7; ++i)
        b.streams[i] = w->packed ? (float*)((char*)w->packed + ws.s[i]) : (float*)(base + ws.s[i]);
    b.chain = w->packed ? (float*)((char*)w->packed + ws.chain) : (float*)(base + ws.chain);
    b.part = ws.part_bytes ? (float*)(base + ws.part) : nullptr;
    b.sort = n_vox > lidf_pointnet_lds_max_voxels() ? (int*)(base + ws.sort) : nullptr;
    int cus;
    if ((rc = cu_count(&cus))) return rc;
    return pointnet_impl(w, inp, vox, n, n_vox, out, b, cus, (hipStream_t)stream, w->packed ? 2 : 0);
}

LIDF_API size_t lidf_pointnet_pack_bytes(void) { return pnet_ws(1, 1).f1; }   // the stream slots

// the module's streams into `packed` as pack jobs of the caller's JobScope (or direct launches without one)
static int pointnet_pack_into(const LidfPointNet* w, void* packed, hipStream_t st) {
    const PnetWs ws = pnet_ws(1, 1);
    PnetBufs b = {};
    for (int i = 0; i < 7; ++i) b.streams[i] = (float*)((char*)packed + ws.s[i]);
    b.chain = (float*)((char*)packed + ws.chain);
    int rc, cus;
    if ((rc = cu_count(&cus))) return rc;
    return pointnet_impl(w, nullptr, nullptr, 0, 0, nullptr, b, cus, st, 1);
}

LIDF_API int lidf_pointnet_pack_f32(const LidfPointNet* w, void* packed, size_t packed_bytes,
                                      lidf_stream_t stream) {
    int rc;
    if ((rc = check_pointnet_w(w))) return rc;
    if (!packed || packed_bytes < lidf_pointnet_pack_bytes()) return LIDF_ERR_WORKSPACE;
    JobScope js;
    if ((rc = pointnet_pack_into(w, packed, (hipStream_t)stream))) return rc;
    CHECK_HIP(flush_jobs(js.jobs, (hipStream_t)stream));
    return LIDF_OK;
}

LIDF_API int lidf_pointnet_pack_guarded_f32(const LidfPointNet* w, void* packed, size_t packed_bytes,
                                              void* guard, lidf_stream_t stream) {
    int rc;
    if ((rc = check_pointnet_w(w))) return rc;
    if (!guard) return LIDF_ERR_BAD_ARG;
    if (!packed || packed_bytes < lidf_pointnet_pack_bytes()) return LIDF_ERR_WORKSPACE;
    const float* ptrs[12];
    long long cnt[12];
    pnet_segs(w, ptrs, cnt, 0);
    CHECK_HIP(lidf_launch_fingerprint(ptrs, cnt, 12, PNET_SALT, (LidfPackGuardState*)guard,
                                      (hipStream_t)stream));
    GuardScope scope((const LidfPackGuardState*)guard);
    if ((rc = lidf_pointnet_pack_f32(w, packed, packed_bytes, stream)))
        return guard_fail(guard, sizeof(LidfPackGuardState), (hipStream_t)stream, rc);
    return LIDF_OK;
}

// ---- stage-2 refinement ----------------------------------------------------------------------
struct RefineWs {
    size_t pnet_inp, pnet_vox, inp_embed, end_voxel, vox_feat, off, pnet, dec, voxpart, raypart, fact, total;
};
// the IEF of stage 2 with the voxel-feature columns of layer 1 as a per-voxel product (defined below)
static size_t refine_fact_bytes(int D);
static int refine_ief_factorised(const LidfDecoder* off, int D, const float* vox_feat, int64_t V,
                                 const float* inp_embed, const int32_t* end_voxel, int64_t R,
                                 float* out, float* voxpart, char* scratch, hipStream_t st,
                                 int pack_mode, const int* R_dev = nullptr, const int* V_dev = nullptr,
                                 void* const* ev_rows = nullptr, const float* rayfeat = nullptr, int Ed = 0,
                                 float* raypart = nullptr, bool make_raypart = false,
                                 bool voxpart_ready = false);
static RefineWs refine_ws(int64_t R, int64_t Nv, int64_t V, int D) {
    RefineWs w;
    size_t o = 0;
    const size_t n = (size_t)(R + Nv > 0 ? R + Nv : 1), r = (size_t)(R > 0 ? R : 1);
    w.pnet_inp = o;  o += align_up(n * 6 * 4, 256);
    w.pnet_vox = o;  o += align_up(n * 4, 256);
    w.inp_embed = o; o += align_up(r * D * 4, 256);
    w.end_voxel = o; o += align_up(r * 4, 256);
    w.vox_feat = o;  o += align_up((size_t)(V > 0 ? V : 1) * 128 * 4, 256);
    w.off = o;       o += align_up(r * 4, 256);
    w.pnet = o;      o += align_up(lidf_pointnet_workspace_bytes(R + Nv, V), 256);
    w.dec = o;       o += align_up(lidf_decoders_workspace_bytes(R, D), 256);
    w.voxpart = o;   o += align_up((size_t)(V > 0 ? V : 1) * LIDF_H1 * 4, 256);
    w.raypart = o;   o += align_up(r * LIDF_H1 * 4, 256);
    w.fact = o;      o += align_up(refine_fact_bytes(D), 256);
    w.total = o;
    return w;
}

LIDF_API size_t lidf_refine_workspace_bytes(int64_t n_rays, int64_t n_valid, int64_t n_vox) {
    return refine_ws(n_rays, n_valid, n_vox, 256 + 2 * (3 + 6 * 16)).total;
}

static int refine_impl(const LidfRefineArgs* q, lidf_stream_t stream, void* const* ev) {
    if (!q) return LIDF_ERR_BAD_ARG;
    const int64_t R = q->n_rays, Nv = q->n_valid, V = q->n_vox, P = q->n_pairs;
    if (R < 0 || Nv < 0 || V < 0 || P < 0) return LIDF_ERR_BAD_ARG;
    if (q->multires < 0 || q->multires > 16 || q->multires_views < 0 || q->multires_views > 16)
        return LIDF_ERR_UNSUPPORTED;
    if (R == 0) return LIDF_OK;
    if (V == 0) return LIDF_ERR_BAD_ARG;
    int rc;
    if (!q->off || !q->pnet) return LIDF_ERR_BAD_ARG;
    if ((rc = check_decoder(q->off))) return rc;
    if (!q->ray_dir || !q->ray_bid || !q->ray_flat || !q->pred_pos || !q->max_pair_id ||
        !q->voxel_bound || !q->voxel_bid || !q->rgb_img || !q->rayfeat || !q->pred_pos_out ||
        (P > 0 && !q->pair_vox) || (Nv > 0 && (!q->valid_inp || !q->valid_vox)))
        return LIDF_ERR_BAD_ARG;
    const int E = 3 + 6 * q->multires, Ed = 3 + 6 * q->multires_views;
    const int D = 256 + E + Ed;
    RefineWs w = refine_ws(R, Nv, V, D);
    if (!q->workspace || q->workspace_bytes < w.total) return LIDF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)q->workspace;
    float* pnet_inp = (float*)(ws + w.pnet_inp);
    int* pnet_vox = (int*)(ws + w.pnet_vox);
    float* inp_embed = (float*)(ws + w.inp_embed);
    int* end_voxel = q->end_voxel_id ? q->end_voxel_id : (int*)(ws + w.end_voxel);
    float* vox_feat = (float*)(ws + w.vox_feat);
    float* off = (float*)(ws + w.off);
    // final_pnet_inp = cat(valid points, predicted points), final_revidx likewise
    // (pipeline.py:1007-1008)
    CellLookup cells = {};
    if (q->voxel_coord) {   // the voxels are cells of a regular grid: end voxel through a cell table
        if (!q->cell_table || q->batch <= 0 || !(q->grid_part > 0.f)) return LIDF_ERR_BAD_ARG;
        long long nc = q->batch;
        for (int k = 0; k < 3; ++k) {
            if (q->grid_res[k] <= 0) return LIDF_ERR_BAD_ARG;
            cells.g.xmin[k] = q->grid_xmin[k];
            cells.g.r[k] = q->grid_res[k];
            nc *= q->grid_res[k];
        }
        if (nc > 0x7fffffffLL) return LIDF_ERR_UNSUPPORTED;
        cells.g.crop = q->grid_part;
        cells.g.B = q->batch;
        cells.coord = q->voxel_coord;
        cells.table = q->cell_table;
        cells.ready = q->cell_table_ready;
    }
    if (Nv > 0) {
        CHECK_HIP(hipMemcpyAsync(pnet_inp, q->valid_inp, (size_t)Nv * 6 * 4,
                                 hipMemcpyDeviceToDevice, st));
        CHECK_HIP(hipMemcpyAsync(pnet_vox, q->valid_vox, (size_t)Nv * 4, hipMemcpyDeviceToDevice,
                                 st));
    }
    CHECK_HIP(lidf_launch_refine_prep(q->pred_pos, (const long long*)q->max_pair_id, q->pair_vox, P,
                                      q->voxel_bound, q->voxel_bid, V, q->ray_bid, q->ray_flat,
                                      q->rgb_img, (long long)q->height * q->width, q->rayfeat,
                                      128 + Ed, q->multires_views, q->multires, q->pnet_pos_rel,
                                      q->pos_rel, R, pnet_inp + (size_t)Nv * 6, pnet_vox + Nv,
                                      inp_embed, D, end_voxel, q->pnet_select, st,
                                      q->voxel_coord ? &cells : nullptr));
    // f32: only embed(pos) is a per-iteration operand row — the ROI / direction columns enter layer 1
    // as a per-ray product (refine_ief_factorised); the split-f16 form keeps whole rows
    CHECK_HIP(lidf_launch_refine_rows_dev(q->pred_pos, end_voxel, q->voxel_bound, q->rayfeat, 128 + Ed,
                                          q->multires_views, q->multires, q->pos_rel, R, nullptr, inp_embed, D,
                                          q->precision == LIDF_PRECISION_F32 ? 1 : 0, st));
    if (ev && ev[0]) CHECK_HIP(hipEventRecord((hipEvent_t)ev[0], st));
    if ((rc = lidf_pointnet_f32(q->pnet, pnet_inp, pnet_vox, R + Nv, V, vox_feat, ws + w.pnet,
                                lidf_pointnet_workspace_bytes(R + Nv, V), stream)))
        return rc;
    if (ev && ev[1]) CHECK_HIP(hipEventRecord((hipEvent_t)ev[1], st));
    if (q->precision != LIDF_PRECISION_F32 && q->precision != LIDF_PRECISION_F16X3)
        return LIDF_ERR_BAD_ARG;
    if (q->precision == LIDF_PRECISION_F32) {
        // inp_embed[:, 0:128] = occ_voxel_feat[end_voxel] (pipeline.py:1016) never materialises: its
        // share of layer 1 is W1[:, 0:128] vox_feat[v] + b1 (+ c), one row per voxel, gathered as
        // the start of the layer-1 accumulators
        // weight streams: the caller's blob (lidf_refine_pack_f32, once per parameter version) or
        // packed here for this call
        if ((rc = refine_ief_factorised(q->off, D, vox_feat, V, inp_embed, end_voxel, R, off,
                                        (float*)(ws + w.voxpart),
                                        q->packed ? (char*)q->packed : ws + w.fact, st, q->packed ? 2 : 0,
                                        nullptr, nullptr, ev ? ev + 2 : nullptr, q->rayfeat, Ed,
                                        q->ray_l1 ? q->ray_l1 : (float*)(ws + w.raypart),
                                        !(q->ray_l1 && q->ray_l1_ready))))
            return rc;
    } else {
        CHECK_HIP(lidf_launch_refine_gather(vox_feat, end_voxel, R, inp_embed, D, st));
        if ((rc = decoders_impl(inp_embed, R, D, D, nullptr, q->off, nullptr, off, ws + w.dec,
                                lidf_decoders_workspace_bytes(R, D), q->precision, stream)))
            return rc;
    }
    CHECK_HIP(lidf_launch_refine_finish(q->pred_pos, off, q->ray_dir, q->offset_range0,
                                        q->offset_range1 - q->offset_range0, R, q->pred_pos_out,
                                        st));
    return LIDF_OK;
}

LIDF_API int lidf_refine_f32(const LidfRefineArgs* q, lidf_stream_t stream) {
    return refine_impl(q, stream, nullptr);
}

LIDF_API int lidf_refine_profile_f32(const LidfRefineArgs* q, void* ev_pnet_begin, void* ev_pnet_end,
                                       void* ev_ief_begin, void* ev_ief_end, lidf_stream_t stream) {
    void* ev[4] = {ev_pnet_begin, ev_pnet_end, ev_ief_begin, ev_ief_end};
    return refine_impl(q, stream, ev);
}

// ---- the evaluation path of a batch of frames without a host round trip (lidf_frame_f32) --------
// PointNet2Stage with device-side counts: pointnet_impl's inference branch, every launch sized for
// the capacities (n_cap points, V_cap voxels) and reading *n_dev / *V_dev on the device.
struct PnetFrameWs {
    size_t pool1, g1, gpart, pool2, sort, total;
};
static PnetFrameWs pnet_frame_ws(int64_t v_cap, int v_lds, int64_t n_cap = 0) {
    (void)v_lds;
    PnetFrameWs w;
    size_t o = 0;
    const size_t V = (size_t)(v_cap > 0 ? v_cap : 1);
    w.pool1 = o; o += align_up(V * 64 * 4, 256);
    w.g1 = o;    o += align_up(V * 64 * 4, 256);
    w.gpart = o; o += align_up(V * 128 * 4, 256);
    w.pool2 = o; o += align_up(V * 128 * 4, 256);
    // (the grouping of a batch's points by voxel: its table of counts, V words at the head, starts zeroed)
    w.sort = o;  o += n_cap > 0 ? align_up(lidf_group_idx_bytes(n_cap, v_cap), 256) : 0;
    w.total = o;
    return w;
}
// (pool1 / pool2 of `ws` must be zero on entry: merged with the caller's other zeroed scratch)
// sort_cap > 0 (several frames per batch: the voxel table is expected to exceed the LDS pooling): beyond
// v_lds voxels the chains walk the points grouped by voxel (the counting sort runs unconditionally).
// Four launches: chain 1 -> [vox_lin1 | voxel half of point_lin3] -> chain 2 -> [vox_lin2 | tail].
static int pointnet_frame(const LidfPointNet* w, const float* inp, const int32_t* vox, int64_t n_cap,
                          const int* n_dev, int64_t V_cap, int v_lds, const int* V_dev, float* out,
                          char* ws, int cus, hipStream_t st, int64_t sort_cap = 0,
                          const VoxTail* tail = nullptr) {
    int rc;
    if ((rc = check_pointnet_w(w))) return rc;
    if (!w->packed) return LIDF_ERR_BAD_ARG;
    const PnetFrameWs f = pnet_frame_ws(V_cap, v_lds, sort_cap);
    const int *perm = nullptr, *n_perm = nullptr;
    if (sort_cap > 0)
        CHECK_HIP(lidf_launch_group_idx(vox, n_cap, n_dev, V_cap, ws + f.sort, &perm, &n_perm, st));
    const PnetWs pw = pnet_ws(1, 1);   // offsets of the packed streams
    float* streams[7];
    for (int i = 0; i < 7; ++i) streams[i] = (float*)((char*)w->packed + pw.s[i]);
    const float* chain = (const float*)((const char*)w->packed + pw.chain);
    float* pool1 = (float*)(ws + f.pool1);
    float* g1 = (float*)(ws + f.g1);
    float* gpart = (float*)(ws + f.gpart);
    float* pool2 = (float*)(ws + f.pool2);
    CHECK_HIP(lidf_launch_pointnet_chain_dev(1, chain, inp, vox, nullptr, pool1, V_cap, v_lds, n_cap,
                                             n_dev, V_dev, perm, n_perm, cus, st));
    if ((rc = run_vox2(pool1, 64, V_cap, V_dev, streams[2], 2, 1, g1, streams[3], 9, 4, 0, gpart, st)))
        return rc;
    CHECK_HIP(lidf_launch_pointnet_chain_dev(2, chain, inp, vox, gpart, pool2, V_cap, v_lds, n_cap,
                                             n_dev, V_dev, perm, n_perm, cus, st));
    return run_vox2(pool2, 128, V_cap, V_dev, streams[6], 4, 1, out, tail ? tail->stream : nullptr,
                    tail ? tail->kq : 0, tail ? tail->nt : 0, 0, tail ? tail->out : nullptr, st);
}

struct FrameWs {
    size_t lb_head, lb_pairs, cell_flag, cell_rank, vox_bid, vox_center, pt_key, pt_rank, pnet,
        query, inp_embed, off, vox_feat_r, voxpart_r, raypart_r, pos_a, pos_b, pnet_abs, sel, dec, total;
    size_t lb_bytes;   // lb_head .. cell_flag: the look-back tickets / status words, zeroed with cell_flag
};
static FrameWs frame_ws(int B, int h, int w, const int32_t* res, int64_t max_pairs, int v_lds,
                        int refine_times) {
    (void)max_pairs;
    FrameWs f;
    size_t o = 0;
    const size_t N = (size_t)B * h * w, C = (size_t)B * res[0] * res[1] * res[2];
    f.lb_head = o;   o += align_up(lidf_frame_head_lb_bytes((long long)N), 256);
    f.lb_pairs = o;  o += align_up(lidf_ray_aabb_onepass_lb_bytes((long long)N), 256);
    f.lb_bytes = o;
    f.cell_flag = o; o += align_up(C * 4, 256);
    f.cell_rank = o; o += align_up((C + 1) * 4, 256);
    f.vox_bid = o;   o += align_up((C + (size_t)B + 1) * 4, 256);   // [V] image of a voxel | [B + 1] first voxel of an image
    f.vox_center = o; o += align_up(C * 12, 256);
    f.pt_key = o;    o += align_up(N * 4, 256);
    f.pt_rank = o;   o += align_up((N + 1) * 4, 256);
    f.pnet = o;      o += align_up(pnet_frame_ws((int64_t)C, v_lds, B >= 2 ? (int64_t)(2 * N) : 0).total, 256);
    f.query = o;     o += align_up(lidf_query_workspace_bytes((int64_t)N, (int64_t)C, (int64_t)B * 32 * h * w), 256);
    const int Dmax = 256 + 2 * (3 + 6 * 16);
    const bool rf = refine_times > 0;
    f.inp_embed = o;  o += rf ? align_up(N * Dmax * 4, 256) : 0;
    f.off = o;        o += rf ? align_up(N * 4, 256) : 0;
    f.vox_feat_r = o; o += rf ? align_up(C * 128 * 4, 256) : 0;
    f.voxpart_r = o;  o += rf ? align_up(C * LIDF_H1 * 4, 256) : 0;
    f.raypart_r = o;  o += rf ? align_up(N * LIDF_H1 * 4, 256) : 0;
    f.pos_a = o;      o += rf ? align_up(N * 12, 256) : 0;
    f.pos_b = o;      o += rf ? align_up(N * 12, 256) : 0;
    f.pnet_abs = o;   o += rf ? align_up(2 * N * 24, 256) : 0;
    f.sel = o;        o += rf ? align_up(N, 256) : 0;
    f.dec = o;        o += rf ? align_up(lidf_decoders_workspace_bytes((int64_t)N, Dmax), 256) : 0;   // split-f16 IEF
    f.total = o;
    return f;
}

static int frame_lds_voxels(int32_t v, size_t C) {
    int l = v > 0 ? v : 128;   // (pipeline.FrameRunner passes 288 for single frames, 128 for batches)
    if (l > 288) l = 288;
    if ((size_t)l > C) l = (int)C;
    return l < 1 ? 1 : l;
}

LIDF_API size_t lidf_frame_workspace_bytes(int32_t batch, int32_t height, int32_t width, const int32_t* res,
                                           int64_t max_pairs, int32_t lds_voxels, int32_t refine_times) {
    if (batch <= 0 || height <= 0 || width <= 0 || !res || res[0] <= 0 || res[1] <= 0 || res[2] <= 0) return 0;
    const size_t C = (size_t)batch * res[0] * res[1] * res[2];
    return frame_ws(batch, height, width, res, max_pairs, frame_lds_voxels(lds_voxels, C), refine_times).total;
}


// ---- the weight streams of every module of a frame in one blob, validated by ONE fingerprint launch ----
struct FramePackLay {
    size_t query, pnet, pnet_r, refine, total;
};
static FramePackLay frame_pack_lay() {
    FramePackLay l;
    size_t o = 0;
    l.query = o;  o += align_up(lidf_query_pack_bytes(), 256);
    l.pnet = o;   o += align_up(lidf_pointnet_pack_bytes(), 256);
    l.pnet_r = o; o += align_up(lidf_pointnet_pack_bytes(), 256);
    l.refine = o; o += align_up(lidf_refine_pack_bytes(16, 16), 256);
    l.total = o;
    return l;
}
#define FRAME_GUARD_STRIDE 64
static_assert(sizeof(LidfPackGuardState) <= FRAME_GUARD_STRIDE, "guard stride");
LIDF_API size_t lidf_frame_pack_bytes(void) { return frame_pack_lay().total; }
LIDF_API size_t lidf_frame_pack_guard_bytes(void) { return (size_t)LIDF_FP_GROUPS * FRAME_GUARD_STRIDE; }

// groups: 0 the query's two decoders, 1 the PointNet, 2 the stage-2 PointNet, 3 the stage-2 decoder (f32)
static int frame_pack_guarded(const LidfFrameArgs* a, char* blob, char* guards, hipStream_t st) {
    const bool rf = a->refine_times > 0, split = a->precision == LIDF_PRECISION_F16X3;
    const int L = a->multires, Lv = a->multires_views;
    const int D = 256 + 2 * (3 + 6 * L) + 3 + 6 * Lv, Dr = 256 + 3 + 6 * L + 3 + 6 * Lv;
    const FramePackLay l = frame_pack_lay();
    const float* ptrs[LIDF_FP_MULTI_SEGS];
    long long cnt[LIDF_FP_MULTI_SEGS];
    int grp[LIDF_FP_MULTI_SEGS];
    unsigned long long salts[LIDF_FP_GROUPS] = {};
    int k = decoder_segs(a->prob, D + (a->prob->is_ief ? 16 : 0), ptrs, cnt, 0);
    k = decoder_segs(a->off, D + (a->off->is_ief ? 16 : 0), ptrs, cnt, k);
    for (int i = 0; i < k; ++i) grp[i] = 0;
    salts[0] = query_salt(a->prob, a->off, L, Lv, a->precision);
    int k0 = k, ngrp = 2;
    k = pnet_segs(a->pnet, ptrs, cnt, k);
    for (int i = k0; i < k; ++i) grp[i] = 1;
    salts[1] = PNET_SALT;
    if (rf) {
        k0 = k;
        k = pnet_segs(a->pnet_refine, ptrs, cnt, k);
        for (int i = k0; i < k; ++i) grp[i] = 2;
        salts[2] = PNET_SALT;
        ngrp = 3;
        if (!split) {   // (the split-f16 IEF of stage 2 packs its rows stream inside the call)
            k0 = k;
            k = decoder_segs(a->off_refine, Dr + (a->off_refine->is_ief ? 16 : 0), ptrs, cnt, k);
            for (int i = k0; i < k; ++i) grp[i] = 3;
            salts[3] = refine_salt(a->off_refine, L, Lv);
            ngrp = 4;
        }
    }
    const size_t gbytes = (size_t)LIDF_FP_GROUPS * FRAME_GUARD_STRIDE;
    CHECK_HIP(lidf_launch_fingerprint_multi(ptrs, cnt, grp, k, salts, ngrp, guards, FRAME_GUARD_STRIDE, st));
    int rc = LIDF_OK;
    JobScope js;
    {
        GuardScope g((const LidfPackGuardState*)guards);
        rc = pack_query_weights(a->prob, a->off, L, Lv, a->precision, blob + l.query, st);
    }
    if (!rc) {
        GuardScope g((const LidfPackGuardState*)(guards + FRAME_GUARD_STRIDE));
        rc = pointnet_pack_into(a->pnet, blob + l.pnet, st);
    }
    if (!rc && rf) {
        GuardScope g((const LidfPackGuardState*)(guards + 2 * FRAME_GUARD_STRIDE));
        rc = pointnet_pack_into(a->pnet_refine, blob + l.pnet_r, st);
    }
    if (!rc && rf && !split) {
        GuardScope g((const LidfPackGuardState*)(guards + 3 * FRAME_GUARD_STRIDE));
        rc = refine_ief_factorised(a->off_refine, Dr, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr,
                                   blob + l.refine, st, 1, nullptr, nullptr, nullptr, nullptr, 3 + 6 * Lv,
                                   nullptr, false);
    }
    if (!rc && flush_jobs(js.jobs, st) != hipSuccess) rc = LIDF_ERR_HIP;
    return rc ? guard_fail(guards, gbytes, st, rc) : LIDF_OK;
}

// Side-stream bookkeeping of one lidf_frame_f32 call: `open` once the first fork was recorded. Every non-zero
// return then joins (lidf_frame_f32 below): ev_join is recorded behind whatever the side stream has queued and
// the caller's stream waits for it — no fork is left open (a capture stays joinable) and nothing of the
// frame still runs on the side stream when the caller's next launch (or free) reaches its stream.
struct ForkState {
    bool open = false;
};
static int frame_impl(const LidfFrameArgs* a_in, lidf_stream_t stream, ForkState* fork);

LIDF_API int lidf_frame_f32(const LidfFrameArgs* a_in, lidf_stream_t stream) {
    ForkState fork;
    const int rc = frame_impl(a_in, stream, &fork);
    if (rc != LIDF_OK && fork.open) {
        if (hipEventRecord((hipEvent_t)a_in->ev_join, (hipStream_t)a_in->aux_stream) == hipSuccess)
            (void)hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)a_in->ev_join, 0);
    }
    return rc;
}

LIDF_API int lidf_event_create(void** out) {
    if (!out) return LIDF_ERR_BAD_ARG;
    hipEvent_t e = nullptr;
    CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *out = (void*)e;
    return LIDF_OK;
}
LIDF_API int lidf_event_destroy(void* event) {
    if (!event) return LIDF_OK;
    CHECK_HIP(hipEventDestroy((hipEvent_t)event));
    return LIDF_OK;
}

// test hook (LidfFrameArgs.fail_after): leave as a failed launch of stage k would
// — honoured only in a process that set LIDF_TEST_FAULTS=1 (read per call, no state kept): a C caller that left
// the trailing field of a grown struct uninitialised does not get spurious mid-frame failures
static inline int frame_fail_stage(const LidfFrameArgs* a) {
    if (!a->fail_after) return 0;
    const char* e = getenv("LIDF_TEST_FAULTS");
    return (e && e[0] == '1') ? a->fail_after : 0;
}
#define FRAME_FAIL_AFTER(k) do { if (fail_stage == (k)) return LIDF_ERR_HIP; } while (0)

static int frame_impl(const LidfFrameArgs* a_in, lidf_stream_t stream, ForkState* fork) {
    if (!a_in) return LIDF_ERR_BAD_ARG;
    LidfFrameArgs a_loc = *a_in;
    LidfFrameArgs* a = &a_loc;
    const int fail_stage = frame_fail_stage(a);
    const int B = a->batch, h = a->height, w = a->width;
    if (B <= 0 || h <= 0 || w <= 0 || a->valid_stride < 1 || a->max_pairs <= 0 || !(a->part_size > 0.f))
        return LIDF_ERR_BAD_ARG;
    if (a->res[0] <= 0 || a->res[1] <= 0 || a->res[2] <= 0 || a->refine_times < 0) return LIDF_ERR_BAD_ARG;
    const int64_t N = (int64_t)B * h * w, C = (int64_t)B * a->res[0] * a->res[1] * a->res[2];
    if (N > 0x15555555LL || C > 0x7fffffffLL || a->max_pairs > 0x7fffffffLL) return LIDF_ERR_UNSUPPORTED;
    const bool own_pack = a->pack_mode != LIDF_FRAME_PACK_CALLER;
    if (a->pack_mode != LIDF_FRAME_PACK_CALLER && a->pack_mode != LIDF_FRAME_PACK_GUARDED &&
        a->pack_mode != LIDF_FRAME_PACK_TRUSTED)
        return LIDF_ERR_BAD_ARG;
    if (!a->rgb || !a->xyz_corrupt || !a->valid_mask || !a->intr || !a->feat_grid || !a->pnet || !a->prob ||
        !a->off || (!own_pack && !a->packed_query) || !a->counts)
        return LIDF_ERR_BAD_ARG;
    if (!a->valid_bid || !a->valid_flat || !a->valid_xyz || !a->valid_rgb || !a->occ_bid_coord ||
        !a->voxel_bound || !a->valid_v_pid || !a->revidx || !a->valid_v_rel_coord || !a->pnet_inp ||
        !a->occ_voxel_feat || !a->ray_bid || !a->ray_flat || !a->ray_pix || !a->ray_dir || !a->pair_off ||
        !a->pair_ray || !a->pair_vox || !a->pair_t || !a->pred_offset || !a->pred_prob ||
        !a->pair_pred_pos || !a->max_pair_id || !a->pred_pos || !a->rayfeat || !a->pred_depth)
        return LIDF_ERR_BAD_ARG;
    if (a->n_valid_idx > 0 && (a->n_valid_idx > N || !a->valid_idx_bid || !a->valid_idx_flat))
        return LIDF_ERR_BAD_ARG;
    const bool rf = a->refine_times > 0;
    const bool split = a->precision == LIDF_PRECISION_F16X3;
    if (rf && (!a->pnet_refine || !a->off_refine || (!split && !own_pack && !a->packed_refine) ||
               !a->pred_pos_refine || !a->end_voxel_id || !a->pred_depth_refine))
        return LIDF_ERR_BAD_ARG;
    int rc, cus;
    if ((rc = check_query_model(a->prob, a->off, a->multires, a->multires_views, a->precision))) return rc;
    if (rf && (rc = check_decoder(a->off_refine))) return rc;
    if ((rc = check_pointnet_w(a->pnet)) || (rf && (rc = check_pointnet_w(a->pnet_refine)))) return rc;
    if ((rc = cu_count(&cus))) return rc;
    // the weight streams: the caller's blobs, or the frame's own blob (validated here by one fingerprint
    // launch over every module, or trusted)
    LidfPointNet pn_loc = *a->pnet, pnr_loc = {};
    if (rf) pnr_loc = *a->pnet_refine;
    if (own_pack && (!a->pack_blob || !a->pack_guard || a->pack_blob_bytes < frame_pack_lay().total))
        return LIDF_ERR_WORKSPACE;
    const int v_lds = frame_lds_voxels(a->lds_voxels, (size_t)C);
    const FrameWs f = frame_ws(B, h, w, a->res, a->max_pairs, v_lds, a->refine_times);
    if (!a->workspace || a->workspace_bytes < f.total) return LIDF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)a->workspace;
    const int64_t grid_floats = (int64_t)B * 32 * h * w;
    const QueryWs qw = query_ws(N, C, a->multires, a->multires_views, grid_floats);
    // side stream (optional): launches that fill a fraction of the device each run side by side —
    //   side: weight-stream guard (fingerprint + early-exit packs), box sums | per-ray features
    //   main: zeroed scratch, head                                           | cells, pairs, points, (guard done) PointNet
    // ev_fork is recorded on the main stream twice (start, rays exist), ev_join on the side stream twice (guard
    // done, per-ray features done); each wait refers to the record issued before it.
    const bool two = a->aux_stream && a->ev_fork && a->ev_join;
    hipStream_t sx = (hipStream_t)a->aux_stream;
    const int Edv = 3 + 6 * a->multires_views;
    bool guard_on_side = false;
    if (two) {
        CHECK_HIP(hipEventRecord((hipEvent_t)a->ev_fork, st));
        fork->open = true;   // from here on every error exit joins (lidf_frame_f32)
        CHECK_HIP(hipStreamWaitEvent(sx, (hipEvent_t)a->ev_fork, 0));
    }
    if (own_pack) {
        const FramePackLay pl = frame_pack_lay();
        char* blob = (char*)a->pack_blob;
        if (a->pack_mode == LIDF_FRAME_PACK_GUARDED) {
            rc = frame_pack_guarded(a, blob, (char*)a->pack_guard, two ? sx : st);
            if (two && !rc) {   // (first record of ev_join: the weight streams are valid; a failure joins at the exit)
                if (hipEventRecord((hipEvent_t)a->ev_join, sx) != hipSuccess) rc = LIDF_ERR_HIP;
                guard_on_side = true;
            }
            if (rc) return rc;
        }
        a->packed_query = blob + pl.query;
        pn_loc.packed = blob + pl.pnet;
        pnr_loc.packed = blob + pl.pnet_r;
        a->packed_refine = blob + pl.refine;
    }
    if (two)
        CHECK_HIP(lidf_launch_rayfeat_phase(a->feat_grid, (float*)(ws + f.query + qw.box), B, h, w, a->ray_dir,
                                            a->ray_pix, a->ray_bid, N, a->counts, a->roi_inp_bbox / 2,
                                            a->multires_views, a->rayfeat, 128 + Edv, 1, sx));
    a->pnet = &pn_loc;
    if (rf) a->pnet_refine = &pnr_loc;
    int* counts = a->counts;
    int* cell_flag = (int*)(ws + f.cell_flag);
    int* cell_rank = (int*)(ws + f.cell_rank);
    int* pt_key = (int*)(ws + f.pt_key);
    int* pt_rank = (int*)(ws + f.pt_rank);
    GridSpec g;
    for (int k = 0; k < 3; ++k) { g.xmin[k] = a->xmin[k]; g.r[k] = a->res[k]; }
    g.crop = a->part_size;
    g.B = B;
    const long long hw = (long long)h * w;

    // 0. every zero-initialised scratch of stage 1 in ONE launch: the look-back tickets / status words
    //    and the voxel marks (adjacent), the max-pool tables of the PointNet, the clamped-box list length
    //    and the tile counter of the query
    const int64_t sort_cap = B >= 2 ? 2 * N : 0;   // several frames: the voxel table outgrows the LDS pooling
    const PnetFrameWs pf = pnet_frame_ws(C, v_lds, sort_cap);
    float* pool1 = (float*)(ws + f.pnet + pf.pool1);
    float* pool2 = (float*)(ws + f.pnet + pf.pool2);
    {
        float* zp[6] = {(float*)(ws + f.lb_head), pool1, pool2, (float*)(ws + f.query + qw.counter),
                        (float*)(ws + f.query + qw.box) + grid_floats, (float*)(ws + f.pnet + pf.sort)};
        const long long zc[6] = {(long long)(f.lb_bytes / 4) + (long long)C, (long long)C * 64,
                                 (long long)C * 128, 1, 1, sort_cap > 0 ? (long long)C : 0};
        CHECK_HIP(lidf_launch_zero_segments(zp, zc, 6, st));
    }
    // the query of step 5 (its per-ray launches go to the side stream when there is one)
    LidfQueryArgs q = {};
    q.n_rays = N; q.ray_dir = a->ray_dir; q.ray_pix = a->ray_pix; q.ray_bid = a->ray_bid;
    q.ray_flat = a->ray_flat;
    q.n_pairs = a->max_pairs; q.pair_off = a->pair_off; q.pair_ray = a->pair_ray;
    q.pair_vox = a->pair_vox; q.pair_t = a->pair_t;
    q.batch = B; q.height = h; q.width = w; q.feat_grid = a->feat_grid;
    q.n_vox = C; q.vox_feat = a->occ_voxel_feat;
    q.vox_center = a->pos_rel ? (float*)(ws + f.vox_center) : nullptr;
    q.prob = a->prob; q.off = a->off;
    q.multires = a->multires; q.multires_views = a->multires_views; q.roi_inp_bbox = a->roi_inp_bbox;
    q.pos_rel = a->pos_rel ? 1 : 0;
    q.offset_range0 = a->offset_range0; q.offset_range1 = a->offset_range1; q.part_size = a->part_size;
    q.pred_offset = a->pred_offset; q.pred_prob = a->pred_prob; q.pair_pred_pos = a->pair_pred_pos;
    q.pred_prob_softmax = a->pred_prob_softmax; q.max_pair_id = a->max_pair_id; q.pred_pos = a->pred_pos;
    q.depth = a->pred_depth;
    q.workspace = ws + f.query;
    q.workspace_bytes = lidf_query_workspace_bytes(N, C, (int64_t)B * 32 * h * w);
    q.rayfeat_out = a->rayfeat;
    q.precision = a->precision;
    q.packed = a->packed_query;
    q.offsets_selected = a->offsets_selected;
    // (f32 stage 2: the per-ray part of its decoder's layer 1, W1[:, ROI | dir] rayfeat[r] — constant
    // over the refine iterations — is a third table of the query's layer-1 launch)
    PointsArgs xr = {};
    if (rf && !split) {
        const int Edr = 3 + 6 * a->multires_views;
        xr.stream = (const float*)((const char*)a->packed_refine + linex_stream_bytes(128));
        xr.nets = 1;
        xr.KQ1 = (128 + Edr + 2 + 7) / 8;
        xr.l1_quads = xr.net_quads = xr.KQ1 * 8;
        xr.n = N; xr.n_dev = counts;
        xr.D = 128 + Edr; xr.has_bias = 0;
        xr.out_base = (float*)(ws + f.raypart_r);
    }
    // 1. valid points (with their rows of the PointNet input), rays, depth map, voxel marks: ONE launch
    //    over the pixels (look-back prefix over its workgroups)
    CHECK_HIP(lidf_launch_frame_head(a->valid_mask, a->miss_mask, a->xyz_corrupt, a->rgb, a->intr, B, h, w,
                                     a->valid_stride, g, ws + f.lb_head, counts, a->valid_bid, a->valid_flat,
                                     a->valid_xyz, a->valid_rgb, cell_flag, pt_key, pt_rank, a->ray_bid,
                                     a->ray_flat, a->ray_pix, a->ray_dir, a->pred_depth,
                                     rf ? a->pred_depth_refine : nullptr, a->valid_idx_bid, a->valid_idx_flat,
                                     a->n_valid_idx > 0 ? a->n_valid_idx : 0, st));
    if (two) {
        CHECK_HIP(hipEventRecord((hipEvent_t)a->ev_fork, st));   // (a second record: the rays exist)
        CHECK_HIP(hipStreamWaitEvent(sx, (hipEvent_t)a->ev_fork, 0));
        CHECK_HIP(lidf_launch_rayfeat_phase(a->feat_grid, (float*)(ws + f.query + qw.box), B, h, w, a->ray_dir,
                                            a->ray_pix, a->ray_bid, N, counts, a->roi_inp_bbox / 2,
                                            a->multires_views, a->rayfeat, 128 + Edv, 2, sx));
        // (the per-ray layer-1 tables stay on the main stream: a launch that fills the device starves the
        // PointNet's light launches beside it — measured, fused kernel start 326 -> 354 us)
    }
    FRAME_FAIL_AFTER(1);
    // 2. occupied voxels: cells -> voxels (V) in one workgroup, points -> PointNet rows
    int* vox_bid = (int*)(ws + f.vox_bid);   // [V] image index of every occupied voxel
    float* vox_center = a->pos_rel ? (float*)(ws + f.vox_center) : nullptr;   // intersect_pos_type 'rel'
    int* vox_start = B > 1 ? vox_bid + C : nullptr;   // (one image: every voxel is its image's)
    CHECK_HIP(lidf_launch_frame_cells(cell_flag, C, g, cell_rank, a->occ_bid_coord, a->voxel_bound, vox_bid,
                                      vox_center, counts, vox_start, st));
    // 3. ray / voxel pairs: count, offsets (cut at max_pairs, P) and fill in ONE launch (on the main stream:
    //    behind the per-ray features it made the side branch the longer one — measured)
    CHECK_HIP(lidf_launch_ray_aabb_onepass(a->ray_dir, a->voxel_bound, a->ray_bid, vox_bid, N, counts,
                                           ws + f.lb_pairs, a->pair_off, a->pair_ray, a->pair_vox, a->pair_t,
                                           a->max_pairs, vox_start, st));
    float* pnet_abs = (rf && !a->refine_pnet_pos_rel) ? (float*)(ws + f.pnet_abs) : nullptr;
    CHECK_HIP(lidf_launch_frame_points(a->valid_xyz, a->valid_rgb, pt_key, pt_rank, cell_rank, g, N, counts,
                                       a->valid_v_pid, a->revidx, a->valid_v_rel_coord, a->pnet_inp,
                                       pnet_abs, st));
    FRAME_FAIL_AFTER(2);
    if (guard_on_side) CHECK_HIP(hipStreamWaitEvent(st, (hipEvent_t)a->ev_join, 0));   // weight streams valid
    if (two) CHECK_HIP(hipEventRecord((hipEvent_t)a->ev_join, sx));   // (second record: the per-ray features)
    // 4. voxel embedding: PointNet over the in-grid valid points
    if ((rc = pointnet_frame(a->pnet, a->pnet_inp, a->revidx, N, counts + LIDF_FC_VALID_IN, C, v_lds,
                             counts + LIDF_FC_VOX, a->occ_voxel_feat, ws + f.pnet, cus, st, sort_cap)))
        return rc;
    FRAME_FAIL_AFTER(3);
    if (two) CHECK_HIP(hipStreamWaitEvent(st, (hipEvent_t)a->ev_join, 0));
    // 5. get_embedding + get_pred + depth
    // (with a side stream the stage-2 table — a third of the layer-1 launch, not read before the first refine
    // iteration — leaves that launch and runs beside the per-point kernel: 104 registers and 17 KiB of LDS next
    // to its 220 and 128 KiB, in the matrix-pipe slots its one wavefront per SIMD leaves free, and in its tail)
    const bool xr_aside = two && rf && !split;
    void* const* pev = a->profile_events;   // (benchmarks: HIP events around the matrix launches)
    if ((rc = query_impl(&q, pev ? pev[0] : nullptr, pev ? pev[1] : nullptr, stream, counts,
                         (rf && !split && !xr_aside) ? &xr : nullptr,
                         two ? (QP_RAYTAB | QP_MAIN) : QP_ALL, xr_aside ? a->ev_fork : nullptr)))
        return rc;
    if (xr_aside) {
        const PointsArgs none = {};
        xr.X = a->rayfeat; xr.ldx = 128 + Edv;
        CHECK_HIP(hipStreamWaitEvent(sx, (hipEvent_t)a->ev_fork, 0));   // (third record: the layer-1 launch is out)
        CHECK_HIP(lidf_launch_l1only_pair(none, none, &xr, cus, sx));
        CHECK_HIP(hipEventRecord((hipEvent_t)a->ev_join, sx));          // (third record)
        CHECK_HIP(hipStreamWaitEvent(st, (hipEvent_t)a->ev_join, 0));
    }
    FRAME_FAIL_AFTER(4);
    if (!rf) return LIDF_OK;

    // 6. stage 2: refine_times x get_pred_refine on the device-resident state
    const int E = 3 + 6 * a->multires, Ed = 3 + 6 * a->multires_views;
    const int D = 256 + E + Ed;
    float* inp_embed = (float*)(ws + f.inp_embed);
    float* offv = (float*)(ws + f.off);
    float* vox_feat_r = (float*)(ws + f.vox_feat_r);
    float* voxpart_r = (float*)(ws + f.voxpart_r);
    unsigned char* sel = nullptr;
    if (!a->refine_use_all_pix) {
        sel = (unsigned char*)(ws + f.sel);
        CHECK_HIP(lidf_launch_frame_select(a->valid_mask, a->ray_bid, a->ray_flat, hw, N, counts, sel, st));
    }
    // the stage-2 PointNet reads [valid points | predicted points]: the predicted rows are written
    // behind the NV valid rows of the same buffer (no copy of the valid rows)
    float* pn_inp = a->refine_pnet_pos_rel ? a->pnet_inp : pnet_abs;
    const float* cur = a->pred_pos;
    const float rr0 = a->refine_offset_range0, rrs = a->refine_offset_range1 - a->refine_offset_range0;
    if (!split) {
        // f32: per iteration ONE per-ray launch (the previous iteration's finish, end voxel through the
        // cell table, PointNet rows, embed(pos) rows, the max-pool tables zeroed), the PointNet in four
        // (its last per-voxel launch also forms the voxel columns of the decoder's layer 1), the decoder.
        const VoxTail tail = {(const float*)a->packed_refine, (128 + 2 + 7) / 8, 8, voxpart_r};
        for (int it = 0; it < a->refine_times; ++it) {
            RefineStepArgs sa = {};
            sa.prev_pos = cur;
            sa.prev_off = it > 0 ? offv : nullptr;
            sa.r0 = rr0; sa.rs = rrs;
            sa.cur_pos = it > 0 ? (float*)(ws + ((it & 1) ? f.pos_b : f.pos_a)) : nullptr;
            sa.ray_dir = a->ray_dir; sa.ray_bid = a->ray_bid; sa.ray_flat = a->ray_flat;
            sa.max_pair_id = (const long long*)a->max_pair_id; sa.pair_vox = a->pair_vox;
            sa.vbound = a->voxel_bound; sa.vox_bid = vox_bid;
            sa.g = g; sa.cell_flag = cell_flag; sa.cell_rank = cell_rank;
            sa.rgb = a->rgb; sa.hw = hw;
            sa.pnet_rel = a->refine_pnet_pos_rel; sa.pos_rel = a->refine_pos_rel; sa.L = a->multires;
            sa.pnet_inp = pn_inp; sa.pnet_vox = a->revidx; sa.sel = sel;
            sa.end_voxel = a->end_voxel_id; sa.inp_embed = inp_embed; sa.ld_e = D;
            sa.dims = counts; sa.row0_dev = counts + LIDF_FC_VALID_IN;
            sa.zero0 = pool1; sa.nzero0 = (long long)C * 64;
            sa.zero1 = pool2; sa.nzero1 = (long long)C * 128;
            CHECK_HIP(lidf_launch_refine_step(sa, N, st));
            if (it > 0) cur = sa.cur_pos;
            if ((rc = pointnet_frame(a->pnet_refine, pn_inp, a->revidx, 2 * N, counts + LIDF_FC_PNET_REFINE, C,
                                     v_lds, counts + LIDF_FC_VOX, vox_feat_r, ws + f.pnet, cus, st, sort_cap,
                                     &tail)))
                return rc;
            if ((rc = refine_ief_factorised(a->off_refine, D, vox_feat_r, C, inp_embed, a->end_voxel_id, N,
                                            offv, voxpart_r, (char*)a->packed_refine, st, 2,
                                            counts + LIDF_FC_RAYS, counts + LIDF_FC_VOX,
                                            (pev && it < 2) ? pev + 2 + 2 * it : nullptr, a->rayfeat,
                                            Ed, (float*)(ws + f.raypart_r), false, true)))
                return rc;
        }
        CHECK_HIP(lidf_launch_refine_finish_dev(cur, offv, a->ray_dir, rr0, rrs, N, counts, a->pred_pos_refine,
                                                a->ray_bid, a->ray_flat, hw, a->pred_depth_refine, st));
        return LIDF_OK;
    }
    for (int it = 0; it < a->refine_times; ++it) {   // split-f16 products: whole decoder rows, layer by layer
        float* out = it == a->refine_times - 1 ? a->pred_pos_refine
                                               : (float*)(ws + ((it & 1) ? f.pos_b : f.pos_a));
        {   // this iteration's zero-initialised scratch in one launch: end voxels + the max-pool tables
            float* zp[3] = {(float*)a->end_voxel_id, pool1, pool2};
            const long long zc[3] = {(long long)N, (long long)C * 64, (long long)C * 128};
            CHECK_HIP(lidf_launch_zero_segments(zp, zc, 3, st));
        }
        CHECK_HIP(lidf_launch_refine_prep_dev(cur, (const long long*)a->max_pair_id, a->pair_vox, a->max_pairs,
                                              a->voxel_bound, vox_bid, C, a->ray_bid, a->ray_flat, a->rgb, hw,
                                              a->refine_pnet_pos_rel, N, pn_inp, a->revidx, a->end_voxel_id,
                                              sel, counts, counts + LIDF_FC_VALID_IN, st, nullptr));
        CHECK_HIP(lidf_launch_refine_rows_dev(cur, a->end_voxel_id, a->voxel_bound, a->rayfeat, 128 + Ed,
                                              a->multires_views, a->multires, a->refine_pos_rel, N, counts,
                                              inp_embed, D, 0, st));
        if ((rc = pointnet_frame(a->pnet_refine, pn_inp, a->revidx, 2 * N, counts + LIDF_FC_PNET_REFINE, C,
                                 v_lds, counts + LIDF_FC_VOX, vox_feat_r, ws + f.pnet, cus, st, sort_cap)))
            return rc;
        // (the voxel feature gathered into the rows, the weights packed per call)
        CHECK_HIP(lidf_launch_refine_gather_dev(vox_feat_r, a->end_voxel_id, N, counts, inp_embed, D, st));
        if ((rc = decoders_impl(inp_embed, N, D, D, nullptr, a->off_refine, nullptr, offv, ws + f.dec,
                                lidf_decoders_workspace_bytes(N, D), LIDF_PRECISION_F16X3, stream,
                                counts + LIDF_FC_RAYS)))
            return rc;
        const bool last = it == a->refine_times - 1;
        CHECK_HIP(lidf_launch_refine_finish_dev(cur, offv, a->ray_dir, rr0, rrs, N, counts,
                                                out, a->ray_bid, a->ray_flat, hw,
                                                last ? a->pred_depth_refine : nullptr, st));
        cur = out;
    }
    return LIDF_OK;
}

// ---- occupied-voxel build ----------------------------------------------------------------------
struct VoxWs {
    size_t cell_flag, cell_rank, pt_key, pt_valid, pt_rank, scan, total;
};
static VoxWs vox_ws(int64_t n, int64_t ncell) {
    VoxWs w;
    size_t o = 0;
    const size_t N = (size_t)(n > 0 ? n : 1), C = (size_t)(ncell > 0 ? ncell : 1);
    w.cell_flag = o; o += align_up(C * 4, 256);
    w.cell_rank = o; o += align_up((C + 1) * 4, 256);
    w.pt_key = o;    o += align_up(N * 4, 256);
    w.pt_valid = o;  o += align_up(N * 4, 256);
    w.pt_rank = o;   o += align_up((N + 1) * 4, 256);
    w.scan = o;      o += align_up(lidf_exclusive_scan_workspace_bytes(N > C ? (int64_t)N : (int64_t)C), 256);
    w.total = o;
    return w;
}

LIDF_API size_t lidf_voxelize_workspace_bytes(int64_t n_pts, int64_t n_cells) {
    return vox_ws(n_pts, n_cells).total;
}

LIDF_API int lidf_voxelize_f32(const float* xyz, const int32_t* bid, int64_t n, int batch,
                               const float* xmin, const int32_t* res, float crop,
                               int32_t* occ_bid_coord, float* voxel_bound, int32_t* valid_pid,
                               int32_t* revidx, float* rel_coord, int32_t* counts, void* workspace,
                               size_t workspace_bytes, lidf_stream_t stream) {
    if (n < 0 || batch <= 0 || !xmin || !res || !(crop > 0.f) || !counts) return LIDF_ERR_BAD_ARG;
    if (res[0] <= 0 || res[1] <= 0 || res[2] <= 0) return LIDF_ERR_BAD_ARG;
    const int64_t ncell = (int64_t)batch * res[0] * res[1] * res[2];
    if (ncell > 0x7fffffffLL || n > 0x7fffffffLL) return LIDF_ERR_UNSUPPORTED;
    if (n > 0 && (!xyz || !bid || !valid_pid || !revidx || !rel_coord)) return LIDF_ERR_BAD_ARG;
    if (!occ_bid_coord || !voxel_bound) return LIDF_ERR_BAD_ARG;
    VoxWs w = vox_ws(n, ncell);
    if (!workspace || workspace_bytes < w.total) return LIDF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    int* cell_flag = (int*)(ws + w.cell_flag);
    int* cell_rank = (int*)(ws + w.cell_rank);
    int* pt_key = (int*)(ws + w.pt_key);
    int* pt_valid = (int*)(ws + w.pt_valid);
    int* pt_rank = (int*)(ws + w.pt_rank);
    GridSpec g;
    for (int a = 0; a < 3; ++a) { g.xmin[a] = xmin[a]; g.r[a] = res[a]; }
    g.crop = crop;
    g.B = batch;
    CHECK_HIP(hipMemsetAsync(cell_flag, 0, (size_t)ncell * 4, st));
    CHECK_HIP(lidf_launch_vox_mark(xyz, bid, n, g, cell_flag, pt_key, pt_valid, st));
    CHECK_HIP(lidf_launch_scan(cell_flag, ncell, cell_rank, (int*)(ws + w.scan), st));
    CHECK_HIP(lidf_launch_scan(pt_valid, n, pt_rank, (int*)(ws + w.scan), st));
    CHECK_HIP(lidf_launch_vox_cells(cell_flag, cell_rank, ncell, g, occ_bid_coord, voxel_bound, st));
    CHECK_HIP(lidf_launch_vox_points(xyz, pt_key, pt_rank, cell_rank, n, g, valid_pid, revidx,
                                     rel_coord, st));
    CHECK_HIP(hipMemcpyAsync(counts, cell_rank + ncell, 4, hipMemcpyDeviceToDevice, st));
    CHECK_HIP(hipMemcpyAsync(counts + 1, pt_rank + (n > 0 ? n : 0), 4, hipMemcpyDeviceToDevice, st));
    return LIDF_OK;
}

// ---- eval depth metrics ---------------------------------------------------------------------------
LIDF_API size_t lidf_depth_metrics_workspace_bytes(void) { return lidf_depth_metrics_ws_bytes(); }

LIDF_API int lidf_depth_metrics_f32(const float* pred_depth, const float* gt_depth,
                                      const void* seg_mask, int32_t seg_dtype, int32_t src_h, int32_t src_w,
                                      int32_t dst_h, int32_t dst_w, float* out, void* workspace,
                                      size_t workspace_bytes, lidf_stream_t stream) {
    if (src_h <= 0 || src_w <= 0 || dst_h <= 0 || dst_w <= 0) return LIDF_ERR_BAD_ARG;
    if (!pred_depth || !gt_depth || !out) return LIDF_ERR_BAD_ARG;
    if (seg_dtype < 0 || seg_dtype > 2 || (seg_dtype != 0 && !seg_mask)) return LIDF_ERR_BAD_ARG;
    if ((int64_t)dst_h * dst_w > 0x7fffffffLL) return LIDF_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < lidf_depth_metrics_ws_bytes()) return LIDF_ERR_WORKSPACE;
    CHECK_HIP(lidf_launch_depth_metrics(pred_depth, gt_depth, seg_mask, seg_mask ? seg_dtype : 0, src_h, src_w,
                                        dst_h, dst_w, out, workspace, (hipStream_t)stream));
    return LIDF_OK;
}

// ---- decoders, training path (forward with kept activations + backward) ---------------------------
// One linear layer through lidf_linear_kernel with the training epilogues.
struct LinEx {
    const float* w; const float* b; int ldw;   // weight [rows, ldw] (+ bias); see `transposed`
    int nout, k;                               // outputs and contraction length
    int transposed;                            // outputs index weight columns, k indexes rows (W^T)
    const LidfDecoder* ief;                    // layer 1 of an IEF: bias += c, column k+1 = u
    const float* X; long long ldx; long long n;
    const int* n_dev;                          // optional device-side row count (n = capacity)
    int c0, k1, c1;                            // operand columns [0,k) -> weight columns c0.., [k,k+k1) -> c1..
    int dcore;                                 // IEF: weight column of the offset encoding (default k)
    const float* addrows; const int* addidx;   // gathered 256-wide terms added before the activation
    const float* addrows2; const int* addidx2;
    const float* xoff;
    int relu; float slope;
    const float* mask_src; long long ld_mask; float mask_slope;
    float* out; long long ld_out; int accumulate;
    // transposed launches only: operand columns [k, k + k2) come from X2 (row stride ldx2) and multiply rows
    // [0, k2) of a SECOND weight matrix w_2 (row stride ldw_2) — one product over two decoders' S (k % 8 == 0)
    const float* X2; long long ldx2; int k2; const float* w_2; int ldw_2;
    // nout = 32 t + 1 (plain transposed products, many rows): the last column through the vector unit beside the
    // t tiles instead of a tile of its own (LinearArgs.xcol)
    int xcol;
};

static int nt_for(int nout) { const int t = (nout + 31) / 32; return t < 1 ? 1 : t > 8 ? 8 : t; }

static size_t linex_stream_bytes(int k) { return align_up((size_t)((k + 2 + 7) / 8) * 8 * 1024, 256); }
// NOTE: the bias / u columns exist in the stream only when the layer has them; a layer with a
// second operand segment passes k = k + k1 here.

static int run_linex(const LinEx& L, float* stream_buf, int cus, hipStream_t st, int pack_mode = 0) {
    if (L.n <= 0 && pack_mode != 1) return LIDF_OK;
    if (L.xcol && (L.nout % 32 != 1 || L.nout < 33 || L.nout > 257 || !L.transposed || L.b || L.ief)) return LIDF_ERR_BAD_ARG;
    const int nt = nt_for(L.xcol ? L.nout - 1 : L.nout);
    if (L.xcol && nt + 1 > 8) return LIDF_ERR_BAD_ARG;   // (the stream slots are sized for 8 quads per k-quad)
    const bool two = L.X2 != nullptr;
    if (two && (!L.transposed || L.k % 8 || L.k2 % 8 || L.k2 <= 0 || L.k1 || L.b || L.ief || !L.w_2)) return LIDF_ERR_BAD_ARG;
    L1Map m = two ? rows_map(L.k, L.c0, L.k2, 0, 0) : rows_map(L.k, L.c0, L.k1, L.c1, L.b ? 1 : 0);
    m.KQ1 = (m.D + 2 + 7) / 8;   // room for the bias and the u column
    m.nt = nt + (L.xcol ? 1 : 0);
    m.xcol = L.xcol ? 1 : 0;
    m.nout = L.nout;
    m.transposed = two ? 2 : L.transposed;
    m.add_u = L.ief ? 1 : 0;
    StreamLayout lay = lidf_make_layout(1, LIDF_MODE_LINEAR, m);
    NetW nw = {};
    nw.w1 = L.w; nw.b1 = L.b; nw.ld1 = L.ldw; nw.dcore = L.dcore ? L.dcore : L.k; nw.is_ief = 0;
    if (L.ief) { nw.is_ief = 1; nw.wenc = L.ief->wenc; nw.benc = L.ief->benc; }
    NetW nw2 = nw;
    if (two) { nw2.w1 = L.w_2; nw2.ld1 = L.ldw_2; }
    if (pack_mode != 2) CHECK_HIP(pack_stream(lay, nw, nw2, m, stream_buf, nullptr, st));
    if (pack_mode == 1) return LIDF_OK;
    LinearArgs a = {};
    a.stream = stream_buf; a.kq1 = m.KQ1; a.X = L.X; a.ldx = L.ldx; a.n = L.n;
    if (two) { a.X2 = L.X2; a.ldx2 = L.ldx2; a.kq_split = L.k / 8; }
    a.n_dev = L.n_dev;
    a.D = m.D; a.has_bias = L.b ? 1 : 0; a.xoff = L.xoff;
    a.addrows = L.addrows; a.addidx = L.addidx; a.ld_add = L.nout;
    a.addrows2 = L.addrows2; a.addidx2 = L.addidx2; a.ld_add2 = L.nout;
    a.relu = L.relu; a.slope = L.slope;
    a.mask_src = L.mask_src; a.ld_mask = L.ld_mask; a.mask_slope = L.mask_slope;
    a.out = L.out; a.ld_out = L.ld_out; a.nout = L.xcol ? L.nout - 1 : L.nout; a.accumulate = L.accumulate;
    a.xcol = L.xcol ? 1 : 0;
    const long long nt128 = (L.n + 127) / 128;
    const int grid = (int)(nt128 < 4LL * cus ? nt128 : 4LL * cus);
    CHECK_HIP(lidf_launch_linear(nt, a, grid, st));
    return LIDF_OK;
}

// ---- RoIAlign of the per-ray boxes at any channel count / output size ------------------------------
LIDF_API int lidf_roi_align_f32(const float* feat_grid, int32_t batch, int32_t channels, int32_t height,
                                  int32_t width, const int32_t* ray_pix, const int32_t* ray_bid,
                                  int64_t n_rays, int32_t roi_inp_bbox, int32_t roi_out_bbox, float* out,
                                  int64_t ld_out, lidf_stream_t stream) {
    if (batch <= 0 || channels <= 0 || height <= 0 || width <= 0 || n_rays < 0 || roi_inp_bbox < 0 ||
        roi_out_bbox <= 0 || roi_out_bbox > 64)
        return LIDF_ERR_BAD_ARG;
    if (n_rays == 0) return LIDF_OK;
    if (!feat_grid || !ray_pix || !ray_bid || !out || ld_out < (int64_t)channels * roi_out_bbox * roi_out_bbox)
        return LIDF_ERR_BAD_ARG;
    if ((int64_t)height * width > 0x7fffffffLL || n_rays > 0x7fffffffLL) return LIDF_ERR_UNSUPPORTED;
    CHECK_HIP(lidf_launch_roi_align(feat_grid, channels, height, width, ray_pix, ray_bid, n_rays,
                                    roi_inp_bbox / 2, roi_out_bbox, out, ld_out, (hipStream_t)stream));
    return LIDF_OK;
}

// ---- a linear layer of any width (the modules at widths other than the shipped ones) ---------------
LIDF_API size_t lidf_linear_workspace_bytes(int32_t k) {
    return k > 0 ? linex_stream_bytes(k) : 0;
}

static int linear_impl(const float* x, int64_t ldx, int64_t n, int32_t k, const float* w, int64_t ldw,
                       const float* b, int32_t nout, int32_t act, float slope,
                       const float* addrows, const int32_t* addidx, int64_t ld_add,
                       const float* addrows2, const int32_t* addidx2, int64_t ld_add2, float* out,
                       int64_t ld_out, float* pool, const int32_t* poolidx, int64_t ld_pool,
                       void* workspace, size_t workspace_bytes, lidf_stream_t stream);

LIDF_API int lidf_linear_f32(const float* x, int64_t ldx, int64_t n, int32_t k, const float* w, int64_t ldw,
                               const float* b, int32_t nout, int32_t act, float slope,
                               const float* addrows, const int32_t* addidx, int64_t ld_add, float* out,
                               int64_t ld_out, float* pool, const int32_t* poolidx, int64_t ld_pool,
                               void* workspace, size_t workspace_bytes, lidf_stream_t stream) {
    return linear_impl(x, ldx, n, k, w, ldw, b, nout, act, slope, addrows, addidx, ld_add, nullptr, nullptr, 0, out,
                       ld_out, pool, poolidx, ld_pool, workspace, workspace_bytes, stream);
}

LIDF_API int lidf_linear_gather2_f32(const float* x, int64_t ldx, int64_t n, int32_t k, const float* w, int64_t ldw,
                                       const float* b, int32_t nout, int32_t act, float slope,
                                       const float* addrows, const int32_t* addidx, int64_t ld_add,
                                       const float* addrows2, const int32_t* addidx2, int64_t ld_add2, float* out,
                                       int64_t ld_out, void* workspace, size_t workspace_bytes,
                                       lidf_stream_t stream) {
    if (!addrows || !addrows2) return LIDF_ERR_BAD_ARG;
    return linear_impl(x, ldx, n, k, w, ldw, b, nout, act, slope, addrows, addidx, ld_add, addrows2, addidx2, ld_add2,
                       out, ld_out, nullptr, nullptr, 0, workspace, workspace_bytes, stream);
}

static int linear_impl(const float* x, int64_t ldx, int64_t n, int32_t k, const float* w, int64_t ldw,
                       const float* b, int32_t nout, int32_t act, float slope,
                       const float* addrows, const int32_t* addidx, int64_t ld_add,
                       const float* addrows2, const int32_t* addidx2, int64_t ld_add2, float* out,
                       int64_t ld_out, float* pool, const int32_t* poolidx, int64_t ld_pool,
                       void* workspace, size_t workspace_bytes, lidf_stream_t stream) {
    if (n < 0 || k <= 0 || nout <= 0 || ldx < k || ldw < k || act < 0 || act > 1) return LIDF_ERR_BAD_ARG;
    if (k > (1 << 20) || nout > (1 << 20) || ld_add > 0x7fffffffLL || ld_add2 > 0x7fffffffLL ||
        ld_pool > 0x7fffffffLL)
        return LIDF_ERR_UNSUPPORTED;
    if (n == 0) return LIDF_OK;
    if (!x || !w || (!out && !pool)) return LIDF_ERR_BAD_ARG;
    if (out && ld_out < nout) return LIDF_ERR_BAD_ARG;
    if (addrows && (!addidx || ld_add < nout)) return LIDF_ERR_BAD_ARG;
    if (addrows2 && (!addidx2 || ld_add2 < nout)) return LIDF_ERR_BAD_ARG;
    // the max-pool epilogue raises whole 32-column tiles of non-negative values
    if (pool && (!poolidx || ld_pool < nout || nout % 32 != 0 || !act || slope != 0.f)) return LIDF_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < linex_stream_bytes(k)) return LIDF_ERR_WORKSPACE;
    int rc, cus;
    if ((rc = cu_count(&cus))) return rc;
    hipStream_t st = (hipStream_t)stream;
    // 256 output columns per launch (8 accumulator tiles); the packed stream of a chunk is consumed by
    // its launch before the next chunk's pack overwrites it (same stream)
    // (evenly sized launches: 385 columns are 7 + 6 tiles, not 8 + 5)
    const int tiles_all = (nout + 31) / 32, launches = (tiles_all + 7) / 8;
    const int cols_per = 32 * ((tiles_all + launches - 1) / launches);
    for (int c0 = 0; c0 < nout; c0 += cols_per) {
        const int cols = nout - c0 < cols_per ? nout - c0 : cols_per;
        const int nt = nt_for(cols);
        L1Map m = rows_map(k, 0, 0, 0, b ? 1 : 0);
        m.KQ1 = (m.D + 2 + 7) / 8;
        m.nt = nt;
        m.nout = cols;
        StreamLayout lay = lidf_make_layout(1, LIDF_MODE_LINEAR, m);
        NetW nw = {};
        nw.w1 = w + (size_t)c0 * ldw; nw.b1 = b ? b + c0 : nullptr; nw.ld1 = (int)ldw; nw.dcore = k; nw.is_ief = 0;
        CHECK_HIP(pack_stream(lay, nw, nw, m, (float*)workspace, nullptr, st));
        LinearArgs a = {};
        a.stream = (const float*)workspace; a.kq1 = m.KQ1; a.X = x; a.ldx = ldx; a.n = n;
        a.D = m.D; a.has_bias = b ? 1 : 0;
        a.addrows = addrows ? addrows + c0 : nullptr; a.addidx = addidx; a.ld_add = (int)ld_add;
        a.addrows2 = addrows2 ? addrows2 + c0 : nullptr; a.addidx2 = addidx2; a.ld_add2 = (int)ld_add2;
        a.relu = act; a.slope = slope;
        a.out = out ? out + c0 : nullptr; a.ld_out = ld_out; a.nout = cols;
        a.pool = pool ? pool + c0 : nullptr; a.poolidx = poolidx; a.ld_pool = (int)ld_pool;
        const long long nt128 = (n + 127) / 128;
        const int grid = (int)(nt128 < 4LL * cus ? nt128 : 4LL * cus);
        CHECK_HIP(lidf_launch_linear(nt, a, grid, st));
    }
    return LIDF_OK;
}

// ---- a whole decoder at gf_dim 32 / 64 / 128 as one register-chained launch (lidf_chain16.hip) ------------
static size_t chain16_stream_bytes(int gf, int k) {
    const int G = gf / 16;
    return (size_t)(((k + 15) / 16) * 4 * G + lidf_chain16_pass_quads(G)) * 1024;
}
LIDF_API size_t lidf_decoder_chain_workspace_bytes(int32_t gf_dim, int32_t k) {
    if ((gf_dim != 32 && gf_dim != 64 && gf_dim != 128) || k <= 0 || k > (1 << 16)) return 0;
    return align_up(chain16_stream_bytes(gf_dim, k), 256) + align_up((size_t)lidf_chain16_aux_floats(gf_dim / 16) * 4, 256);
}

LIDF_API int lidf_decoder_chain_f32(const LidfDecoder* dec, int32_t gf_dim, int32_t inp_dim, const float* x,
                                      int64_t ldx, int32_t k, int32_t w1_col0, int64_t n, const float* voxpart,
                                      const int32_t* vox_idx, const float* raypart, const int32_t* ray_idx,
                                      float* out, int32_t prepacked, void* workspace, size_t workspace_bytes,
                                      lidf_stream_t stream) {
    if (!dec || n < 0 || k <= 0 || ldx < k || inp_dim <= 0 || w1_col0 < 0 || w1_col0 + k > inp_dim)
        return LIDF_ERR_BAD_ARG;
    if (gf_dim != 32 && gf_dim != 64 && gf_dim != 128) return LIDF_ERR_UNSUPPORTED;
    if (!dec->w1 || !dec->w2 || !dec->b2 || !dec->w3 || !dec->b3 || !dec->w4 || !dec->b4) return LIDF_ERR_BAD_ARG;
    if (dec->is_ief && (!dec->wenc || !dec->benc || dec->n_iter < 1 || dec->n_iter > 64)) return LIDF_ERR_BAD_ARG;
    if (n == 0) return LIDF_OK;
    if (!x || !out || (vox_idx && !voxpart) || (ray_idx && !raypart)) return LIDF_ERR_BAD_ARG;
    if (k > (1 << 16)) return LIDF_ERR_UNSUPPORTED;
    const size_t need = lidf_decoder_chain_workspace_bytes(gf_dim, k);
    if (!workspace || workspace_bytes < need) return LIDF_ERR_WORKSPACE;
    int rc, cus;
    if ((rc = cu_count(&cus))) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int G = gf_dim / 16;
    float* sbuf = (float*)workspace;
    float* aux = (float*)((char*)workspace + align_up(chain16_stream_bytes(gf_dim, k), 256));
    StreamLayout lay = {};
    lay.nets = 1; lay.mode = LIDF_MODE_CHAIN16;
    lay.total = (int)(chain16_stream_bytes(gf_dim, k) / 4);
    L1Map m = {};
    m.n0 = k; m.c0 = w1_col0; m.nt = G;
    const NetW nw = to_netw(dec, inp_dim);
    // (prepacked: the workspace still holds this decoder's stream for this k / w1_col0 — the slabs of one query)
    if (!prepacked) CHECK_HIP(pack_stream(lay, nw, nw, m, sbuf, aux, st));
    Chain16Args a = {};
    a.stream = sbuf; a.aux = aux; a.KQ = (k + 15) / 16; a.E = k; a.n = n;
    a.X = x; a.ldx = ldx;
    a.vox = vox_idx; a.voxpart = voxpart; a.ray = ray_idx; a.raypart = raypart;
    a.npass = dec->is_ief ? dec->n_iter : 1;
    a.init = dec->is_ief ? dec->init_offset : 0.f;
    a.sigmoid = dec->use_sigmoid;
    a.out = out;
    CHECK_HIP(lidf_launch_chain16(gf_dim, a, cus, st));
    return LIDF_OK;
}

// partial blocks of the weight-gradient reduction: 512 row slices x (128 x 256 block + its bias part)
#define WG_SCRATCH_FLOATS ((size_t)512 * LIDF_WG_SLAB)
#define ACT_ROW_FLOATS LIDF_ACT_ROW_FLOATS   // per row and pass: H1 | H2 | H3 | offset in | sign words (lidf_device.h)

// ---- weight gradient of a linear layer of any width: C += A^T B, db += column sums of A -----------
LIDF_API size_t lidf_wgrad_workspace_bytes(void) { return WG_SCRATCH_FLOATS * 4; }

LIDF_API int lidf_wgrad_f32(const float* a, int64_t lda, int32_t m, const float* b, int64_t ldb, int32_t n_cols,
                              int64_t n, float* c, int64_t ldc, float* db, void* workspace,
                              size_t workspace_bytes, lidf_stream_t stream) {
    if (n < 0 || m <= 0 || n_cols <= 0 || lda < m || ldb < n_cols || ldc < n_cols) return LIDF_ERR_BAD_ARG;
    if (m > (1 << 20) || n_cols > (1 << 20) || ldc > 0x7fffffffLL) return LIDF_ERR_UNSUPPORTED;
    if (n == 0) return LIDF_OK;
    if (!a || !b || !c) return LIDF_ERR_BAD_ARG;
    // without the scratch area the partial blocks are added with float atomics (order not fixed)
    float* wgs = (workspace && workspace_bytes >= WG_SCRATCH_FLOATS * 4) ? (float*)workspace : nullptr;
    CHECK_HIP(lidf_launch_wgrad(a, lda, m, b, ldb, n_cols, n, c, (int)ldc, db, wgs, wgs ? WG_SCRATCH_FLOATS : 0,
                                (hipStream_t)stream));
    return LIDF_OK;
}

LIDF_API size_t lidf_decoder_train_act_floats(int64_t n, int32_t n_pass) {
    if (n <= 0 || n_pass <= 0) return 0;
    return (size_t)n * ((size_t)n_pass * ACT_ROW_FLOATS + 1);  // + the pre-activation output
}

// ---- training forward as ONE launch per decoder: the register-chained rows kernel in the mode that
// keeps H1 | H2 | H3 | offset-in of every pass (LIDF_MODE_TRAIN, lidf_points.hip) -----------------
static size_t chain_stream_bytes(int D) {
    L1Map m = rows_map(D, 0, 0, 0, 1);
    return align_up((size_t)lidf_make_layout(1, LIDF_MODE_ROWS, m).total * 4, 256) +
           align_up(LIDF_AUX_FLOATS * 4, 256);
}
// X [n, m.D] (row stride ldx) are the layer-1 operand rows named by `m`; voxpart / raypart (or NULL)
// are added through pair_vox / pair_ray. passes: npass x ACT_ROW_FLOATS x n, pre [n], out [n].
static int run_chain_train(const LidfDecoder* dec, int dcore, const L1Map& m, const float* X,
                           int64_t ldx, int64_t n, const int32_t* pair_vox, const int32_t* pair_ray,
                           const float* voxpart, const float* raypart, float* passes, float* pre,
                           float* out, char* sbuf, int cus, hipStream_t st,
                           int mode = LIDF_MODE_TRAIN, int pack_mode = 0, const int* n_dev = nullptr) {
    const StreamLayout lay = lidf_make_layout(1, LIDF_MODE_ROWS, m);
    float* stream_buf = (float*)sbuf;
    float* aux = (float*)(sbuf + align_up((size_t)lay.total * 4, 256));
    const NetW nw = to_netw(dec, dcore);
    if (pack_mode != 2) CHECK_HIP(pack_stream(lay, nw, nw, m, stream_buf, aux, st));
    if (pack_mode == 1) return LIDF_OK;
    PointsArgs a = {};
    a.stream = stream_buf; a.aux = aux; a.nets = 1;
    a.l1_quads = lay.l1_quads; a.net_quads = lay.net_quads; a.n = n;
    a.n_dev = n_dev;
    fill_net_args(a, 0, dec, out, 0);
    a.X = X; a.ldx = ldx; a.D = m.D; a.KQ1 = m.KQ1; a.has_bias = m.add_bias;
    a.pair_vox = pair_vox; a.pair_ray = pair_ray; a.voxpart = voxpart; a.raypart = raypart;
    a.tr_passes[0] = passes; a.tr_pass_floats = (long long)n * ACT_ROW_FLOATS; a.tr_pre[0] = pre;
#ifdef LIDF_PROFILE
    a.out_base = pre;   // development build: the phase counters land in the first 16 floats
#endif
    const long long ntile = (n + 127) / 128;
    CHECK_HIP(lidf_launch_points(mode, a, (int)(ntile < cus ? ntile : cus), st));
    return LIDF_OK;
}

// scratch of refine_ief_factorised: [layer-1 stream of the per-voxel launch | of the per-ray launch (ROI +
// direction columns, sized for the widest direction embedding) | stream + aux of the chain]
#define REFINE_RAY_K (128 + 3 + 6 * 16)
// ... | stream + aux of the 16 x 16 x 4 decoder (lidf_ief16.hip): E <= 99 embed(pos) columns = 7 k-quads]
static size_t ief16_stream_bytes(int E) { return (size_t)(((E + 15) / 16) * 16 + IEF16_PASS_QUADS) * 1024; }
static size_t refine_fact_bytes(int D) {
    return linex_stream_bytes(128) + linex_stream_bytes(REFINE_RAY_K) + chain_stream_bytes(D - 128) +
           align_up(ief16_stream_bytes(3 + 6 * 16), 256) + align_up(IEF16_AUX_FLOATS * 4, 256);
}
// LIDF_IEF16=0 in the environment: the 32 x 32 rows kernel of rounds 2-3 instead (A/B measurements)
static bool use_ief16() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("LIDF_IEF16");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}

static int refine_ief_factorised(const LidfDecoder* off, int D, const float* vox_feat, int64_t V,
                                 const float* inp_embed, const int32_t* end_voxel, int64_t R,
                                 float* out, float* voxpart, char* scratch, hipStream_t st,
                                 int pack_mode, const int* R_dev, const int* V_dev, void* const* ev_rows,
                                 const float* rayfeat, int Ed, float* raypart, bool make_raypart,
                                 bool voxpart_ready) {
    // Layer 1 of the stage-2 decoder on [vox feat 128 | ROI 128 | embed(pos) E | embed(dir) Ed] as three
    // partial products: per voxel (W1[:, 0:128] vox_feat[v] + b1 (+c), gathered by the end voxel), per
    // ray (W1[:, ROI | dir] rayfeat[r]: constant over the refine iterations — the frame path forms it
    // once per frame) and, per iteration, only the E columns of embed(pos): 7 k-quads instead of 26.
    int rc, cus;
    if ((rc = cu_count(&cus))) return rc;
    const int ld1 = D + (off->is_ief ? 16 : 0);
    const int E = D - 256 - Ed;
    if (E < 0 || Ed < 0 || 128 + Ed > REFINE_RAY_K) return LIDF_ERR_UNSUPPORTED;
    char* s_vox = scratch;
    char* s_ray = scratch + linex_stream_bytes(128);
    char* s_chain = s_ray + linex_stream_bytes(REFINE_RAY_K);
    LinEx L = {};
    L.w = off->w1; L.b = off->b1; L.ldw = ld1; L.nout = LIDF_H1; L.k = 128; L.dcore = D;
    L.ief = off->is_ief ? off : nullptr;   // bias += c
    L.X = vox_feat; L.ldx = 128; L.n = V; L.out = voxpart; L.ld_out = LIDF_H1;
    L.n_dev = V_dev;
    // (voxpart_ready: the frame path forms the per-voxel rows in the launch of the PointNet's last
    // per-voxel layer — VoxTail, the same stream)
    if (!(voxpart_ready && pack_mode == 2) && (rc = run_linex(L, (float*)s_vox, cus, st, pack_mode))) return rc;
    if (make_raypart || pack_mode == 1) {
        LinEx Lr = {};
        Lr.w = off->w1; Lr.b = nullptr; Lr.ldw = ld1; Lr.nout = LIDF_H1;
        Lr.k = 128; Lr.c0 = 128;                 // ROI columns
        Lr.k1 = Ed; Lr.c1 = 256 + E;             // direction embedding
        Lr.dcore = D;
        Lr.X = rayfeat; Lr.ldx = 128 + Ed; Lr.n = R; Lr.out = raypart; Lr.ld_out = LIDF_H1;
        Lr.n_dev = R_dev;
        if ((rc = run_linex(Lr, (float*)s_ray, cus, st, pack_mode))) return rc;
    }
    // the decoder on 16-ray sub-tiles (lidf_ief16.hip): its stream rides behind the others
    char* s16 = s_chain + chain_stream_bytes(D - 128);
    float* aux16 = (float*)(s16 + align_up(ief16_stream_bytes(3 + 6 * 16), 256));
    if (pack_mode != 2) {
        StreamLayout lay = {};
        lay.nets = 1; lay.mode = LIDF_MODE_IEF16;
        lay.total = (int)(ief16_stream_bytes(E) / 4);
        L1Map m16 = {};
        m16.n0 = E; m16.c0 = 256;
        const NetW nw = to_netw(off, D);
        CHECK_HIP(pack_stream(lay, nw, nw, m16, (float*)s16, aux16, st));
    }
    if (!use_ief16() || pack_mode == 1) {
        if (ev_rows && ev_rows[0] && pack_mode != 1) CHECK_HIP(hipEventRecord((hipEvent_t)ev_rows[0], st));
        rc = run_chain_train(off, D, rows_map(E, 256, 0, 0, 0), inp_embed ? inp_embed + 256 : nullptr, D, R,
                             end_voxel, nullptr, voxpart, raypart, nullptr, nullptr, out, s_chain, cus, st,
                             LIDF_MODE_ROWS_GATHER, pack_mode, R_dev);
        if (!rc && ev_rows && ev_rows[1] && pack_mode != 1) CHECK_HIP(hipEventRecord((hipEvent_t)ev_rows[1], st));
        return rc;
    }
    if (R <= 0) return LIDF_OK;
    if (ev_rows && ev_rows[0]) CHECK_HIP(hipEventRecord((hipEvent_t)ev_rows[0], st));   // (benchmarks only)
    Ief16Args ia = {};
    ia.stream = (const float*)s16; ia.aux = aux16; ia.KQ = (E + 15) / 16; ia.E = E;
    ia.n = R; ia.n_dev = R_dev;
    ia.X = inp_embed + 256; ia.ldx = D;
    ia.vox = end_voxel; ia.voxpart = voxpart; ia.raypart = raypart;
    ia.npass = off->is_ief ? off->n_iter : 1;
    ia.init = off->is_ief ? off->init_offset : 0.f;
    ia.sigmoid = off->use_sigmoid;
    ia.out = out;
    CHECK_HIP(lidf_launch_ief16(ia, cus, st));
    if (ev_rows && ev_rows[1]) CHECK_HIP(hipEventRecord((hipEvent_t)ev_rows[1], st));
    return LIDF_OK;
}

LIDF_API size_t lidf_refine_pack_bytes(int32_t multires, int32_t multires_views) {
    if (multires < 0 || multires > 16 || multires_views < 0 || multires_views > 16) return 0;
    return align_up(refine_fact_bytes(256 + 3 + 6 * multires + 3 + 6 * multires_views), 256);
}

LIDF_API int lidf_refine_pack_f32(const LidfDecoder* off, int32_t multires, int32_t multires_views,
                                    void* packed, size_t packed_bytes, lidf_stream_t stream) {
    if (!off) return LIDF_ERR_BAD_ARG;
    int rc;
    if ((rc = check_decoder(off))) return rc;
    const size_t need = lidf_refine_pack_bytes(multires, multires_views);
    if (!need) return LIDF_ERR_UNSUPPORTED;
    if (!packed || packed_bytes < need) return LIDF_ERR_WORKSPACE;
    const int D = 256 + 3 + 6 * multires + 3 + 6 * multires_views;
    JobScope js;
    if ((rc = refine_ief_factorised(off, D, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr,
                                    (char*)packed, (hipStream_t)stream, 1, nullptr, nullptr, nullptr, nullptr,
                                    3 + 6 * multires_views, nullptr, false)))
        return rc;
    CHECK_HIP(flush_jobs(js.jobs, (hipStream_t)stream));
    return LIDF_OK;
}

LIDF_API int lidf_refine_pack_guarded_f32(const LidfDecoder* off, int32_t multires,
                                            int32_t multires_views, void* packed, size_t packed_bytes,
                                            void* guard, lidf_stream_t stream) {
    if (!off || !guard) return LIDF_ERR_BAD_ARG;
    int rc;
    if ((rc = check_decoder(off))) return rc;
    const size_t need = lidf_refine_pack_bytes(multires, multires_views);
    if (!need) return LIDF_ERR_UNSUPPORTED;
    if (!packed || packed_bytes < need) return LIDF_ERR_WORKSPACE;
    const int D = 256 + 3 + 6 * multires + 3 + 6 * multires_views;
    const float* ptrs[LIDF_FP_MAX_SEGS];
    long long cnt[LIDF_FP_MAX_SEGS];
    const int k = decoder_segs(off, D + (off->is_ief ? 16 : 0), ptrs, cnt, 0);
    const unsigned long long salt = refine_salt(off, multires, multires_views);
    CHECK_HIP(lidf_launch_fingerprint(ptrs, cnt, k, salt, (LidfPackGuardState*)guard,
                                      (hipStream_t)stream));
    GuardScope scope((const LidfPackGuardState*)guard);
    if ((rc = lidf_refine_pack_f32(off, multires, multires_views, packed, packed_bytes, stream)))
        return guard_fail(guard, sizeof(LidfPackGuardState), (hipStream_t)stream, rc);
    return LIDF_OK;
}

struct TrainWs {
    size_t stream, dz1, dz2, dz3, S, goff, wg, chain, small, total;
};
static TrainWs train_ws(int64_t n, int d) {
    TrainWs w;
    const size_t N = (size_t)(n > 0 ? n : 1);
    size_t o = 0;
    const int kmax = d > 256 ? d : 256;
    w.stream = o; o += linex_stream_bytes(kmax);
    w.dz1 = o;    o += align_up(N * LIDF_H1 * 4, 256);
    w.dz2 = o;    o += align_up(N * LIDF_H2 * 4, 256);
    w.dz3 = o;    o += align_up(N * LIDF_H3 * 4, 256);
    w.S = o;      o += align_up(N * LIDF_H1 * 4, 256);   // running sum of dZ1 over the IEF's passes
    w.goff = o;   o += align_up(N * 4, 256);
    w.wg = o;     o += align_up(WG_SCRATCH_FLOATS * 4, 256);
    w.chain = o;  o += chain_stream_bytes(d);
    w.small = o;  o += 512 * 4;   // B sums of the passes (the IEF's first pass, lidf_ief_finish_kernel)
    w.total = o;
    return w;
}
LIDF_API size_t lidf_decoder_train_workspace_bytes(int64_t n, int32_t d) { return train_ws(n, d).total; }

static inline const unsigned* act_m1(const float* pass, int64_t n) { return (const unsigned*)(pass + (size_t)n * LIDF_ACT_M1); }
static inline const unsigned* act_m2(const float* pass, int64_t n) { return (const unsigned*)(pass + (size_t)n * LIDF_ACT_M2); }
static inline const float* act_h1(const float* act, int64_t n, int k) { return act + (size_t)k * n * ACT_ROW_FLOATS; }

LIDF_API int lidf_decoder_forward_train_f32(const float* inp, int64_t n, int32_t d, int64_t ld_inp,
                                              const LidfDecoder* dec, float* out, float* act,
                                              void* workspace, size_t workspace_bytes,
                                              lidf_stream_t stream) {
    if (n < 0 || d <= 0 || ld_inp < d || !dec) return LIDF_ERR_BAD_ARG;
    int rc;
    if ((rc = check_decoder(dec))) return rc;
    if (n == 0) return LIDF_OK;
    if (!inp || !out || !act) return LIDF_ERR_BAD_ARG;
    const TrainWs w = train_ws(n, d);
    if (!workspace || workspace_bytes < w.total) return LIDF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* sbuf = (float*)((char*)workspace + w.stream);
    int cus;
    if ((rc = cu_count(&cus))) return rc;
    const int npass = dec->is_ief ? dec->n_iter : 1;
    const int ld1 = d + (dec->is_ief ? 16 : 0);
    float* pre = act + (size_t)npass * n * ACT_ROW_FLOATS;  // final pre-activation
    (void)ld1; (void)sbuf;
    // [inp | enc(off)] W1^T + b1 = inp W1x^T + (b1 + c) + u * off, then the chain, in registers
    if ((rc = run_chain_train(dec, d, rows_map(d, 0, 0, 0, 1), inp, ld_inp, n, nullptr, nullptr, nullptr,
                              nullptr, act, pre, out, (char*)workspace + w.chain, cus, st)))
        return rc;
    return LIDF_OK;
}

// The backward of one decoder up to (and without) the input gradient: every parameter gradient, and S = the sum
// of dZ1 over the passes left in the workspace (*S_out) for the caller's input-gradient product.
static int decoder_backward_core(const float* inp, int64_t n, int32_t d, int64_t ld_inp, const LidfDecoder* dec,
                                 const float* act, const float* g_out, const LidfDecoderGrads* grads, char* ws,
                                 const TrainWs& w, int cus, hipStream_t st, float** S_out) {
    const int npass = dec->is_ief ? dec->n_iter : 1;
    const int ld1 = d + (dec->is_ief ? 16 : 0);
    float* sbuf = (float*)(ws + w.stream);
    float* dz1 = (float*)(ws + w.dz1);
    float* dz2 = (float*)(ws + w.dz2);
    float* dz3 = (float*)(ws + w.dz3);
    float* goff = (float*)(ws + w.goff);
    float* wgs = (float*)(ws + w.wg);
    const float* pre = act + (size_t)npass * n * ACT_ROW_FLOATS;
    float* S = (float*)(ws + w.S);
    float* small = (float*)(ws + w.small);
    CHECK_HIP(hipMemsetAsync(small, 0, 512 * 4, st));
    // through the output activation: goff = dL/d(off_n)
    CHECK_HIP(lidf_launch_out_act(pre, n, dec->use_sigmoid, nullptr, g_out, goff, st));
    // W3^T | W2^T of the chained input-gradient launches: one stream for every pass
    CHECK_HIP(lidf_launch_pack_dgrad(dec->w3, dec->w2, sbuf, st));
    // Everything of layer 1 except the offset encoding sees the same operand in every pass of the IEF, so
    // its weight gradient and the input gradient are ONE product each with S = the sum of dZ1 over the
    // passes (as the factorised backward of the query does, qdec_backward_impl) instead of one per pass:
    // the first pass processed writes its dZ1 into S, the middle ones add theirs in the sweep that handles
    // the offset-encoding columns, and the IEF's first pass (constant offset-in: its share of those columns
    // follows from column sums, lidf_ief_finish_kernel) is added into S by the chained launch itself.
    if (npass == 1) S = dz1;
    for (int k = npass - 1; k >= 0; --k) {
        const float* h1 = act_h1(act, n, k);
        const float* h2 = h1 + (size_t)n * LIDF_H1;
        const float* h3 = h2 + (size_t)n * LIDF_H2;
        const float* offin = h3 + (size_t)n * LIDF_H3;
        // y_k = w4 . H3 + b4 ;  dL/dy_k = dL/d(off_{k+1}) = goff
        // dZ3 = (goff (x) w4) * lrelu'(Z3), d w4, d b4: one pass over H3
        CHECK_HIP(lidf_launch_l4_backward(goff, h3, dec->w4, 0.02f, n, dz3, grads->w4, grads->b4, wgs, st));
        CHECK_HIP(lidf_launch_wgrad(dz3, LIDF_H3, LIDF_H3, h2, LIDF_H2, LIDF_H2, n, grads->w3, LIDF_H2, grads->b3, wgs, WG_SCRATCH_FLOATS, st));
        // dZ2 = (dZ3 W3) * lrelu'(Z2), dZ1 = (dZ2 W2) * lrelu'(Z1): one register-chained launch
        const bool first_pass_short = dec->is_ief && npass > 1 && k == 0;
        float* dz1k = (k == npass - 1 || first_pass_short) ? S : dz1;
        CHECK_HIP(lidf_launch_dgrad_chain(nullptr, nullptr, dz3, act_m2(h1, n), act_m1(h1, n), n, 0.02f, dz2, dz1k,
                                          first_pass_short ? 1 : 0, sbuf, cus, st));
        CHECK_HIP(lidf_launch_wgrad(dz2, LIDF_H2, LIDF_H2, h1, LIDF_H1, LIDF_H1, n, grads->w2, LIDF_H1, grads->b2, wgs, WG_SCRATCH_FLOATS, st));
        // the 16 offset-encoding columns of layer 1 (enc_k = off_k wenc^T + benc): d W1[:, d:], d wenc,
        // d benc and d off_k = d off_{k+1} + d enc . wenc in one sweep over this pass's dZ1
        if (dec->is_ief && !first_pass_short)
            CHECK_HIP(lidf_launch_ief_tail(dz1k, offin, dec->w1 + d, ld1, dec->wenc, dec->benc, n,
                                           k == npass - 1 ? 0 : 2, S, goff, grads->w1 + d, grads->wenc,
                                           grads->benc, npass > 1 ? small : nullptr, wgs, st));
    }
    // d W1[:, 0:d] = S^T inp, d b1 = column sums of S
    CHECK_HIP(lidf_launch_wgrad(S, LIDF_H1, LIDF_H1, inp, ld_inp, d, n, grads->w1, ld1, grads->b1, wgs, WG_SCRATCH_FLOATS, st));
    if (dec->is_ief && npass > 1)
        CHECK_HIP(lidf_launch_ief_first_pass(grads->b1, small, dec->init_offset, dec->w1 + d, ld1, dec->wenc,
                                             dec->benc, grads->w1 + d, grads->wenc, grads->benc, st));
    *S_out = S;
    return LIDF_OK;
}

static int check_decoder_grads(const LidfDecoder* dec, const LidfDecoderGrads* grads) {
    if (!grads->w1 || !grads->b1 || !grads->w2 || !grads->b2 || !grads->w3 || !grads->b3 ||
        !grads->w4 || !grads->b4 || (dec->is_ief && (!grads->wenc || !grads->benc)))
        return LIDF_ERR_BAD_ARG;
    return LIDF_OK;
}

// d inp = S W1[:, 0:d] (+ S2 W1_2[:, 0:d]: the pair's joint product over K = 512), in launches of at most 8 output
// tiles of 32 columns, evenly sized (385 columns: 7 + 6 tiles instead of 8 + 8)
static int decoder_input_grad(const float* S, const LidfDecoder* dec, const float* S2, const LidfDecoder* dec2,
                              int64_t n, int32_t d, float* d_inp, int64_t ld_dinp, float* sbuf, int cus,
                              hipStream_t st) {
    LinEx L = {};
    L.n = n; L.transposed = 1; L.mask_slope = 0.02f;
    const int ld1 = d + (dec->is_ief ? 16 : 0);
    // 32 t + 1 columns (385 = 12 x 32 + 1) over more than a handful of rows: the last column rides through the vector
    // unit of the last launch (6 + 6 tiles instead of 7 + 6)
    // (a stream slot holds 8 quads per k-quad: a launch of 8 tiles has no room for the column's quad)
    bool xcol = d % 32 == 1 && d > 32 && n > 16 * 128;
    if (xcol) {
        const int t = (d - 1) / 32, l = (t + 7) / 8;
        if ((t + l - 1) / l >= 8) xcol = false;
    }
    // (385 columns as ONE launch of 12 tiles + the column — the operand rows read once — was measured slower: 12.03
    // against 11.41 ms per rows step, 11.1 against 10.6 with the pair node; docs/history.md section 13)
    const int dt = xcol ? d - 1 : d;
    const int tiles = (dt + 31) / 32, launches = (tiles + 7) / 8, per = (tiles + launches - 1) / launches;
    for (int c0 = 0; c0 < dt; c0 += 32 * per) {
        int cols = dt - c0 < 32 * per ? dt - c0 : 32 * per;
        L.xcol = 0;
        if (xcol && c0 + cols == dt) { cols += 1; L.xcol = 1; }
        L.w = dec->w1 + c0; L.ldw = ld1; L.nout = cols; L.k = LIDF_H1; L.X = S; L.ldx = LIDF_H1;
        if (S2) {
            L.X2 = S2; L.ldx2 = LIDF_H1; L.k2 = LIDF_H1;
            L.w_2 = dec2->w1 + c0; L.ldw_2 = d + (dec2->is_ief ? 16 : 0);
        }
        L.out = d_inp + c0; L.ld_out = ld_dinp; L.accumulate = 0;
        int rc;
        if ((rc = run_linex(L, sbuf, cus, st))) return rc;
    }
    return LIDF_OK;
}

LIDF_API int lidf_decoder_backward_f32(const float* inp, int64_t n, int32_t d, int64_t ld_inp,
                                         const LidfDecoder* dec, const float* act,
                                         const float* g_out, float* d_inp, int64_t ld_dinp,
                                         const LidfDecoderGrads* grads, void* workspace,
                                         size_t workspace_bytes, lidf_stream_t stream) {
    if (n < 0 || d <= 0 || ld_inp < d || !dec || !grads) return LIDF_ERR_BAD_ARG;
    int rc;
    if ((rc = check_decoder(dec))) return rc;
    if ((rc = check_decoder_grads(dec, grads))) return rc;
    if (d_inp && ld_dinp < d) return LIDF_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    // gradients start at zero (also what an empty batch returns)
    CHECK_HIP(zero_decoder_grads(grads, d + (dec->is_ief ? 16 : 0), dec->is_ief, st));
    if (n == 0) return LIDF_OK;
    if (!inp || !act || !g_out) return LIDF_ERR_BAD_ARG;
    const TrainWs w = train_ws(n, d);
    if (!workspace || workspace_bytes < w.total) return LIDF_ERR_WORKSPACE;
    char* ws = (char*)workspace;
    int cus;
    if ((rc = cu_count(&cus))) return rc;
    float* S = nullptr;
    if ((rc = decoder_backward_core(inp, n, d, ld_inp, dec, act, g_out, grads, ws, w, cus, st, &S))) return rc;
    if (d_inp) return decoder_input_grad(S, dec, nullptr, nullptr, n, d, d_inp, ld_dinp, (float*)(ws + w.stream), cus, st);
    return LIDF_OK;
}

// ---- both decoders on the same rows (models/pipeline.py:434-435: prob_dec(inp), offset_dec(inp)) as ONE backward:
// the two input gradients are one K = 512 product over [S_prob | S_off] and the rows' gradient is stored once —
// not two K = 256 products, two stores of [n, d] and the accumulation autograd runs when the modules are two nodes.
// Workspace: [decoder workspace of prob | of off | stream of the K = 512 product].
LIDF_API size_t lidf_decoder_pair_workspace_bytes(int64_t n, int32_t d) {
    return 2 * align_up(train_ws(n, d).total, 256) + linex_stream_bytes(2 * LIDF_H1);
}
LIDF_API size_t lidf_decoder_pair_workspace_offset(int64_t n, int32_t d, int32_t which) {
    return (size_t)(which ? 1 : 0) * align_up(train_ws(n, d).total, 256);
}

LIDF_API int lidf_decoder_pair_backward_f32(const float* inp, int64_t n, int32_t d, int64_t ld_inp,
                                              const LidfDecoder* prob, const LidfDecoder* off,
                                              const float* act_prob, const float* act_off,
                                              const float* g_prob, const float* g_off, float* d_inp,
                                              int64_t ld_dinp, const LidfDecoderGrads* grads_prob,
                                              const LidfDecoderGrads* grads_off, void* workspace,
                                              size_t workspace_bytes, lidf_stream_t stream) {
    if (n < 0 || d <= 0 || ld_inp < d || !prob || !off || !grads_prob || !grads_off) return LIDF_ERR_BAD_ARG;
    int rc;
    if ((rc = check_decoder(prob)) || (rc = check_decoder(off))) return rc;
    if ((rc = check_decoder_grads(prob, grads_prob)) || (rc = check_decoder_grads(off, grads_off))) return rc;
    if (d_inp && ld_dinp < d) return LIDF_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    CHECK_HIP(zero_decoder_grads(grads_prob, d + (prob->is_ief ? 16 : 0), prob->is_ief, st));
    CHECK_HIP(zero_decoder_grads(grads_off, d + (off->is_ief ? 16 : 0), off->is_ief, st));
    if (n == 0) return LIDF_OK;
    if (!inp || !act_prob || !act_off || !g_prob || !g_off) return LIDF_ERR_BAD_ARG;
    const TrainWs w = train_ws(n, d);
    const size_t one = align_up(w.total, 256);
    if (!workspace || workspace_bytes < lidf_decoder_pair_workspace_bytes(n, d)) return LIDF_ERR_WORKSPACE;
    char* ws = (char*)workspace;
    int cus;
    if ((rc = cu_count(&cus))) return rc;
    float *Sp = nullptr, *So = nullptr;
    if ((rc = decoder_backward_core(inp, n, d, ld_inp, prob, act_prob, g_prob, grads_prob, ws, w, cus, st, &Sp))) return rc;
    if ((rc = decoder_backward_core(inp, n, d, ld_inp, off, act_off, g_off, grads_off, ws + one, w, cus, st, &So))) return rc;
    if (d_inp) return decoder_input_grad(Sp, prob, So, off, n, d, d_inp, ld_dinp, (float*)(ws + 2 * one), cus, st);
    return LIDF_OK;
}

// ---- query, training path: decoder input rows and their gradient ------------------------------------
LIDF_API int lidf_build_rows_f32(const int32_t* pair_ray, const int32_t* pair_vox,
                                   const float* pair_t, const float* ray_dir,
                                   const float* vox_center, int32_t pos_rel, const float* vox_feat,
                                   const float* rayfeat, int32_t multires, int32_t multires_views,
                                   int64_t n_pairs, float* rows, lidf_stream_t stream) {
    if (n_pairs < 0 || multires < 0 || multires > 16 || multires_views < 0 || multires_views > 16)
        return LIDF_ERR_BAD_ARG;
    if (n_pairs == 0) return LIDF_OK;
    if (!pair_ray || !pair_vox || !pair_t || !ray_dir || !vox_feat || !rayfeat || !rows)
        return LIDF_ERR_BAD_ARG;
    if (pos_rel && !vox_center) return LIDF_ERR_BAD_ARG;
    const int E = 3 + 6 * multires, Ed = 3 + 6 * multires_views;
    CHECK_HIP(lidf_launch_build_rows(pair_ray, pair_vox, pair_t, ray_dir, vox_center, pos_rel,
                                     vox_feat, rayfeat, 128 + Ed, multires, Ed, n_pairs, rows,
                                     256 + 2 * E + Ed, (hipStream_t)stream));
    return LIDF_OK;
}

LIDF_API int lidf_rows_backward_f32(const float* d_rows, const int32_t* pair_off,
                                      const int32_t* pair_vox, int64_t n_rays, int64_t n_pairs,
                                      int64_t n_vox, int32_t multires, int32_t multires_views,
                                      float* d_vox_feat, float* d_rayfeat, lidf_stream_t stream) {
    if (n_pairs < 0 || n_rays < 0 || n_vox < 0 || multires < 0 || multires > 16 ||
        multires_views < 0 || multires_views > 16)
        return LIDF_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int E = 3 + 6 * multires, Ed = 3 + 6 * multires_views;
    if (d_vox_feat && n_vox > 0) CHECK_HIP(hipMemsetAsync(d_vox_feat, 0, (size_t)n_vox * 128 * 4, st));
    if (n_rays == 0) return LIDF_OK;
    if (!pair_off || (n_pairs > 0 && (!d_rows || !pair_vox))) return LIDF_ERR_BAD_ARG;
    CHECK_HIP(lidf_launch_rows_backward(d_rows, 256 + 2 * E + Ed, 2 * E, pair_off, pair_vox, n_rays,
                                        n_pairs, Ed, d_vox_feat, d_rayfeat, 128 + Ed, st));
    return LIDF_OK;
}

LIDF_API int lidf_ray_features_backward_f32(const float* d_rayfeat, const int32_t* ray_pix,
                                              const int32_t* ray_bid, int64_t n_rays, int32_t batch,
                                              int32_t height, int32_t width, int32_t roi_inp_bbox,
                                              int32_t multires_views, float* d_feat_grid,
                                              void* workspace, size_t workspace_bytes,
                                              lidf_stream_t stream) {
    if (n_rays < 0 || batch <= 0 || height <= 0 || width <= 0 || !d_feat_grid) return LIDF_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    CHECK_HIP(hipMemsetAsync(d_feat_grid, 0, (size_t)batch * 32 * height * width * 4, st));
    if (n_rays == 0) return LIDF_OK;
    if (!d_rayfeat || !ray_pix || !ray_bid) return LIDF_ERR_BAD_ARG;
    // with room for a [B,128,h,w] scratch image the unclamped boxes take the atomic-free gather path
    // (+ B*h*w ints: pixels named by a single ray are stored instead of added atomically, and the
    // rays with a clamped box are parked as well and accumulated tile by tile through LDS)
    const size_t need = (size_t)batch * 128 * height * width * 4;
    const size_t need2 = need + (size_t)batch * height * width * 4;
    float* gimg = (workspace && workspace_bytes >= need) ? (float*)workspace : nullptr;
    int* pix_rays = (workspace && workspace_bytes >= need2) ? (int*)((char*)workspace + need) : nullptr;
    CHECK_HIP(lidf_launch_rayfeat_backward(d_rayfeat, 128 + 3 + 6 * multires_views, ray_pix, ray_bid,
                                           n_rays, roi_inp_bbox / 2, batch, height, width, d_feat_grid,
                                           gimg, pix_rays, st));
    return LIDF_OK;
}

// ---- query decoders, factorised training path -------------------------------------------------------
LIDF_API int lidf_pe_rows_f32(const int32_t* pair_ray, const int32_t* pair_vox, const float* pair_t,
                                const float* ray_dir, const float* vox_center, int32_t pos_rel,
                                int32_t multires, int64_t n_pairs, float* pe, lidf_stream_t stream) {
    if (n_pairs < 0 || multires < 0 || multires > 16) return LIDF_ERR_BAD_ARG;
    if (n_pairs == 0) return LIDF_OK;
    if (!pair_ray || !pair_vox || !pair_t || !ray_dir || !pe || (pos_rel && !vox_center))
        return LIDF_ERR_BAD_ARG;
    if (n_pairs * 2 * (3 + 6 * multires) > 0x7fffffffLL * 256) return LIDF_ERR_UNSUPPORTED;
    CHECK_HIP(lidf_launch_pe_rows(pair_ray, pair_vox, pair_t, ray_dir, vox_center, pos_rel, multires,
                                  n_pairs, pe, (hipStream_t)stream));
    return LIDF_OK;
}

static inline size_t qact_pass(int64_t P) { return (size_t)P * ACT_ROW_FLOATS; }

LIDF_API size_t lidf_query_decoder_act_floats(int64_t n_pairs, int64_t n_rays, int64_t n_vox,
                                                int32_t n_pass) {
    if (n_pairs < 0 || n_rays < 0 || n_vox < 0 || n_pass <= 0) return 0;
    return (size_t)(n_vox + n_rays) * LIDF_H1 + (size_t)n_pass * qact_pass(n_pairs) + (size_t)n_pairs;
}

struct QTrainWs {
    size_t stream, dz1, dz2, dz3, S, goff, dvox, dray, wg, seg, seg_bytes, chain, small, total;
};
static QTrainWs qtrain_ws(int64_t P, int64_t R, int64_t V) {
    QTrainWs w;
    const size_t N = (size_t)(P > 0 ? P : 1);
    size_t o = 0;
    w.stream = o; o += linex_stream_bytes(256);
    w.dz1 = o;    o += align_up(N * LIDF_H1 * 4, 256);
    w.dz2 = o;    o += align_up(N * LIDF_H2 * 4, 256);
    w.dz3 = o;    o += align_up(N * LIDF_H3 * 4, 256);
    w.S = o;      o += align_up(N * LIDF_H1 * 4, 256);
    w.goff = o;   o += align_up(N * 4, 256);
    w.dvox = o;   o += align_up((size_t)(V > 0 ? V : 1) * LIDF_H1 * 4, 256);
    w.dray = o;   o += align_up((size_t)(R > 0 ? R : 1) * LIDF_H1 * 4, 256);
    w.wg = o;     o += align_up(WG_SCRATCH_FLOATS * 4, 256);
    w.seg_bytes = lidf_seg_sum_idx_ws_bytes(P, V);
    w.seg = o;    o += align_up(w.seg_bytes, 256);
    w.chain = o;  o += chain_stream_bytes(2 * (3 + 6 * 16));
    w.small = o;  o += 512 * 4;   // B sums of the passes | column sums of the running sum
    w.total = o;
    return w;
}
LIDF_API size_t lidf_query_decoder_workspace_bytes(int64_t n_pairs, int64_t n_rays, int64_t n_vox) {
    return qtrain_ws(n_pairs, n_rays, n_vox).total;
}

static int check_qtrain(const LidfQueryTrainArgs* q) {
    if (!q || !q->dec) return LIDF_ERR_BAD_ARG;
    if (q->n_pairs < 0 || q->n_rays < 0 || q->n_vox < 0) return LIDF_ERR_BAD_ARG;
    if (q->multires < 0 || q->multires > 16 || q->multires_views < 0 || q->multires_views > 16)
        return LIDF_ERR_BAD_ARG;
    if (q->n_pairs > 0 && (!q->pair_off || !q->pair_ray || !q->pair_vox || !q->pe || !q->vox_feat ||
                           !q->rayfeat || q->n_rays == 0 || q->n_vox == 0))
        return LIDF_ERR_BAD_ARG;
    return check_decoder(q->dec);
}

// The factorised decoder forward with kept activations. E2 = width of the per-pair operand `pe` (the query:
// embed(enter) | embed(leave); stage 2: embed(pos)); W1's columns are [voxel feature 128 | ROI 128 | pe E2 |
// embed(dir) Ed | (IEF) offset encoding 16]. Three weight streams: the per-voxel layer-1 launch (s_vox), the
// per-ray one (s_ray) and the chain (s_chain): pack_mode 0 = pack and run, 1 = pack only, 2 = run on streams
// packed earlier. parts: 1 = voxpart, 2 = raypart, 4 = chain (which of the three launches to run).
static int qdec_forward_impl(const LidfQueryTrainArgs* q, int E2, float* out, float* voxpart, float* raypart,
                             float* passes, float* pre, float* s_vox, float* s_ray, char* s_chain, int cus,
                             hipStream_t st, int pack_mode = 0, int parts = 7, int pe_ld = 0) {
    int rc;
    const int64_t P = q->n_pairs, R = q->n_rays, V = q->n_vox;
    const LidfDecoder* dec = q->dec;
    const int Ed = 3 + 6 * q->multires_views;
    const int D = 256 + E2 + Ed, ld1 = D + (dec->is_ief ? 16 : 0);
    LinEx L = {};
    // voxpart[v] = W1[:, 0:128] vox_feat[v] + b1 (+ c) ; raypart[r] = W1[:, rgb | dir] rayfeat[r]
    if (parts & 1) {
        L.w = dec->w1; L.b = dec->b1; L.ldw = ld1; L.nout = LIDF_H1; L.k = 128; L.dcore = D;
        L.ief = dec->is_ief ? dec : nullptr;   // bias += c (no offset operand: xoff stays NULL)
        L.X = q->vox_feat; L.ldx = 128; L.n = V; L.out = voxpart; L.ld_out = LIDF_H1;
        if ((rc = run_linex(L, s_vox, cus, st, pack_mode))) return rc;
    }
    if (parts & 2) {
        L = {};
        L.w = dec->w1; L.ldw = ld1; L.nout = LIDF_H1; L.k = 128; L.c0 = 128; L.k1 = Ed; L.c1 = 256 + E2;
        L.X = q->rayfeat; L.ldx = 128 + Ed; L.n = R; L.out = raypart; L.ld_out = LIDF_H1;
        if ((rc = run_linex(L, s_ray, cus, st, pack_mode))) return rc;
    }
    // layer 1 = W1[:, pe columns] PE + voxpart[voxel] + raypart[ray] (+ u * off), then the chain,
    // in registers; every pass's H1 | H2 | H3 | offset-in is kept
    if (parts & 4) {
        // (pe_ld: row stride of `pe` when its rows are a column window of wider rows)
        if ((rc = run_chain_train(dec, D, rows_map(E2, 256, 0, 0, 0), q->pe, pe_ld ? pe_ld : E2, P, q->pair_vox,
                                  q->pair_ray, voxpart, raypart, passes, pre, out, s_chain, cus, st, LIDF_MODE_TRAIN,
                                  pack_mode)))
            return rc;
    }
    return LIDF_OK;
}

LIDF_API int lidf_query_decoder_forward_train_f32(const LidfQueryTrainArgs* q, float* out, float* act,
                                                    void* workspace, size_t workspace_bytes,
                                                    lidf_stream_t stream) {
    int rc;
    if ((rc = check_qtrain(q))) return rc;
    const int64_t P = q->n_pairs, R = q->n_rays, V = q->n_vox;
    if (P == 0) return LIDF_OK;
    if (!out || !act) return LIDF_ERR_BAD_ARG;
    const QTrainWs w = qtrain_ws(P, R, V);
    if (!workspace || workspace_bytes < w.total) return LIDF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* sbuf = (float*)((char*)workspace + w.stream);
    int cus;
    if ((rc = cu_count(&cus))) return rc;
    const LidfDecoder* dec = q->dec;
    const int E2 = 2 * (3 + 6 * q->multires);
    const int npass = dec->is_ief ? dec->n_iter : 1;
    float* voxpart = act;
    float* raypart = voxpart + (size_t)V * LIDF_H1;
    float* passes = raypart + (size_t)R * LIDF_H1;
    float* pre = passes + (size_t)npass * qact_pass(P);
    // (one stream slot serves the two layer-1 launches in turn: each is consumed before the next pack)
    return qdec_forward_impl(q, E2, out, voxpart, raypart, passes, pre, sbuf, sbuf, (char*)workspace + w.chain,
                             cus, st);
}

LIDF_API size_t lidf_query_forward_train_workspace_bytes(int64_t n_rays, int64_t n_vox) {
    return lidf_query_workspace_bytes(n_rays, n_vox, 0);
}

LIDF_API int lidf_query_forward_train_f32(const LidfQueryTrainArgs* q, const LidfDecoder* offset_dec,
                                            const float* pair_t, const float* ray_dir,
                                            const float* vox_center, int32_t pos_rel, float* out_prob,
                                            float* out_off, float* act_prob, float* act_off,
                                            void* workspace, size_t workspace_bytes,
                                            lidf_stream_t stream) {
    int rc;
    if ((rc = check_qtrain(q))) return rc;
    if ((rc = check_decoder(offset_dec))) return rc;
    const int64_t P = q->n_pairs, R = q->n_rays, V = q->n_vox;
    if (P > 0x7fffffffLL || R > 0x15555555LL) return LIDF_ERR_UNSUPPORTED;
    const int L = q->multires, Lv = q->multires_views;
    if ((rc = check_query_model(q->dec, offset_dec, L, Lv, LIDF_PRECISION_F32))) return rc;
    if (P == 0) return LIDF_OK;
    if (!pair_t || !ray_dir || !out_prob || !out_off || !act_prob || !act_off || (pos_rel && !vox_center))
        return LIDF_ERR_BAD_ARG;
    const QueryWs w = query_ws(R, V, L, Lv);
    if (!workspace || workspace_bytes < w.total) return LIDF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    if ((rc = pack_query_weights(q->dec, offset_dec, L, Lv, LIDF_PRECISION_F32, ws, st))) return rc;
    const float* aux_pts = (const float*)(ws + w.aux_pts);
    float* voxpart = (float*)(ws + w.voxpart);
    float* raypart = (float*)(ws + w.raypart);
    int cus;
    if ((rc = cu_count(&cus))) return rc;
    const int E = 3 + 6 * L, Ed = 3 + 6 * Lv;
    L1Map mf = {};
    mf.L = L;
    const StreamLayout lf = lidf_make_layout(2, LIDF_MODE_FUSED, mf);
    const L1Map mv = rows_map(128, 0, 0, 0, 1);
    const StreamLayout lv = lidf_make_layout(2, LIDF_MODE_L1ONLY, mv);
    const L1Map mr = rows_map(128, 128, Ed, 256 + 2 * E, 0);
    const StreamLayout lr = lidf_make_layout(2, LIDF_MODE_L1ONLY, mr);
    {   // per-voxel and per-ray parts of layer 1, both nets: [V,512], [R,512]
        PointsArgs av = {};
        av.stream = (const float*)(ws + w.stream_vox); av.aux = aux_pts;
        av.nets = 2; av.l1_quads = lv.l1_quads; av.net_quads = lv.net_quads;
        av.n = V; av.X = q->vox_feat; av.ldx = 128;
        av.D = mv.D; av.KQ1 = mv.KQ1; av.has_bias = 1; av.out_base = voxpart;
        PointsArgs a = {};
        a.stream = (const float*)(ws + w.stream_ray); a.aux = aux_pts;
        a.nets = 2; a.l1_quads = lr.l1_quads; a.net_quads = lr.net_quads;
        a.n = R; a.X = q->rayfeat; a.ldx = 128 + Ed;
        a.D = mr.D; a.KQ1 = mr.KQ1; a.has_bias = 0; a.out_base = raypart;
        CHECK_HIP(lidf_launch_l1only_pair(a, av, nullptr, cus, st));
    }
    PointsArgs a = {};
    a.stream = (const float*)(ws + w.stream_pts); a.aux = aux_pts;
    a.nets = 2; a.l1_quads = lf.l1_quads; a.net_quads = lf.net_quads;
    a.n = P;
    fill_net_args(a, 0, q->dec, out_prob, 0);
    fill_net_args(a, 1, offset_dec, out_off, 1);
    a.pair_ray = q->pair_ray; a.pair_vox = q->pair_vox; a.pair_t = pair_t;
    a.ray_dir = ray_dir; a.voxpart = voxpart; a.raypart = raypart;
    a.vox_center = vox_center; a.pos_rel = pos_rel; a.L = L;
    a.pair_pred_pos = nullptr;   // the differentiable tail (lidf_query_tail_f32) forms it
    const LidfDecoder* decs[2] = {q->dec, offset_dec};
    float* acts[2] = {act_prob, act_off};
    for (int i = 0; i < 2; ++i) {
        const int np = decs[i]->is_ief ? decs[i]->n_iter : 1;
        // act = [voxpart V x 256 | raypart R x 256 (not filled here, not read by the backward) | passes | pre]
        a.tr_passes[i] = acts[i] + (size_t)(V + R) * LIDF_H1;
        a.tr_pre[i] = a.tr_passes[i] + (size_t)np * qact_pass(P);
    }
    a.tr_pass_floats = (long long)qact_pass(P);
    const long long nt = (P + 127) / 128;
    CHECK_HIP(lidf_launch_points(LIDF_MODE_FUSED, a, (int)(nt < cus ? nt : cus), st));
    return LIDF_OK;
}

// The training forward with offset_dec on the selected pair of every ray only (the training counterpart of
// LidfQueryArgs.offsets_selected): prob_dec on every pair with its activations kept, the per-ray softmax / arg-max,
// offset_dec on the one-pair-per-ray list with ITS activations kept (R rows), positions.
extern "C" hipError_t lidf_launch_sel_from_ids(const long long*, const int*, const float*, long long, long long, int*,
                                               int*, float*, hipStream_t);
LIDF_API int lidf_query_forward_train_selected_f32(const LidfQueryTrainArgs* q, const LidfDecoder* offset_dec,
                                                     const float* pair_t, const float* ray_dir,
                                                     const float* vox_center, int32_t pos_rel, float offset_range0,
                                                     float offset_range1, float part_size,
                                                     const int64_t* max_pair_id_in, float* out_prob, float* softmax,
                                                     int64_t* max_pair_id, float* pred_offset, float* pair_pred_pos,
                                                     float* pred_pos, float* act_prob, float* act_off_rows,
                                                     void* workspace, size_t workspace_bytes, lidf_stream_t stream) {
    int rc;
    if ((rc = check_qtrain(q))) return rc;
    if ((rc = check_decoder(offset_dec))) return rc;
    const int64_t P = q->n_pairs, R = q->n_rays, V = q->n_vox;
    if (P > 0x7fffffffLL || R > 0x15555555LL) return LIDF_ERR_UNSUPPORTED;
    const int L = q->multires, Lv = q->multires_views;
    if ((rc = check_query_model(q->dec, offset_dec, L, Lv, LIDF_PRECISION_F32))) return rc;
    if (R == 0) return LIDF_OK;
    if (!max_pair_id || !pred_pos) return LIDF_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (P == 0) {   // every ray selects the dummy row (pipeline.py:452-454)
        CHECK_HIP(lidf_launch_ray_reduce_dev(nullptr, nullptr, q->pair_off, R, 0, nullptr, nullptr, nullptr, nullptr, 0,
                                             nullptr, (long long*)max_pair_id, pred_pos, nullptr, st, nullptr, nullptr,
                                             nullptr, nullptr, nullptr));
        return LIDF_OK;
    }
    if (!pair_t || !ray_dir || !out_prob || !softmax || !pred_offset || !pair_pred_pos || !act_prob || !act_off_rows ||
        (pos_rel && !vox_center))
        return LIDF_ERR_BAD_ARG;
    const QueryWs w = query_ws(R, V, L, Lv);
    if (!workspace || workspace_bytes < w.total) return LIDF_ERR_WORKSPACE;
    char* ws = (char*)workspace;
    if ((rc = pack_query_weights(q->dec, offset_dec, L, Lv, LIDF_PRECISION_F32, ws, st))) return rc;
    const float* aux_pts = (const float*)(ws + w.aux_pts);
    const float* stream_pts = (const float*)(ws + w.stream_pts);
    float* voxpart = (float*)(ws + w.voxpart);
    float* raypart = (float*)(ws + w.raypart);
    int cus;
    if ((rc = cu_count(&cus))) return rc;
    const int E = 3 + 6 * L, Ed = 3 + 6 * Lv;
    L1Map mf = {};
    mf.L = L;
    const StreamLayout lf = lidf_make_layout(2, LIDF_MODE_FUSED, mf);
    const L1Map mv = rows_map(128, 0, 0, 0, 1);
    const StreamLayout lv = lidf_make_layout(2, LIDF_MODE_L1ONLY, mv);
    const L1Map mr = rows_map(128, 128, Ed, 256 + 2 * E, 0);
    const StreamLayout lr = lidf_make_layout(2, LIDF_MODE_L1ONLY, mr);
    {   // per-voxel and per-ray parts of layer 1, both nets: [V,512], [R,512]
        PointsArgs av = {};
        av.stream = (const float*)(ws + w.stream_vox); av.aux = aux_pts;
        av.nets = 2; av.l1_quads = lv.l1_quads; av.net_quads = lv.net_quads;
        av.n = V; av.X = q->vox_feat; av.ldx = 128;
        av.D = mv.D; av.KQ1 = mv.KQ1; av.has_bias = 1; av.out_base = voxpart;
        PointsArgs a = {};
        a.stream = (const float*)(ws + w.stream_ray); a.aux = aux_pts;
        a.nets = 2; a.l1_quads = lr.l1_quads; a.net_quads = lr.net_quads;
        a.n = R; a.X = q->rayfeat; a.ldx = 128 + Ed;
        a.D = mr.D; a.KQ1 = mr.KQ1; a.has_bias = 0; a.out_base = raypart;
        CHECK_HIP(lidf_launch_l1only_pair(a, av, nullptr, cus, st));
    }
    PointsArgs a = {};
    a.stream = stream_pts; a.aux = aux_pts;
    a.nets = 1; a.l1_quads = lf.l1_quads; a.net_quads = lf.net_quads;
    a.part_ld = 512; a.part_off = 0;
    a.n = P;
    fill_net_args(a, 0, q->dec, out_prob, 0);
    a.pair_ray = q->pair_ray; a.pair_vox = q->pair_vox; a.pair_t = pair_t;
    a.ray_dir = ray_dir; a.voxpart = voxpart; a.raypart = raypart;
    a.vox_center = vox_center; a.pos_rel = pos_rel; a.L = L;
    a.r0 = offset_range0;
    a.rscale = offset_range1 - offset_range0;
    a.sqrt3 = (float)1.7320508075688772;
    a.part_size = part_size;
    a.tr_passes[0] = act_prob + (size_t)(V + R) * LIDF_H1;
    a.tr_pre[0] = a.tr_passes[0] + qact_pass(P);
    a.tr_pass_floats = (long long)qact_pass(P);
    const long long nt = (P + 127) / 128;
    CHECK_HIP(lidf_launch_points(LIDF_MODE_FUSED, a, (int)(nt < cus ? nt : cus), st));
    char* sb = ws + w.sel;
    const size_t Rc = (size_t)R;
    int* sel_ray = (int*)sb;
    int* sel_vox = (int*)(sb + Rc * 4);
    float* sel_t = (float*)(sb + Rc * 8);
    float* off_sel = (float*)(sb + Rc * 16);
    float* pos_sel = (float*)(sb + Rc * 20);
    const bool given = max_pair_id_in != nullptr;
    CHECK_HIP(lidf_launch_ray_reduce_dev(out_prob, nullptr, q->pair_off, R, P, nullptr, nullptr, nullptr, nullptr, 0,
                                         softmax, (long long*)max_pair_id, nullptr, nullptr, st, q->pair_vox, pair_t,
                                         given ? nullptr : sel_ray, sel_vox, sel_t));
    if (given)
        CHECK_HIP(lidf_launch_sel_from_ids((const long long*)max_pair_id_in, q->pair_vox, pair_t, R, P, sel_ray, sel_vox,
                                           sel_t, st));
    const int npo = offset_dec->is_ief ? offset_dec->n_iter : 1;
    PointsArgs ao = a;
    ao.stream = stream_pts + (size_t)lf.net_quads * 256; ao.aux = aux_pts + LIDF_AUX_FLOATS;
    ao.part_off = 256;
    ao.n = R;
    fill_net_args(ao, 0, offset_dec, off_sel, 1);
    ao.pair_ray = sel_ray; ao.pair_vox = sel_vox; ao.pair_t = sel_t;
    ao.pair_pred_pos = pos_sel;
    ao.tr_passes[0] = act_off_rows + (size_t)(V + R) * LIDF_H1;
    ao.tr_pre[0] = ao.tr_passes[0] + (size_t)npo * qact_pass(R);
    ao.tr_pass_floats = (long long)qact_pass(R);
    const long long ntr = (R + 127) / 128;
    CHECK_HIP(lidf_launch_points(LIDF_MODE_FUSED, ao, (int)(ntr < cus ? ntr : cus), st));
    CHECK_HIP(lidf_launch_selected_finish((const long long*)(given ? max_pair_id_in : max_pair_id), off_sel, pos_sel, R, P,
                                          nullptr, nullptr, nullptr, nullptr, 0, pred_offset, pair_pred_pos, pred_pos,
                                          nullptr, st));
    return LIDF_OK;
}

// The factorised decoder backward (see qdec_forward_impl for E2 and the column layout of W1).
struct QdecBwd {
    int E2;
    bool zero_grads;       // false: the parameter gradients are accumulated into (a later iteration of stage 2)
    float* S_keep;         // != NULL: the running sum S of dZ1 over the passes lands here and everything per ray
                           // (segment sums per ray, dW1[:, ROI | dir], d_rayfeat) is left to the caller
    float* d_pe;           // optional [P, E2]: dL/d pe = S W1[:, 256 : 256 + E2]
    float *s_dvox, *s_dpe; // stream slots of the transposed launches (d_vox_feat, d_pe); NULL: the workspace's
    int pack_mode;         // of those two launches: 0 pack and run, 2 packed earlier
    const float* s_dgrad;  // the chained input-gradient launch's stream packed earlier (NULL: packed here, once)
    bool goff_ready;       // the caller left dL/d(pre-activation of the last pass) in the workspace's goff (g_out unused)
    // the layer-1 weight gradient over the per-pair operand on WIDER rows than `pe`: l1rows [P, l1_cols] (row stride
    // l1_ld) meet W1's columns [l1_c0, l1_c0 + l1_cols) — stage 2 keeps [ROI | embed(pos) | embed(dir)] rows in W1's
    // own column order, so that one product covers all three and no sum of S over the iterations is needed
    const float* l1rows;
    int l1_ld, l1_cols, l1_c0;
    // per-voxel sums of S through a grouping of the pairs that exists already (stage 2: the PointNet's sort of its
    // points, whose rows row0.. are the rays' predicted points): perm / vstart / first of lidf_pointnet_train.hip,
    // n = rows of that sort, partial = its scratch. NULL: the pairs are sorted here (lidf_launch_seg_sum_idx).
    const int *ss_perm, *ss_vstart, *ss_first;
    long long ss_row0, ss_n;
    float* ss_partial;
};
static int qdec_backward_impl(const LidfQueryTrainArgs* q, const QdecBwd& o, const float* passes, const float* pre,
                              const float* g_out, float* d_vox_feat, float* d_rayfeat, int accumulate_inputs,
                              const LidfDecoderGrads* grads, char* ws, const QTrainWs& w, int cus, hipStream_t st) {
    int rc;
    const LidfDecoder* dec = q->dec;
    const int64_t P = q->n_pairs, R = q->n_rays, V = q->n_vox;
    const int E2 = o.E2, Ed = 3 + 6 * q->multires_views;
    const int D = 256 + E2 + Ed, ld1 = D + (dec->is_ief ? 16 : 0);
    const int npass = dec->is_ief ? dec->n_iter : 1;
    float* sbuf = (float*)(ws + w.stream);
    float* dz1 = (float*)(ws + w.dz1);
    float* dz2 = (float*)(ws + w.dz2);
    float* dz3 = (float*)(ws + w.dz3);
    float* S = o.S_keep ? o.S_keep : (float*)(ws + w.S);
    float* goff = (float*)(ws + w.goff);
    float* wgs = (float*)(ws + w.wg);
    float* dvox = (float*)(ws + w.dvox);
    float* dray = (float*)(ws + w.dray);
    float* small = (float*)(ws + w.small);
    CHECK_HIP(hipMemsetAsync(small, 0, 512 * 4, st));
    if (!o.goff_ready) CHECK_HIP(lidf_launch_out_act(pre, P, dec->use_sigmoid, nullptr, g_out, goff, st));
    // W3^T | W2^T of the chained input-gradient launches: the same stream for every pass
    const float* dgs = o.s_dgrad;
    if (!dgs) {
        CHECK_HIP(lidf_launch_pack_dgrad(dec->w3, dec->w2, sbuf, st));
        dgs = sbuf;
    }
    for (int k = npass - 1; k >= 0; --k) {
        const float* h1 = passes + (size_t)k * qact_pass(P);
        const float* h2 = h1 + (size_t)P * LIDF_H1;
        const float* h3 = h2 + (size_t)P * LIDF_H2;
        const float* offin = h3 + (size_t)P * LIDF_H3;
        CHECK_HIP(lidf_launch_l4_backward(goff, h3, dec->w4, 0.02f, P, dz3, grads->w4, grads->b4, wgs, st));
        CHECK_HIP(lidf_launch_wgrad(dz3, LIDF_H3, LIDF_H3, h2, LIDF_H2, LIDF_H2, P, grads->w3, LIDF_H2, grads->b3, wgs, WG_SCRATCH_FLOATS, st));
        // dZ2, dZ1 of this pass: one register-chained launch. S = the running sum of dZ1 over the
        // passes (everything of layer 1 except the offset encoding sees the same operand in every
        // pass): the first pass processed writes its dZ1 straight into S, the others add theirs in
        // the same sweep over dZ1 that handles the offset-encoding columns of layer 1
        // The IEF's first pass (k = 0, the last one processed) has the constant initial offset as
        // its offset-in: its share of the offset-encoding gradients follows from column sums of S
        // (lidf_ief_finish_kernel), so its dZ1 is added into S by the chained launch itself and never
        // swept again.
        if (npass == 1) {
            if (o.S_keep) dz1 = S;
            else S = dz1;
        }
        const bool first_pass_short = dec->is_ief && npass > 1 && k == 0;
        float* dz1k = (k == npass - 1 || first_pass_short) ? S : dz1;
        CHECK_HIP(lidf_launch_dgrad_chain(nullptr, nullptr, dz3, act_m2(h1, P), act_m1(h1, P), P, 0.02f, dz2, dz1k,
                                          first_pass_short ? 1 : 0, (float*)dgs, cus, st));
        CHECK_HIP(lidf_launch_wgrad(dz2, LIDF_H2, LIDF_H2, h1, LIDF_H1, LIDF_H1, P, grads->w2, LIDF_H1, grads->b2, wgs, WG_SCRATCH_FLOATS, st));
        if (!first_pass_short)
            CHECK_HIP(lidf_launch_ief_tail(dz1k, offin, dec->is_ief ? dec->w1 + D : nullptr, ld1, dec->wenc,
                                           dec->benc, P, k == npass - 1 ? 0 : 2, S, goff,
                                           grads->w1 + D, grads->wenc, grads->benc,
                                           dec->is_ief && npass > 1 ? small : nullptr, wgs, st));
    }
    // layer 1, the pass-independent operands: S = sum over passes of dZ1
    const bool short_first = dec->is_ief && npass > 1;
    // (column sums of S through the weight-gradient launch's bias path when the first pass needs them)
    if (o.l1rows)
        CHECK_HIP(lidf_launch_wgrad(S, LIDF_H1, LIDF_H1, o.l1rows, o.l1_ld, o.l1_cols, P, grads->w1 + o.l1_c0, ld1,
                                    short_first ? small + 256 : nullptr, wgs, WG_SCRATCH_FLOATS, st));
    else
        CHECK_HIP(lidf_launch_wgrad(S, LIDF_H1, LIDF_H1, q->pe, E2, E2, P, grads->w1 + 256, ld1,
                                    short_first ? small + 256 : nullptr, wgs, WG_SCRATCH_FLOATS, st));
    if (short_first)
        CHECK_HIP(lidf_launch_ief_first_pass(small + 256, small, dec->init_offset, dec->w1 + D, ld1,
                                             dec->wenc, dec->benc, grads->w1 + D, grads->wenc,
                                             grads->benc, st));
    if (o.ss_perm)
        CHECK_HIP(lidf_launch_pnet_gather_segsum(S, o.ss_perm, o.ss_row0, o.ss_vstart, o.ss_first, V, o.ss_n,
                                                 o.ss_partial, dvox, st));
    else
        CHECK_HIP(lidf_launch_seg_sum_idx(S, q->pair_vox, P, V, dvox, w.seg_bytes ? ws + w.seg : nullptr,
                                          w.seg_bytes, st));
    // voxel part: voxpart[v] = W1[:, 0:128] vox_feat[v] + b1 (+ c)
    CHECK_HIP(lidf_launch_wgrad(dvox, LIDF_H1, LIDF_H1, q->vox_feat, 128, 128, V, grads->w1, ld1, grads->b1, wgs, WG_SCRATCH_FLOATS, st));
    if (!o.S_keep) {
        CHECK_HIP(lidf_launch_seg_sum_ray(S, LIDF_H1, q->pair_off, R, dray, st));
        // ray part: raypart[r] = W1[:, 128:256] roi[r] + W1[:, 256+E2:] embed(dir)[r]
        CHECK_HIP(lidf_launch_wgrad(dray, LIDF_H1, LIDF_H1, q->rayfeat, 128 + Ed, 128, R, grads->w1 + 128, ld1, nullptr, wgs, WG_SCRATCH_FLOATS, st));
        CHECK_HIP(lidf_launch_wgrad(dray, LIDF_H1, LIDF_H1, q->rayfeat + 128, 128 + Ed, Ed, R, grads->w1 + 256 + E2, ld1, nullptr, wgs, WG_SCRATCH_FLOATS, st));
    }
    LinEx L = {};
    L.transposed = 1; L.ldw = ld1; L.k = LIDF_H1; L.ldx = LIDF_H1; L.accumulate = accumulate_inputs ? 1 : 0;
    if (d_vox_feat) {
        L.w = dec->w1; L.nout = 128; L.X = dvox; L.n = V; L.out = d_vox_feat; L.ld_out = 128;
        if ((rc = run_linex(L, o.s_dvox ? o.s_dvox : sbuf, cus, st, o.s_dvox ? o.pack_mode : 0))) return rc;
    }
    if (o.d_pe) {
        L.w = dec->w1 + 256; L.nout = E2; L.X = S; L.n = P; L.out = o.d_pe; L.ld_out = E2; L.accumulate = 0;
        if ((rc = run_linex(L, o.s_dpe ? o.s_dpe : sbuf, cus, st, o.s_dpe ? o.pack_mode : 0))) return rc;
        L.accumulate = accumulate_inputs ? 1 : 0;
    }
    if (d_rayfeat && !o.S_keep) {
        L.w = dec->w1 + 128; L.nout = 128; L.X = dray; L.n = R; L.out = d_rayfeat; L.ld_out = 128 + Ed;
        if ((rc = run_linex(L, sbuf, cus, st))) return rc;
        L.w = dec->w1 + 256 + E2; L.nout = Ed; L.out = d_rayfeat + 128;
        if ((rc = run_linex(L, sbuf, cus, st))) return rc;
    }
    return LIDF_OK;
}

LIDF_API int lidf_query_decoder_backward_f32(const LidfQueryTrainArgs* q, const float* act,
                                               const float* g_out, float* d_vox_feat,
                                               float* d_rayfeat, int32_t accumulate_inputs,
                                               const LidfDecoderGrads* grads, void* workspace,
                                               size_t workspace_bytes, lidf_stream_t stream) {
    int rc;
    if ((rc = check_qtrain(q))) return rc;
    if (!grads) return LIDF_ERR_BAD_ARG;
    const LidfDecoder* dec = q->dec;
    if (!grads->w1 || !grads->b1 || !grads->w2 || !grads->b2 || !grads->w3 || !grads->b3 ||
        !grads->w4 || !grads->b4 || (dec->is_ief && (!grads->wenc || !grads->benc)))
        return LIDF_ERR_BAD_ARG;
    const int64_t P = q->n_pairs, R = q->n_rays, V = q->n_vox;
    hipStream_t st = (hipStream_t)stream;
    const int E2 = 2 * (3 + 6 * q->multires), Ed = 3 + 6 * q->multires_views;
    const int D = 256 + E2 + Ed, ld1 = D + (dec->is_ief ? 16 : 0);
    const int npass = dec->is_ief ? dec->n_iter : 1;
    CHECK_HIP(zero_decoder_grads(grads, ld1, dec->is_ief, st));
    if (!accumulate_inputs) {
        if (d_vox_feat && V > 0) CHECK_HIP(hipMemsetAsync(d_vox_feat, 0, (size_t)V * 128 * 4, st));
        if (d_rayfeat && R > 0) CHECK_HIP(hipMemsetAsync(d_rayfeat, 0, (size_t)R * (128 + Ed) * 4, st));
    }
    if (P == 0) return LIDF_OK;
    if (!act || !g_out) return LIDF_ERR_BAD_ARG;
    const QTrainWs w = qtrain_ws(P, R, V);
    if (!workspace || workspace_bytes < w.total) return LIDF_ERR_WORKSPACE;
    int cus;
    if ((rc = cu_count(&cus))) return rc;
    const float* voxpart = act;
    const float* raypart = voxpart + (size_t)V * LIDF_H1;
    const float* passes = raypart + (size_t)R * LIDF_H1;
    const float* pre = passes + (size_t)npass * qact_pass(P);
    QdecBwd o = {};
    o.E2 = E2; o.zero_grads = true;
    return qdec_backward_impl(q, o, passes, pre, g_out, d_vox_feat, d_rayfeat, accumulate_inputs, grads,
                              (char*)workspace, w, cus, st);
}

// The decoder's backward from ONE pair per ray (the offset decoder under the reference's losses): see
// lidf_gather_sel_rows_kernel. The compacted problem — R pairs, pair r on ray r — goes through the same launches
// as the dense one.
struct QRowsWs { size_t pvox, poff, g, pe, act, inner, total; QTrainWs w; };
static QRowsWs qrows_ws(int64_t R, int64_t V, int E2, int npass) {
    QRowsWs l;
    const size_t N = (size_t)(R > 0 ? R : 1);
    size_t o = 0;
    l.pvox = o;  o += align_up(N * 4, 256);
    l.poff = o;  o += align_up((N + 1) * 4, 256);
    l.g = o;     o += align_up(N * 4, 256);
    l.pe = o;    o += align_up(N * (size_t)E2 * 4, 256);
    l.act = o;   o += align_up(((size_t)npass * qact_pass((int64_t)N) + N) * 4, 256);
    l.w = qtrain_ws(R, R, V);
    l.inner = o; o += l.w.total;
    l.total = o;
    return l;
}
LIDF_API size_t lidf_query_decoder_rows_workspace_bytes(int64_t n_rays, int64_t n_vox, int32_t multires,
                                                          int32_t n_pass) {
    if (n_rays < 0 || n_vox < 0 || multires < 0 || multires > 16 || n_pass <= 0) return 0;
    return qrows_ws(n_rays, n_vox, 2 * (3 + 6 * multires), n_pass).total;
}
extern "C" hipError_t lidf_launch_gather_sel_rows(const float*, long long, int, const long long*, long long,
                                                  const int*, const float*, int, const float*, const float*,
                                                  const float*, float, float*, int*, float*, float*, int*, hipStream_t);
LIDF_API int lidf_query_decoder_backward_rows_f32(const LidfQueryTrainArgs* q, const float* act, int32_t act_is_rows,
                                                    const int64_t* rows, const float* g_pred_pos,
                                                    const float* g_offset_rows, const float* ray_dir, float scale,
                                                    float* d_vox_feat,
                                                    float* d_rayfeat, int32_t accumulate_inputs,
                                                    const LidfDecoderGrads* grads, void* workspace,
                                                    size_t workspace_bytes, lidf_stream_t stream) {
    int rc;
    if ((rc = check_qtrain(q))) return rc;
    if (!grads) return LIDF_ERR_BAD_ARG;
    const LidfDecoder* dec = q->dec;
    if (!grads->w1 || !grads->b1 || !grads->w2 || !grads->b2 || !grads->w3 || !grads->b3 ||
        !grads->w4 || !grads->b4 || (dec->is_ief && (!grads->wenc || !grads->benc)))
        return LIDF_ERR_BAD_ARG;
    const int64_t P = q->n_pairs, R = q->n_rays, V = q->n_vox;
    hipStream_t st = (hipStream_t)stream;
    const int E2 = 2 * (3 + 6 * q->multires), Ed = 3 + 6 * q->multires_views;
    const int D = 256 + E2 + Ed, ld1 = D + (dec->is_ief ? 16 : 0);
    const int npass = dec->is_ief ? dec->n_iter : 1;
    CHECK_HIP(zero_decoder_grads(grads, ld1, dec->is_ief, st));
    if (!accumulate_inputs) {
        if (d_vox_feat && V > 0) CHECK_HIP(hipMemsetAsync(d_vox_feat, 0, (size_t)V * 128 * 4, st));
        if (d_rayfeat && R > 0) CHECK_HIP(hipMemsetAsync(d_rayfeat, 0, (size_t)R * (128 + Ed) * 4, st));
    }
    if (P == 0 || R == 0) return LIDF_OK;
    if (!act || !rows || !ray_dir) return LIDF_ERR_BAD_ARG;
    const QRowsWs l = qrows_ws(R, V, E2, npass);
    if (!workspace || workspace_bytes < l.total) return LIDF_ERR_WORKSPACE;
    int cus;
    if ((rc = cu_count(&cus))) return rc;
    char* ws = (char*)workspace;
    int* pvox = (int*)(ws + l.pvox);
    int* poff = (int*)(ws + l.poff);
    float* g = (float*)(ws + l.g);
    float* pe = (float*)(ws + l.pe);
    const float* passes_src = act + (size_t)(V + R) * LIDF_H1;   // behind voxpart | raypart (lidf_query_decoder_act_floats)
    // (act_is_rows: the activations are the list's own — lidf_query_forward_train_selected_f32 — and stay where they are)
    const float* passes = act_is_rows ? passes_src : (const float*)(ws + l.act);
    CHECK_HIP(lidf_launch_gather_sel_rows(act_is_rows ? nullptr : passes_src, P, npass, (const long long*)rows, R,
                                          q->pair_vox, q->pe, E2, g_pred_pos, g_offset_rows, ray_dir, scale,
                                          (float*)(ws + l.act), pvox, pe, g, poff, st));
    LidfQueryTrainArgs qc = *q;
    qc.n_pairs = R;
    qc.pair_off = poff; qc.pair_ray = poff; qc.pair_vox = pvox; qc.pe = pe;   // (pair r on ray r: pair_ray = 0..R-1 = poff)
    QdecBwd o = {};
    o.E2 = E2; o.zero_grads = true;
    return qdec_backward_impl(&qc, o, passes, passes + (size_t)npass * qact_pass(R), g, d_vox_feat, d_rayfeat,
                              accumulate_inputs, grads, ws + l.inner, l.w, cus, st);
}

// ---- per-pair / per-ray tail of get_pred, forward and adjoint -------------------------------------
LIDF_API int lidf_query_tail_f32(const float* pred_offset, const float* pred_prob,
                                   const int32_t* pair_off, const int32_t* pair_ray,
                                   const float* pair_t, const float* ray_dir, int64_t n_rays,
                                   int64_t n_pairs, float offset_range0, float offset_range1,
                                   float part_size, const int64_t* max_pair_id_in,
                                   float* pair_pred_pos, float* softmax, int64_t* max_pair_id,
                                   float* pred_pos, lidf_stream_t stream) {
    if (n_rays < 0 || n_pairs < 0) return LIDF_ERR_BAD_ARG;
    if (n_rays == 0) return LIDF_OK;
    if (!pair_off || (n_pairs > 0 && (!pred_offset || !pred_prob || !pair_ray || !pair_t || !ray_dir ||
                                      !pair_pred_pos)))
        return LIDF_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    CHECK_HIP(lidf_launch_pair_pos(pred_offset, pair_ray, pair_t, ray_dir, n_pairs, offset_range0,
                                   offset_range1 - offset_range0, (float)1.7320508075688772,
                                   part_size, pair_pred_pos, st));
    // softmax over the pairs of a ray, arg-max, select (pipeline.py:442-454)
    CHECK_HIP(lidf_launch_ray_reduce(pred_prob, pair_pred_pos, pair_off, n_rays, n_pairs, nullptr,
                                     nullptr, 0, softmax, (long long*)max_pair_id,
                                     max_pair_id_in ? nullptr : pred_pos, nullptr, st));
    // training with ground-truth labels selects the pair itself (pipeline.py:444-446)
    if (max_pair_id_in && pred_pos)
        CHECK_HIP(lidf_launch_ray_select(pair_pred_pos, (const long long*)max_pair_id_in, n_rays,
                                         n_pairs, pred_pos, st));
    return LIDF_OK;
}

LIDF_API int lidf_query_tail_backward_f32(const float* g_pair_pred_pos, const float* g_pred_pos,
                                            const int64_t* max_pair_id, const int32_t* pair_ray,
                                            const float* ray_dir, int64_t n_rays, int64_t n_pairs,
                                            float offset_range0, float offset_range1,
                                            float part_size, float* d_pred_offset,
                                            lidf_stream_t stream) {
    if (n_rays < 0 || n_pairs < 0) return LIDF_ERR_BAD_ARG;
    if (n_pairs == 0) return LIDF_OK;
    if (!pair_ray || !ray_dir || !d_pred_offset || (g_pred_pos && !max_pair_id)) return LIDF_ERR_BAD_ARG;
    const float k = (offset_range1 - offset_range0) * (float)1.7320508075688772 * part_size;
    CHECK_HIP(lidf_launch_pair_pos_backward(g_pair_pred_pos, g_pred_pos, (const long long*)max_pair_id,
                                            pair_ray, ray_dir, n_pairs, k, d_pred_offset,
                                            (hipStream_t)stream));
    return LIDF_OK;
}

// ---- positional encoding, backward ---------------------------------------------------------------
LIDF_API int lidf_embed_backward_f32(const float* x, const float* g_out, int64_t n, int multires,
                                       float* d_x, lidf_stream_t stream) {
    if (n < 0 || multires < 0 || multires > 16) return LIDF_ERR_BAD_ARG;
    if (n == 0) return LIDF_OK;
    if (!x || !g_out || !d_x) return LIDF_ERR_BAD_ARG;
    CHECK_HIP(lidf_launch_embed_backward(x, g_out, n, multires, d_x, (hipStream_t)stream));
    return LIDF_OK;
}

// ---- PointNet2Stage, training path ----------------------------------------------------------------
struct PnetAct {   // offsets in floats into the caller's `act`
    size_t f1, f2, f4, f5, pool1, g1, pool2, out, arg1, arg2, total;
};
static PnetAct pnet_act(int64_t n, int64_t v) {
    PnetAct a;
    const size_t N = (size_t)(n > 0 ? n : 1), V = (size_t)(v > 0 ? v : 1);
    size_t o = 0;
    a.f1 = o; o += N * 32;
    a.f2 = o; o += N * 64;
    a.f4 = o; o += N * 128;
    a.f5 = o; o += N * 128;
    a.pool1 = o; o += V * 64;
    a.g1 = o; o += V * 64;
    a.pool2 = o; o += V * 128;
    a.out = o; o += V * 128;
    a.arg1 = o; o += V * 64;    // int32
    a.arg2 = o; o += V * 128;   // int32
    a.total = o;
    return a;
}
struct PnetTrainWs {
    size_t s[7], gpart, stream, dz_out, dp2, dz5, dz4, s4, dg1, df2, dp1, dz1, wg, total;
};
static PnetTrainWs pnet_train_ws(int64_t n, int64_t v) {
    PnetTrainWs w;
    const size_t N = (size_t)(n > 0 ? n : 1), V = (size_t)(v > 0 ? v : 1);
    size_t o = 0;
    const int ks[7] = {6, 32, 64, 64, 64, 128, 128};
    const int nts[7] = {1, 2, 2, 4, 4, 4, 4};
    for (int i = 0; i < 7; ++i) { w.s[i] = o; o += lin_stream_bytes(ks[i], nts[i]); }
    w.gpart = o;  o += align_up(V * 128 * 4, 256);
    w.stream = o; o += linex_stream_bytes(256);
    w.dz_out = o; o += align_up(V * 128 * 4, 256);
    w.dp2 = o;    o += align_up(V * 128 * 4, 256);
    w.dz5 = o;    o += align_up(N * 128 * 4, 256);
    w.dz4 = o;    o += align_up(N * 128 * 4, 256);
    w.s4 = o;     o += align_up(V * 128 * 4, 256);
    w.dg1 = o;    o += align_up(V * 64 * 4, 256);
    w.df2 = o;    o += align_up(N * 64 * 4, 256);
    w.dp1 = o;    o += align_up(V * 64 * 4, 256);
    w.dz1 = o;    o += align_up(N * 32 * 4, 256);
    w.wg = o;     o += align_up(WG_SCRATCH_FLOATS * 4, 256);
    w.total = o;
    return w;
}

// (lidf_pointnet_train_act_floats / _workspace_bytes: lidf_api_pointnet_train.inc — the larger of this path's and
// the training chains' needs)

// forward with the rows kept: layer by layer through lidf_linear_kernel (mode 0: pack and run, 2: streams packed
// earlier), then the arg row of every pooled entry
static int pointnet_train_forward_impl(const LidfPointNet* w, const float* inp, const int32_t* vox, int64_t n,
                                       int64_t n_vox, float* out, float* act, float* gpart,
                                       float* const streams[7], int cus, hipStream_t st, int mode) {
    int rc;
    const PnetAct a = pnet_act(n, n_vox);
    PnetBufs b = {};
    b.f1 = act + a.f1; b.f2 = act + a.f2; b.f4 = act + a.f4; b.f5 = act + a.f5;
    b.pool1 = act + a.pool1; b.g1 = act + a.g1; b.pool2 = act + a.pool2;
    b.gpart = gpart;
    for (int i = 0; i < 7; ++i) b.streams[i] = streams[i];
    if ((rc = pointnet_impl(w, inp, vox, n, n_vox, out, b, cus, st, mode))) return rc;
    if (mode == 1) return LIDF_OK;
    if (out != act + a.out)
        CHECK_HIP(hipMemcpyAsync(act + a.out, out, (size_t)n_vox * 128 * 4, hipMemcpyDeviceToDevice, st));
    // arg of every pooled entry: the lowest row attaining the maximum
    CHECK_HIP(hipMemsetAsync(act + a.arg1, 0x7f, (size_t)n_vox * 64 * 4, st));
    CHECK_HIP(hipMemsetAsync(act + a.arg2, 0x7f, (size_t)n_vox * 128 * 4, st));
    CHECK_HIP(lidf_launch_segmax_arg(b.f2, vox, b.pool1, n, 64, (int*)(act + a.arg1), st));
    CHECK_HIP(lidf_launch_segmax_arg(b.f5, vox, b.pool2, n, 128, (int*)(act + a.arg2), st));
    return LIDF_OK;
}

static int pointnet_forward_train_layers(const LidfPointNet* w, const float* inp,
                                               const int32_t* vox, int64_t n, int64_t n_vox,
                                               float* out, float* act, void* workspace,
                                               size_t workspace_bytes, lidf_stream_t stream) {
    if (!w || n < 0 || n_vox < 0) return LIDF_ERR_BAD_ARG;
    if (n_vox == 0) return LIDF_OK;
    if (!out || !act || (n > 0 && (!inp || !vox))) return LIDF_ERR_BAD_ARG;
    int rc;
    if ((rc = check_pointnet_w(w))) return rc;
    const PnetTrainWs ws = pnet_train_ws(n, n_vox);
    if (!workspace || workspace_bytes < ws.total) return LIDF_ERR_WORKSPACE;
    char* base = (char*)workspace;
    float* streams[7];
    for (int i = 0; i < 7; ++i) streams[i] = (float*)(base + ws.s[i]);
    int cus;
    if ((rc = cu_count(&cus))) return rc;
    return pointnet_train_forward_impl(w, inp, vox, n, n_vox, out, act, (float*)(base + ws.gpart), streams, cus,
                                       (hipStream_t)stream, 0);
}

// The backward. slots: stream slots of its seven transposed launches (NULL: one slot of the workspace serves
// them in turn), pack_mode of those launches (0 pack and run, 1 pack only, 2 packed earlier); zero_grads false:
// the parameter gradients are accumulated into.
static int pointnet_backward_impl(const LidfPointNet* w, const float* inp, const int32_t* vox, int64_t n,
                                  int64_t n_vox, const float* act, const float* g_out, float* d_inp,
                                  const LidfPointNetGrads* g, char* base, const PnetTrainWs& ws, bool zero_grads,
                                  float* const* slots, int pack_mode, int cus, hipStream_t st) {
    int rc;
    const bool po = pack_mode == 1;
    if (zero_grads && !po) {
        float* const zp[12] = {g->w_p1, g->b_p1, g->w_p2, g->b_p2, g->w_v1, g->b_v1, g->w_p3, g->b_p3,
                               g->w_p4, g->b_p4, g->w_v2, g->b_v2};
        const long long zc[12] = {32 * 6, 32, 64 * 32, 64, 64 * 64, 64, 128 * 128, 128, 128 * 128, 128, 128 * 128, 128};
        CHECK_HIP(lidf_launch_zero_segments(zp, zc, 12, st));
    }
    if (d_inp && n > 0 && !po) CHECK_HIP(hipMemsetAsync(d_inp, 0, (size_t)n * 6 * 4, st));
    if (n_vox == 0 && !po) return LIDF_OK;
    const PnetAct a = pnet_act(n, n_vox);
    float* sbuf = (float*)(base + ws.stream);
    float* dz_out = (float*)(base + ws.dz_out);
    float* dp2 = (float*)(base + ws.dp2);
    float* dz5 = (float*)(base + ws.dz5);
    float* dz4 = (float*)(base + ws.dz4);
    float* s4 = (float*)(base + ws.s4);
    float* dg1 = (float*)(base + ws.dg1);
    float* df2 = (float*)(base + ws.df2);
    float* dp1 = (float*)(base + ws.dp1);
    float* dz1 = (float*)(base + ws.dz1);
    float* wgs = (float*)(base + ws.wg);
    const float *f1 = act + a.f1, *f2 = act + a.f2, *f4 = act + a.f4;
    const float *pool1 = act + a.pool1, *g1 = act + a.g1, *pool2 = act + a.pool2, *outv = act + a.out;
    const int *arg1 = (const int*)(act + a.arg1), *arg2 = (const int*)(act + a.arg2);
    const int64_t V = n_vox;
    int slot = 0;
    auto lin = [&](const LinEx& L) { float* sp = slots ? slots[slot] : sbuf; ++slot; return run_linex(L, sp, cus, st, pack_mode); };
    // out = relu(vox_lin2(pool2))
    if (!po) {
        CHECK_HIP(lidf_launch_relu_mask(g_out, outv, V * 128, dz_out, st));
        CHECK_HIP(lidf_launch_wgrad(dz_out, 128, 128, pool2, 128, 128, V, g->w_v2, 128, g->b_v2, wgs, WG_SCRATCH_FLOATS, st));
    }
    LinEx L = {};
    L.transposed = 1; L.mask_slope = 0.f;
    L.n = V; L.w = w->w_v2; L.ldw = 128; L.nout = 128; L.k = 128; L.X = dz_out; L.ldx = 128;
    L.out = dp2; L.ld_out = 128;
    if ((rc = lin(L))) return rc;
    if (n == 0 && !po) return LIDF_OK;   // no points: the pooled inputs were the zero fill
    // pool2 = segmax(f5): d f5 goes to the arg rows; f5 = relu(point_lin4(f4))
    if (!po) {
        CHECK_HIP(lidf_launch_segmax_backward(dp2, arg2, vox, pool2, n, 128, 0, dz5, st));
        CHECK_HIP(lidf_launch_wgrad(dz5, 128, 128, f4, 128, 128, n, g->w_p4, 128, g->b_p4, wgs, WG_SCRATCH_FLOATS, st));
    }
    L.n = n; L.w = w->w_p4; L.X = dz5; L.mask_src = f4; L.ld_mask = 128; L.out = dz4;
    if ((rc = lin(L))) return rc;
    // f4 = relu(point_lin3(cat(g1[vox], f2))): weight columns 0..63 meet g1[vox], 64..127 meet f2
    if (!po) {
        CHECK_HIP(hipMemsetAsync(s4, 0, (size_t)V * 128 * 4, st));
        CHECK_HIP(lidf_launch_seg_sum_rows(dz4, vox, n, 128, s4, st));
        CHECK_HIP(lidf_launch_wgrad(s4, 128, 128, g1, 64, 64, V, g->w_p3, 128, nullptr, wgs, WG_SCRATCH_FLOATS, st));
        CHECK_HIP(lidf_launch_wgrad(dz4, 128, 128, f2, 64, 64, n, g->w_p3 + 64, 128, g->b_p3, wgs, WG_SCRATCH_FLOATS, st));
    }
    L.mask_src = nullptr;
    L.n = n; L.w = w->w_p3 + 64; L.ldw = 128; L.nout = 64; L.k = 128; L.X = dz4; L.ldx = 128;
    L.out = df2; L.ld_out = 64;
    if ((rc = lin(L))) return rc;
    // g1 = relu(vox_lin1(pool1))
    L.n = V; L.w = w->w_p3; L.X = s4; L.mask_src = g1; L.ld_mask = 64; L.out = dg1;
    if ((rc = lin(L))) return rc;
    if (!po) CHECK_HIP(lidf_launch_wgrad(dg1, 64, 64, pool1, 64, 64, V, g->w_v1, 64, g->b_v1, wgs, WG_SCRATCH_FLOATS, st));
    L.mask_src = nullptr;
    L.w = w->w_v1; L.ldw = 64; L.nout = 64; L.k = 64; L.X = dg1; L.ldx = 64; L.out = dp1; L.ld_out = 64;
    if ((rc = lin(L))) return rc;
    // pool1 = segmax(f2): added to the gradient f2 receives through the concat; f2 = relu(point_lin2(f1))
    if (!po) {
        CHECK_HIP(lidf_launch_segmax_backward(dp1, arg1, vox, pool1, n, 64, 1, df2, st));
        CHECK_HIP(lidf_launch_relu_mask(df2, f2, n * 64, df2, st));
        CHECK_HIP(lidf_launch_wgrad(df2, 64, 64, f1, 32, 32, n, g->w_p2, 32, g->b_p2, wgs, WG_SCRATCH_FLOATS, st));
    }
    L.n = n; L.w = w->w_p2; L.ldw = 32; L.nout = 32; L.k = 64; L.X = df2; L.ldx = 64;
    L.mask_src = f1; L.ld_mask = 32; L.out = dz1; L.ld_out = 32;
    if ((rc = lin(L))) return rc;
    if (!po) CHECK_HIP(lidf_launch_wgrad(dz1, 32, 32, inp, 6, 6, n, g->w_p1, 6, g->b_p1, wgs, WG_SCRATCH_FLOATS, st));
    if (d_inp || po) {
        L.mask_src = nullptr;
        L.w = w->w_p1; L.ldw = 6; L.nout = 6; L.k = 32; L.X = dz1; L.ldx = 32; L.out = d_inp; L.ld_out = 6;
        if ((rc = lin(L))) return rc;
    }
    return LIDF_OK;
}
#define PNET_BWD_SLOTS 7
static const int PNET_BWD_SLOT_K[PNET_BWD_SLOTS] = {128, 128, 128, 128, 64, 64, 32};   // contraction lengths, in launch order

static int pointnet_backward_layers(const LidfPointNet* w, const float* inp, const int32_t* vox,
                                          int64_t n, int64_t n_vox, const float* act,
                                          const float* g_out, float* d_inp,
                                          const LidfPointNetGrads* g, void* workspace,
                                          size_t workspace_bytes, lidf_stream_t stream) {
    if (!w || !g || n < 0 || n_vox < 0) return LIDF_ERR_BAD_ARG;
    int rc;
    if ((rc = check_pointnet_w(w))) return rc;
    if (!g->w_p1 || !g->b_p1 || !g->w_p2 || !g->b_p2 || !g->w_v1 || !g->b_v1 || !g->w_p3 || !g->b_p3 ||
        !g->w_p4 || !g->b_p4 || !g->w_v2 || !g->b_v2)
        return LIDF_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (n_vox > 0 && (!act || !g_out || (n > 0 && (!inp || !vox)))) return LIDF_ERR_BAD_ARG;
    const PnetTrainWs ws = pnet_train_ws(n, n_vox);
    if (n_vox > 0 && (!workspace || workspace_bytes < ws.total)) return LIDF_ERR_WORKSPACE;
    int cus;
    if ((rc = cu_count(&cus))) return rc;
    return pointnet_backward_impl(w, inp, vox, n, n_vox, act, g_out, d_inp, g, (char*)workspace, ws, true, nullptr,
                                  0, cus, st);
}

#include "lidf_api_pointnet_train.inc"
#include "lidf_api_refine_train.inc"
