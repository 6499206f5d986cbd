// lidf_torch_ext.cpp — thin pybind11 torch-extension shim over the C ABI of liblidf_hip.so
// (include/lidf_hip.h), the counterpart of the reference's operator boundary:
//   extensions/ray_aabb/ray_aabb_cuda.cpp:20-37, extensions/pcl_aabb/pcl_aabb_cuda.cpp:20-37
//   (pybind module with one `forward(Tensor...) -> Tensor | vector<Tensor>` per op, built by
//   torch.utils.cpp_extension, extensions/*/jit.py:2-3).
// The shim only validates tensors (device, dtype, contiguity, shapes), allocates the outputs with
// torch's allocator, fetches the CURRENT HIP stream and calls the C ABI; a non-zero status becomes
// TORCH_CHECK(false, lidf_strerror(rc)) like the reference's CHECK_* macros. No compute, no state.
// Built by implicit_depth_amd/csrc/build.py (g++, links liblidf_hip.so next to it).
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/extension.h>

#include <vector>

#include "lidf_hip.h"

namespace {

using torch::Tensor;

#define CHECK_DEV(x) TORCH_CHECK((x).is_cuda(), #x " must be a CUDA tensor")
#define CHECK_CONTIG(x) TORCH_CHECK((x).is_contiguous(), #x " must be contiguous")
#define CHECK_IN(x) \
    CHECK_DEV(x);   \
    CHECK_CONTIG(x)
#define CHECK_F32(x) TORCH_CHECK((x).scalar_type() == torch::kFloat32, #x " must be float32")
#define CHECK_I32(x) TORCH_CHECK((x).scalar_type() == torch::kInt32, #x " must be int32")

void check_rc(int rc) { TORCH_CHECK(rc == LIDF_OK, lidf_strerror(rc)); }

// [n, 3] points / directions against [V, 6] boxes with one image index per row of each: the kernels
// index these shapes unchecked, so a wrong inner dimension or a short index tensor would read out of
// bounds on the device
void check_box_shapes(const Tensor& pts, const Tensor& voxel_bound, const Tensor& pts_bid,
                      const Tensor& voxel_bid, const char* what) {
    TORCH_CHECK(pts.dim() == 2 && pts.size(1) == 3, what, " must be [n,3]");
    TORCH_CHECK(voxel_bound.dim() == 2 && voxel_bound.size(1) == 6, "voxel_bound must be [V,6]");
    TORCH_CHECK(pts_bid.dim() == 1 && pts_bid.size(0) == pts.size(0), "one image index per row of ", what);
    TORCH_CHECK(voxel_bid.dim() == 1 && voxel_bid.size(0) == voxel_bound.size(0),
                "voxel_bid must be [V]");
}

// every tensor of a call on the device of the first one (the shim launches on that device's current
// stream, with that device current)
void same_device(std::initializer_list<const Tensor*> ts) {
    const Tensor* first = nullptr;
    for (const Tensor* t : ts) {
        if (!t || !t->defined()) continue;
        if (!first) first = t;
        TORCH_CHECK(t->device() == first->device(), "all tensors must be on the same device");
    }
}
void same_device(const Tensor& ref, const std::vector<Tensor>& ts) {
    for (const auto& t : ts) TORCH_CHECK(t.device() == ref.device(), "all tensors must be on the same device");
}

lidf_stream_t current_stream(const Tensor& t) {
    return (lidf_stream_t)c10::hip::getCurrentHIPStream(t.device().index()).stream();
}

// int64 index tensors of the reference's data_dict are narrowed once (the kernels take int32)
Tensor as_i32(const Tensor& t, const char* name) {
    if (t.scalar_type() == torch::kInt32) return t;
    TORCH_CHECK(t.scalar_type() == torch::kInt64, name, " must be int32 or int64");
    return t.to(torch::kInt32);
}

// [w1,b1,w2,b2,w3,b3,w4,b4] (IMNet) or [..., offset_enc.weight, offset_enc.bias] (IEF): the state
// dict order of models/implicit_net.py
LidfDecoder decoder_of(const std::vector<Tensor>& w, int64_t n_iter, double init_offset,
                       bool use_sigmoid) {
    TORCH_CHECK(w.size() == 8 || w.size() == 10, "decoder weights: 8 (IMNet) or 10 (IEF) tensors");
    for (const auto& t : w) {
        CHECK_IN(t);
        CHECK_F32(t);
    }
    LidfDecoder d = {};
    d.w1 = w[0].data_ptr<float>(); d.b1 = w[1].data_ptr<float>();
    d.w2 = w[2].data_ptr<float>(); d.b2 = w[3].data_ptr<float>();
    d.w3 = w[4].data_ptr<float>(); d.b3 = w[5].data_ptr<float>();
    d.w4 = w[6].data_ptr<float>(); d.b4 = w[7].data_ptr<float>();
    d.is_ief = w.size() == 10;
    if (d.is_ief) {
        d.wenc = w[8].data_ptr<float>();
        d.benc = w[9].data_ptr<float>();
        d.n_iter = (int32_t)n_iter;
        d.init_offset = (float)init_offset;
    } else {
        d.n_iter = 1;
    }
    d.use_sigmoid = use_sigmoid ? 1 : 0;
    return d;
}

// ---- extensions/ray_aabb: forward(ray_dir, voxel_bound, ray_bid, voxel_bid) -> [mask, dist] ------
std::vector<Tensor> ray_aabb(Tensor ray_dir, Tensor voxel_bound, Tensor ray_bid, Tensor voxel_bid) {
    CHECK_IN(ray_dir); CHECK_IN(voxel_bound); CHECK_IN(ray_bid); CHECK_IN(voxel_bid);
    CHECK_F32(ray_dir); CHECK_F32(voxel_bound); CHECK_I32(ray_bid); CHECK_I32(voxel_bid);
    same_device({&ray_dir, &voxel_bound, &ray_bid, &voxel_bid});
    check_box_shapes(ray_dir, voxel_bound, ray_bid, voxel_bid, "ray_dir");
    const c10::DeviceGuard guard(ray_dir.device());
    const int64_t R = ray_dir.size(0), V = voxel_bound.size(0);
    auto mask = torch::zeros({V, R}, ray_bid.options());       // ray_aabb_cuda_kernel.cu:105-106
    auto dist = torch::zeros({V, R, 2}, ray_dir.options());
    check_rc(lidf_ray_aabb_dense_f32(ray_dir.data_ptr<float>(), voxel_bound.data_ptr<float>(),
                                     ray_bid.data_ptr<int32_t>(), voxel_bid.data_ptr<int32_t>(), R, V,
                                     mask.data_ptr<int32_t>(), dist.data_ptr<float>(),
                                     current_stream(ray_dir)));
    return {mask, dist};
}

// ---- extensions/pcl_aabb: forward(pcl, voxel_bound, pcl_bid, voxel_bid) -> mask ------------------
Tensor pcl_aabb(Tensor pcl, Tensor voxel_bound, Tensor pcl_bid, Tensor voxel_bid) {
    CHECK_IN(pcl); CHECK_IN(voxel_bound); CHECK_IN(pcl_bid); CHECK_IN(voxel_bid);
    CHECK_F32(pcl); CHECK_F32(voxel_bound); CHECK_I32(pcl_bid); CHECK_I32(voxel_bid);
    same_device({&pcl, &voxel_bound, &pcl_bid, &voxel_bid});
    check_box_shapes(pcl, voxel_bound, pcl_bid, voxel_bid, "pcl");
    const c10::DeviceGuard guard(pcl.device());
    const int64_t N = pcl.size(0), V = voxel_bound.size(0);
    auto mask = torch::zeros({V, N}, pcl_bid.options());
    check_rc(lidf_pcl_aabb_dense_f32(pcl.data_ptr<float>(), voxel_bound.data_ptr<float>(),
                                     pcl_bid.data_ptr<int32_t>(), voxel_bid.data_ptr<int32_t>(), N, V,
                                     mask.data_ptr<int32_t>(), current_stream(pcl)));
    return mask;
}

// ---- compact ray-major candidates (replaces ray_aabb.forward + nonzero, pipeline.py:277-285) -----
std::vector<Tensor> compute_ray_aabb(Tensor ray_dir, Tensor voxel_bound, Tensor ray_bid,
                                     Tensor voxel_bid) {
    CHECK_IN(ray_dir); CHECK_IN(voxel_bound); CHECK_IN(ray_bid); CHECK_IN(voxel_bid);
    CHECK_F32(ray_dir); CHECK_F32(voxel_bound); CHECK_I32(ray_bid); CHECK_I32(voxel_bid);
    same_device({&ray_dir, &voxel_bound, &ray_bid, &voxel_bid});
    check_box_shapes(ray_dir, voxel_bound, ray_bid, voxel_bid, "ray_dir");
    const c10::DeviceGuard guard(ray_dir.device());
    const int64_t R = ray_dir.size(0), V = voxel_bound.size(0);
    auto st = current_stream(ray_dir);
    auto iopt = ray_bid.options();
    auto count = torch::empty({std::max<int64_t>(R, 1)}, iopt);
    auto pair_off = torch::zeros({R + 1}, iopt);
    if (R > 0) {
        check_rc(lidf_ray_aabb_count_f32(ray_dir.data_ptr<float>(), voxel_bound.data_ptr<float>(),
                                         ray_bid.data_ptr<int32_t>(), voxel_bid.data_ptr<int32_t>(), R,
                                         V, count.data_ptr<int32_t>(), st));
        const size_t wsb = lidf_exclusive_scan_workspace_bytes(R);
        auto ws = torch::empty({(int64_t)wsb}, iopt.dtype(torch::kUInt8));
        check_rc(lidf_exclusive_scan_i32(count.data_ptr<int32_t>(), R, pair_off.data_ptr<int32_t>(),
                                         ws.data_ptr(), wsb, st));
    }
    const int64_t P = pair_off[R].item<int32_t>();  // the host sizes the outputs, as nonzero() does
    auto pair_ray = torch::empty({P}, iopt), pair_vox = torch::empty({P}, iopt);
    auto pair_t = torch::empty({P, 2}, ray_dir.options());
    if (P > 0)
        check_rc(lidf_ray_aabb_fill_f32(ray_dir.data_ptr<float>(), voxel_bound.data_ptr<float>(),
                                        ray_bid.data_ptr<int32_t>(), voxel_bid.data_ptr<int32_t>(), R, V,
                                        pair_off.data_ptr<int32_t>(), pair_ray.data_ptr<int32_t>(),
                                        pair_vox.data_ptr<int32_t>(), pair_t.data_ptr<float>(), st));
    return {pair_off, pair_ray, pair_vox, pair_t};
}

// ---- IMNet.forward / IEF.forward on [n, D] rows (models/pipeline.py:434-435) ----------------------
std::vector<Tensor> forward_decoders(Tensor inp, std::vector<Tensor> prob_w, std::vector<Tensor> off_w,
                                     int64_t off_n_iter, double off_init, bool use_sigmoid,
                                     int64_t precision) {
    CHECK_DEV(inp); CHECK_F32(inp);
    TORCH_CHECK(inp.dim() == 2 && inp.stride(1) == 1, "inp_feat must be [n, D] with unit column stride");
    TORCH_CHECK(!prob_w.empty() || !off_w.empty(), "need at least one decoder");
    same_device(inp, prob_w); same_device(inp, off_w);
    const c10::DeviceGuard guard(inp.device());
    const int64_t n = inp.size(0), d = inp.size(1), ld = n > 1 ? inp.stride(0) : d;
    LidfDecoder dp = {}, dof = {};
    if (!prob_w.empty()) dp = decoder_of(prob_w, 1, 0.0, use_sigmoid);
    if (!off_w.empty()) dof = decoder_of(off_w, off_n_iter, off_init, use_sigmoid);
    Tensor out_p = prob_w.empty() ? Tensor() : torch::empty({n, 1}, inp.options());
    Tensor out_o = off_w.empty() ? Tensor() : torch::empty({n, 1}, inp.options());
    const size_t wsb = lidf_decoders_workspace_bytes(n, (int)d);
    auto ws = torch::empty({(int64_t)std::max<size_t>(wsb, 1)}, inp.options().dtype(torch::kUInt8));
    auto fn = precision == LIDF_PRECISION_F16X3 ? lidf_decoders_split_f32 : lidf_decoders_f32;
    check_rc(fn(inp.data_ptr<float>(), n, (int)d, ld, prob_w.empty() ? nullptr : &dp,
                off_w.empty() ? nullptr : &dof, prob_w.empty() ? nullptr : out_p.data_ptr<float>(),
                off_w.empty() ? nullptr : out_o.data_ptr<float>(), ws.data_ptr(), wsb,
                current_stream(inp)));
    return {out_p, out_o};
}

// ---- get_embedding + get_pred (+ depth write-back), models/pipeline.py:338-466, :593-596 ----------
// Returns [pred_offset [P,1], pred_prob_end [P,1], pair_pred_pos [P,3], pred_prob_end_softmax [P],
//          max_pair_id [R] i64, pred_pos [R,3]]; `depth` [B,h,w] (optional) is updated in place.
std::vector<Tensor> forward_query(Tensor ray_dir, Tensor ray_pix, Tensor ray_bid,
                                  c10::optional<Tensor> ray_flat, Tensor pair_off, Tensor pair_ray,
                                  Tensor pair_vox, Tensor pair_t, Tensor feat_grid, Tensor vox_feat,
                                  c10::optional<Tensor> vox_center, std::vector<Tensor> prob_w,
                                  std::vector<Tensor> off_w, int64_t off_n_iter, double off_init,
                                  bool use_sigmoid, int64_t multires, int64_t multires_views,
                                  int64_t roi_inp_bbox, bool pos_rel, double offset_range0,
                                  double offset_range1, double part_size, c10::optional<Tensor> depth,
                                  int64_t precision) {
    CHECK_IN(ray_dir); CHECK_IN(pair_off); CHECK_IN(pair_ray); CHECK_IN(pair_vox); CHECK_IN(pair_t);
    CHECK_IN(feat_grid); CHECK_IN(vox_feat);
    CHECK_F32(ray_dir); CHECK_F32(pair_t); CHECK_F32(feat_grid); CHECK_F32(vox_feat);
    CHECK_I32(pair_off); CHECK_I32(pair_ray); CHECK_I32(pair_vox);
    Tensor pix = as_i32(ray_pix, "ray_pix").contiguous(), bid = as_i32(ray_bid, "ray_bid").contiguous();
    CHECK_DEV(pix); CHECK_DEV(bid);
    same_device({&ray_dir, &pix, &bid, &pair_off, &pair_ray, &pair_vox, &pair_t, &feat_grid, &vox_feat,
                 ray_flat.has_value() ? &*ray_flat : nullptr, vox_center.has_value() ? &*vox_center : nullptr,
                 depth.has_value() ? &*depth : nullptr});
    same_device(ray_dir, prob_w); same_device(ray_dir, off_w);
    const c10::DeviceGuard guard(ray_dir.device());
    TORCH_CHECK(ray_dir.dim() == 2 && ray_dir.size(1) == 3, "ray_dir must be [R,3]");
    TORCH_CHECK(feat_grid.dim() == 4 && feat_grid.size(1) == 32, "feat_grid must be [B,32,h,w]");
    TORCH_CHECK(vox_feat.dim() == 2 && vox_feat.size(1) == 128, "vox_feat must be [V,128]");
    const int64_t R = ray_dir.size(0), P = pair_ray.size(0), V = vox_feat.size(0);
    const int64_t B = feat_grid.size(0), h = feat_grid.size(2), w = feat_grid.size(3);
    TORCH_CHECK(pair_off.size(0) == R + 1, "pair_off must have R+1 entries");
    TORCH_CHECK(pix.dim() == 2 && pix.size(0) == R && pix.size(1) == 2 && bid.dim() == 1 && bid.size(0) == R,
                "ray_pix / ray_bid must be [R,2] / [R]");
    TORCH_CHECK(pair_ray.dim() == 1 && pair_off.dim() == 1, "pair_ray / pair_off must be 1-d");
    TORCH_CHECK(pair_vox.dim() == 1 && pair_vox.size(0) == P && pair_t.dim() == 2 && pair_t.size(0) == P &&
                    pair_t.size(1) == 2, "pair_vox / pair_t must be [P] / [P,2]");
    LidfDecoder dp = decoder_of(prob_w, 1, 0.0, use_sigmoid);
    LidfDecoder dof = decoder_of(off_w, off_n_iter, off_init, use_sigmoid);
    auto fopt = ray_dir.options();
    auto pred_offset = torch::empty({P, 1}, fopt), pred_prob = torch::empty({P, 1}, fopt);
    auto pair_pred_pos = torch::empty({P, 3}, fopt), softmax = torch::empty({P}, fopt);
    auto max_pair_id = torch::empty({R}, fopt.dtype(torch::kInt64));
    auto pred_pos = torch::empty({R, 3}, fopt);
    const size_t wsb = lidf_query_workspace_bytes(R, V, B * 32 * h * w);
    auto ws = torch::empty({(int64_t)wsb}, fopt.dtype(torch::kUInt8));
    LidfQueryArgs q = {};
    q.n_rays = R; q.ray_dir = ray_dir.data_ptr<float>(); q.ray_pix = pix.data_ptr<int32_t>();
    q.ray_bid = bid.data_ptr<int32_t>();
    Tensor flat;
    if (ray_flat.has_value()) {
        flat = as_i32(*ray_flat, "ray_flat").contiguous();
        CHECK_DEV(flat);
        TORCH_CHECK(flat.dim() == 1 && flat.size(0) == R, "ray_flat must be [R]");
        q.ray_flat = flat.data_ptr<int32_t>();
    }
    q.n_pairs = P; q.pair_off = pair_off.data_ptr<int32_t>();
    q.pair_ray = pair_ray.data_ptr<int32_t>(); q.pair_vox = pair_vox.data_ptr<int32_t>();
    q.pair_t = pair_t.data_ptr<float>();
    q.batch = (int32_t)B; q.height = (int32_t)h; q.width = (int32_t)w;
    q.feat_grid = feat_grid.data_ptr<float>();
    q.n_vox = V; q.vox_feat = vox_feat.data_ptr<float>();
    if (vox_center.has_value()) {
        CHECK_IN(*vox_center); CHECK_F32(*vox_center);
        TORCH_CHECK(vox_center->size(0) == V && vox_center->size(1) == 3, "vox_center must be [V,3]");
        q.vox_center = vox_center->data_ptr<float>();
    }
    q.prob = &dp; q.off = &dof;
    q.multires = (int32_t)multires; q.multires_views = (int32_t)multires_views;
    q.roi_inp_bbox = (int32_t)roi_inp_bbox; q.pos_rel = pos_rel ? 1 : 0;
    q.offset_range0 = (float)offset_range0; q.offset_range1 = (float)offset_range1;
    q.part_size = (float)part_size;
    q.pred_offset = pred_offset.data_ptr<float>(); q.pred_prob = pred_prob.data_ptr<float>();
    q.pair_pred_pos = pair_pred_pos.data_ptr<float>(); q.pred_prob_softmax = softmax.data_ptr<float>();
    q.max_pair_id = max_pair_id.data_ptr<int64_t>(); q.pred_pos = pred_pos.data_ptr<float>();
    if (depth.has_value()) {
        CHECK_IN(*depth); CHECK_F32(*depth);
        TORCH_CHECK(depth->dim() == 3 && depth->size(0) == B && depth->size(1) == h && depth->size(2) == w,
                    "depth must be [B,h,w]");
        TORCH_CHECK(ray_flat.has_value(), "depth needs ray_flat");
        q.depth = depth->data_ptr<float>();
    }
    q.workspace = ws.data_ptr(); q.workspace_bytes = wsb;
    q.precision = (int32_t)precision;
    check_rc(lidf_query_f32(&q, current_stream(ray_dir)));
    return {pred_offset, pred_prob, pair_pred_pos, softmax, max_pair_id, pred_pos};
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "liblidf_hip torch-extension shim (C ABI: include/lidf_hip.h)";
    // the shim was compiled against LIDF_ABI_VERSION of the header; a liblidf_hip.so of another
    // version beside it (stale build: both are git-ignored artefacts) must not be driven
    TORCH_CHECK(lidf_version() == LIDF_ABI_VERSION, "liblidf_hip.so answers ABI ", lidf_version(),
                ", lidf_torch_ext was compiled against ABI ", LIDF_ABI_VERSION, " — rebuild both");
    m.def("abi_version", []() { return lidf_version(); });
    m.def("ray_aabb", &ray_aabb, "dense ray/voxel slab test: [mask, dist] (extensions/ray_aabb forward)");
    m.def("pcl_aabb", &pcl_aabb, "dense point/voxel inside test: mask (extensions/pcl_aabb forward)");
    m.def("compute_ray_aabb", &compute_ray_aabb, "compact ray-major pairs: [pair_off, pair_ray, pair_vox, pair_t]");
    m.def("forward_decoders", &forward_decoders, "IMNet / IEF forward on [n, D] rows: [prob, offset]");
    m.def("forward_query", &forward_query, "fused get_embedding + get_pred (+ depth)");
}
