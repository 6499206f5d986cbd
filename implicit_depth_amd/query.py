"""query.py — host-side mirror of the LIDF query methods for the MI355X path.

Reference counterparts (paths relative to /root/reference/src):
  ray_dirs            <- LIDF.get_miss_ray, dense part         models/pipeline.py:208-220
  compute_ray_aabb    <- LIDF.compute_ray_aabb                 models/pipeline.py:271-296
  lidf_query          <- LIDF.get_embedding + LIDF.get_pred    models/pipeline.py:338-466
                         + depth write-back                     models/pipeline.py:593-596
  lidf_refine         <- RefineNet.forward / get_pred_refine   models/pipeline.py:922-1041
  get_occ_vox_bound   <- batch_get_occupied_idx + LIDF.get_occ_vox_bound
                                            utils/point_utils.py:12-76, models/pipeline.py:162-201
Outputs use the reference's data_dict key names. Pairs are kept RAY-MAJOR (CSR over rays, voxels
ascending inside a ray) instead of the reference's voxel-major nonzero() order; `to_reference_order`
gives the permutation back for code that needs the reference's order.

All compute is in liblidf_hip.so; torch is used for device memory and the current stream only.
"""
import contextlib
import ctypes as C
import os

import torch

from . import _lib
from .decoders import _decoder_struct


def _i32(t, name):
    if t.dtype != torch.int32:
        raise RuntimeError("%s must be int32 (got %s)" % (name, t.dtype))
    return t


def _as_i32(t, name):
    """Index tensors of the reference's data_dict are int64 (torch.nonzero); the kernels take
    int32. An int64 tensor is narrowed here (one cast) instead of being misread as int32."""
    if t is None or t.dtype == torch.int32:
        return t
    if t.dtype == torch.int64:
        return t.to(torch.int32)
    raise RuntimeError("%s must be int32 or int64 (got %s)" % (name, t.dtype))


def _f32(t, name):
    if t.dtype != torch.float32:
        raise RuntimeError("%s must be float32 (got %s)" % (name, t.dtype))
    return t


def _refuse_autograd(fn, alt, tensors, modules):
    """The inference entry points detach everything they touch. Called with autograd recording and an
    input or parameter that requires grad, they would hand back outputs the loss cannot reach — and
    training would silently stall. Refuse loudly instead."""
    if not torch.is_grad_enabled():
        return
    hot = [n for n, t in tensors if t is not None and t.requires_grad]
    hot += ["%s.%s" % (mn, pn) for mn, m in modules if m is not None
            for pn, p in m.named_parameters() if p.requires_grad]
    if hot:
        raise RuntimeError(
            "%s is the inference path (outputs are detached) but autograd is recording and %s "
            "require%s grad: wrap the call in torch.no_grad() for evaluation, or use %s for training"
            % (fn, ", ".join(hot[:3]) + (" ..." if len(hot) > 3 else ""), "s" if len(hot) == 1 else "", alt))


def ray_dirs(fx, fy, cx, cy, h, w):
    """Unit ray direction of every pixel: [bs, h*w, 3] (models/pipeline.py:215-220)."""
    intr = torch.stack((fx.float(), fy.float(), cx.float(), cy.float()), 1).contiguous()
    _lib.require_cuda(intr, names=["intrinsics"])
    bs = intr.shape[0]
    out = torch.empty((bs, h * w, 3), dtype=torch.float32, device=intr.device)
    with torch.cuda.device(intr.device):
        _lib.check(_lib.lib().lidf_ray_dirs_f32(_lib.ptr(intr), bs, h, w, _lib.ptr(out),
                                                _lib.current_stream(intr.device)))
    return out


MASK_DTYPES = {torch.float32: 0, torch.uint8: 1, torch.bool: 1, torch.int32: 2, torch.int64: 3}


def _compact_mask(mask, intr):
    """mark -> scan -> compact of the non-zero pixels of mask [bs,h,w] (lidf_miss_ray_count /
    lidf_miss_ray_fill_f32); intr [bs,4] or None (no ray directions)."""
    if mask.dim() == 4 and mask.shape[1] == 1:
        mask = mask[:, 0]
    if mask.dim() != 3:
        raise RuntimeError("mask must be [bs,h,w] or [bs,1,h,w]")
    if mask.dtype not in MASK_DTYPES:
        raise RuntimeError("mask dtype %s is not supported" % mask.dtype)
    mask = mask.contiguous()
    _lib.require_cuda(mask, intr, names=["mask", "intrinsics"])
    bs, h, w = mask.shape
    if intr is not None and intr.shape[0] != bs:
        raise RuntimeError("fx/fy/cx/cy must have one entry per image")
    dev = mask.device
    L = _lib.lib()
    n = bs * h * w
    wsb = L.lidf_miss_ray_workspace_bytes(n)
    ws = torch.empty((max(wsb, 1),), dtype=torch.uint8, device=dev)
    cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
    mt = MASK_DTYPES[mask.dtype]
    with torch.cuda.device(dev):
        st = _lib.current_stream(dev)
        _lib.check(L.lidf_miss_ray_count(_lib.ptr(mask), mt, n, _lib.ptr(cnt), _lib.ptr(ws), wsb, st))
        R = int(cnt.item())  # host needs R to size the outputs (as nonzero() does)
        i32 = dict(dtype=torch.int32, device=dev)
        i64 = dict(dtype=torch.int64, device=dev)
        out = {
            "ray_bid": torch.empty((R,), **i32), "ray_flat": torch.empty((R,), **i32),
            "ray_pix": torch.empty((R, 2), **i32),
            "miss_bid": torch.empty((R,), **i64), "miss_flat_img_id": torch.empty((R,), **i64),
            "miss_img_ind": torch.empty((R, 2), **i64), "total_miss_sample_num": R,
        }
        if intr is not None:
            out["miss_ray_dir"] = torch.empty((R, 3), dtype=torch.float32, device=dev)
        if R > 0:
            _lib.check(L.lidf_miss_ray_fill_f32(
                _lib.ptr(mask), mt, _lib.ptr(intr), bs, h, w, _lib.ptr(ws), wsb,
                _lib.ptr(out["ray_bid"]), _lib.ptr(out["ray_flat"]), _lib.ptr(out["ray_pix"]),
                _lib.ptr(out.get("miss_ray_dir")), _lib.ptr(out["miss_bid"]),
                _lib.ptr(out["miss_flat_img_id"]), _lib.ptr(out["miss_img_ind"]), st))
    return out


def get_miss_ray(mask, fx, fy, cx, cy):
    """LIDF.get_miss_ray (models/pipeline.py:203-269), eval flavour: the pixels where `mask`
    ([bs,h,w] or [bs,1,h,w]; float, bool/uint8, int32 or int64 — data_dict['pred_mask'] /
    ['corrupt_mask']) is non-zero, in torch.nonzero's (image, pixel) order, with their ray
    directions — mark, scan and compact on the device (lidf_miss_ray_count / _fill_f32).

    Returns the reference's data_dict entries miss_bid [R] i64, miss_flat_img_id [R] i64,
    miss_ray_dir [R,3] f32, miss_img_ind [R,2] i64 (x, y), total_miss_sample_num, plus the int32
    forms the query kernels take: ray_bid, ray_flat [R], ray_pix [R,2]. R == 0 is the reference's
    'no miss ray' early exit (pipeline.py:676, :686-687). The train-only random window of
    pipeline.py:232-254 is a slice [start:start+miss_sample_num] of these outputs per image."""
    intr = torch.stack((fx.float(), fy.float(), cx.float(), cy.float()), 1).contiguous()
    return _compact_mask(mask, intr)


def sample_miss_rays(miss, bs, miss_sample_num):
    """The train-only sub-sampling of LIDF.get_miss_ray (models/pipeline.py:232-254) applied to
    get_miss_ray's output: per image a contiguous window of miss_sample_num rays at a random start
    — drawn with np.random.choice(start_range), the reference's own call, so a run seeded like the
    reference selects the same windows — or every ray of an image that has no more than that.
    Returns a dict with the same keys, sliced (views / index_select: plumbing, no compute)."""
    import numpy as np
    R = miss["total_miss_sample_num"]
    if miss_sample_num == -1 or bs * miss_sample_num >= R:
        return miss
    cnt = torch.bincount(miss["miss_bid"], minlength=bs).cpu().tolist()
    sel, sid = [], 0
    for c in cnt:
        if c > miss_sample_num:
            start = int(np.random.choice(c - miss_sample_num + 1)) + sid
            sel.append((start, start + miss_sample_num))
        else:
            sel.append((sid, sid + c))
        sid += c
    out = {}
    for k, v in miss.items():
        if torch.is_tensor(v):
            out[k] = torch.cat([v[a:b] for a, b in sel], 0).contiguous()
    out["total_miss_sample_num"] = int(sum(b - a for a, b in sel))
    return out


def nonzero_pixels(mask):
    """torch.nonzero(mask.view(bs,-1)) of LIDF.get_valid_points (models/pipeline.py:144-146) with
    the same device compaction: {'bid', 'flat'} int64 [N]."""
    out = _compact_mask(mask, None)
    return {"bid": out["miss_bid"], "flat": out["miss_flat_img_id"]}


def compute_ray_aabb(ray_dir, voxel_bound, ray_bid, voxel_bid, voxel_coord=None, grid_dims=None,
                     batch=None):
    """Compact ray-major ray/voxel intersection list (replaces the dense ray_aabb.forward +
    torch.nonzero of models/pipeline.py:277-285).

    Returns (pair_off [R+1] i32, pair_ray [P] i32, pair_vox [P] i32, pair_t [P,2] f32); inside a
    ray the voxels ascend, which is the reference's order restricted to that ray.  P == 0 is the
    reference's "no intersection pair" early exit (pipeline.py:287-289).

    voxel_coord [V,3] (occ_vox_global_coord), grid_dims (rx,ry,rz) and batch — all three as
    get_occ_vox_bound returns them — select the regular-grid walk (lidf_ray_aabb_grid_*): a ray
    visits only the cells whose per-axis intervals meet instead of every voxel; the result is
    bit-identical. The walk assumes what get_occ_vox_bound produces: every (frame, cell) at most once
    and the list sorted by (frame, x, y, z) — with duplicated or unsorted voxels leave voxel_coord
    out (the voxel-by-voxel kernels make no such assumption)."""
    _lib.require_cuda(ray_dir, voxel_bound, ray_bid, voxel_bid,
                      names=["ray_dir", "voxel_bound", "ray_bid", "voxel_bid"])
    _f32(ray_dir, "ray_dir"), _f32(voxel_bound, "voxel_bound")
    _i32(ray_bid, "ray_bid"), _i32(voxel_bid, "voxel_bid")
    R, V = ray_dir.shape[0], voxel_bound.shape[0]
    if tuple(ray_dir.shape) != (R, 3) or tuple(voxel_bound.shape) != (V, 6):
        raise RuntimeError("ray_dir / voxel_bound must be [R,3] / [V,6]")
    if tuple(ray_bid.shape) != (R,) or tuple(voxel_bid.shape) != (V,):
        raise RuntimeError("ray_bid / voxel_bid must be [R] / [V]")
    dev = ray_dir.device
    L = _lib.lib()
    use_grid = voxel_coord is not None
    if use_grid:
        if grid_dims is None or batch is None:
            raise RuntimeError("voxel_coord needs grid_dims=(rx,ry,rz) and batch")
        voxel_coord = _as_i32(voxel_coord, "voxel_coord").contiguous()
        _lib.require_cuda(voxel_coord, names=["voxel_coord"])
        if tuple(voxel_coord.shape) != (V, 3):
            raise RuntimeError("voxel_coord must be [V,3]")
        rx, ry, rz = (int(v) for v in grid_dims)
        gwb = L.lidf_ray_aabb_grid_workspace_bytes(int(batch), rx, ry, rz)
        if gwb == 0:
            raise RuntimeError("unsupported grid dimensions %s x batch %s" % (grid_dims, batch))
        gws = torch.empty((gwb,), dtype=torch.uint8, device=dev)
    count = torch.empty((max(R, 1),), dtype=torch.int32, device=dev)
    pair_off = torch.zeros((R + 1,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        st = _lib.current_stream(dev)
        if R > 0:
            if use_grid:
                _lib.check(L.lidf_ray_aabb_grid_build_f32(
                    _lib.ptr(voxel_bound), _lib.ptr(voxel_bid), _lib.ptr(voxel_coord), V, int(batch),
                    rx, ry, rz, _lib.ptr(gws), gwb, st))
                _lib.check(L.lidf_ray_aabb_grid_count_f32(
                    _lib.ptr(ray_dir), _lib.ptr(ray_bid), R, int(batch), rx, ry, rz, _lib.ptr(gws), gwb,
                    _lib.ptr(count), st))
            else:
                _lib.check(L.lidf_ray_aabb_count_f32(_lib.ptr(ray_dir), _lib.ptr(voxel_bound),
                                                     _lib.ptr(ray_bid), _lib.ptr(voxel_bid), R, V,
                                                     _lib.ptr(count), st))
            wsb = L.lidf_exclusive_scan_workspace_bytes(R)
            ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
            _lib.check(L.lidf_exclusive_scan_i32(_lib.ptr(count), R, _lib.ptr(pair_off),
                                                 _lib.ptr(ws), wsb, st))
        P = int(pair_off[-1].item())  # host needs P to size the outputs (as nonzero() does)
        pair_ray = torch.empty((P,), dtype=torch.int32, device=dev)
        pair_vox = torch.empty((P,), dtype=torch.int32, device=dev)
        pair_t = torch.empty((P, 2), dtype=torch.float32, device=dev)
        if P > 0:
            if use_grid:
                _lib.check(L.lidf_ray_aabb_grid_fill_f32(
                    _lib.ptr(ray_dir), _lib.ptr(ray_bid), R, int(batch), rx, ry, rz, _lib.ptr(gws), gwb,
                    _lib.ptr(pair_off), _lib.ptr(pair_ray), _lib.ptr(pair_vox), _lib.ptr(pair_t), st))
            else:
                _lib.check(L.lidf_ray_aabb_fill_f32(_lib.ptr(ray_dir), _lib.ptr(voxel_bound),
                                                    _lib.ptr(ray_bid), _lib.ptr(voxel_bid), R, V,
                                                    _lib.ptr(pair_off), _lib.ptr(pair_ray),
                                                    _lib.ptr(pair_vox), _lib.ptr(pair_t), st))
    return pair_off, pair_ray, pair_vox, pair_t


def to_reference_order(pair_ray, pair_vox):
    """Permutation `perm` such that ray-major arrays indexed by `perm` are in the reference's
    voxel-major nonzero() order (models/pipeline.py:283)."""
    key = pair_vox.long() * (int(pair_ray.max().item()) + 1 if pair_ray.numel() else 1) + pair_ray.long()
    return torch.argsort(key, stable=True)


PRECISIONS = {"f32": 0, "f16x3": 1}


def _packed_weights(prob_dec, offset_dec, multires, multires_views, precision, dp, do, dev):
    """Packed weight streams of the fused query, kept per prob_dec (_lib.PACK_CACHE) and re-validated
    on the device by every call (lidf_query_pack_guarded_f32): a fingerprint of the raw parameter
    buffers of both decoders is compared with the one the streams were built from and the pack
    kernels run only when it differs. The host never decides from torch's version counters — those
    miss `p.data.mul_()` / `p.data.copy_()` (EMA, clipping, old-style optimizers). No host sync.
    _lib.freeze_packed(prob_dec) skips the check (pack once, trust until invalidate_packed)."""
    L = _lib.lib()
    key = (multires, multires_views, precision, str(dev))   # both decoders are covered by the fingerprint
    e = _lib.packed_entry(_lib.PACK_CACHE, prob_dec, key, L.lidf_query_pack_bytes(), dev)
    frozen = prob_dec in _lib.FROZEN
    if frozen and e.frozen_ready:
        return e.blob
    with torch.cuda.device(dev):
        _lib.check(L.lidf_query_pack_guarded_f32(
            C.byref(dp), C.byref(do), multires, multires_views, PRECISIONS[precision], _lib.ptr(e.blob),
            e.blob.numel(), _lib.ptr(e.guard), _lib.current_stream(dev)))
    e.frozen_ready = frozen
    return e.blob


def lidf_query(ray_dir, ray_pix, ray_bid, pair_off, pair_ray, pair_vox, pair_t, feat_grid,
               vox_feat, prob_dec, offset_dec, multires=8, multires_views=4, roi_inp_bbox=8,
               offset_range=(0.0, 1.0), part_size=0.25, vox_center=None, pos_rel=False,
               ray_flat=None, depth=None, want_softmax=True, workspace=None, profile_events=None,
               want_rayfeat=False, precision="f32", roi_out_bbox=2, offsets="all"):
    """Fused get_embedding + get_pred (+ depth write-back) through lidf_query_f32.

    offsets="selected" (opt-in, f32, shipped widths): offset_dec runs on the arg-max pair of every ray only —
    everything downstream of get_pred reads pair_pred_pos through max_pair_id alone (models/pipeline.py:
    453-454), so pred_prob_end, the softmax, max_pair_id, pred_pos and the depth map are bit-identical to the
    default while pred_offset / pair_pred_pos are written ONLY at the selected pairs (the other rows are left
    as NaN); about half the matrix work on a real frame.

    Widths other than the shipped configuration (feat_grid channels != 32, roi_out_bbox != 2, vox_feat
    width != 128, decoders with gf_dim != 64): the same function layer by layer on materialised rows
    (generic.query; f32 only).

    precision: "f32" (default) or "f16x3" — the decoders' matrix products evaluated as three
    f16-piece products per term with f32 accumulation (f32-level accuracy, see lidf_hip.h).

    ray_dir [R,3] f32, ray_pix [R,2] i32 (x,y), ray_bid [R] i32, pair_* ray-major (see
    compute_ray_aabb), feat_grid [B,32,h,w] f32 (full_rgb_feat), vox_feat [V,128] f32
    (occ_voxel_feat), prob_dec: IMNet, offset_dec: IEF or IMNet (decoders.py).
    Returns a dict with the reference's data_dict keys (models/pipeline.py:460-466):
    pred_offset [P,1], pred_prob_end [P,1], pair_pred_pos [P,3], pred_prob_end_softmax [P],
    max_pair_id [R] i64, pred_pos [R,3]; `depth` [B,h,w] is updated in place if given."""
    _refuse_autograd("lidf_query", "lidf_query_train",
                     (("feat_grid", feat_grid), ("vox_feat", vox_feat)),
                     (("prob_dec", prob_dec), ("offset_dec", offset_dec)))
    # the reference's index tensors (miss_bid, miss_flat_img_id, miss_img_ind) are int64
    ray_pix, ray_bid, ray_flat = (_as_i32(ray_pix, "ray_pix"), _as_i32(ray_bid, "ray_bid"),
                                  _as_i32(ray_flat, "ray_flat"))
    tensors = [ray_dir, ray_pix, ray_bid, pair_off, pair_ray, pair_vox, pair_t, feat_grid,
               vox_feat, vox_center, ray_flat, depth]
    names = ["ray_dir", "ray_pix", "ray_bid", "pair_off", "pair_ray", "pair_vox", "pair_t",
             "feat_grid", "vox_feat", "vox_center", "ray_flat", "depth"]
    _lib.require_cuda(*tensors, names=names)
    for t, n in ((ray_dir, "ray_dir"), (pair_t, "pair_t"), (feat_grid, "feat_grid"),
                 (vox_feat, "vox_feat"), (vox_center, "vox_center"), (depth, "depth")):
        if t is not None:
            _f32(t, n)
    for t, n in ((pair_off, "pair_off"), (pair_ray, "pair_ray"), (pair_vox, "pair_vox")):
        _i32(t, n)
    if feat_grid.dim() != 4 or vox_feat.dim() != 2:
        raise RuntimeError("feat_grid must be [B,C,h,w], vox_feat [V,F]")
    from .decoders import is_shipped
    shipped = (feat_grid.shape[1] == 32 and vox_feat.shape[1] == 128 and roi_out_bbox == 2
               and is_shipped(prob_dec) and is_shipped(offset_dec))
    E, Ed = 3 + 6 * multires, 3 + 6 * multires_views
    D = vox_feat.shape[1] + feat_grid.shape[1] * roi_out_bbox * roi_out_bbox + 2 * E + Ed
    if prob_dec.inp_dim != D or offset_dec.inp_dim != D:
        raise RuntimeError("decoder inp_dim must be %d for this configuration" % D)
    dev = ray_dir.device
    R, P, V = ray_dir.shape[0], pair_ray.shape[0], vox_feat.shape[0]
    B, _, h, w = feat_grid.shape
    if pair_off.shape[0] != R + 1:
        raise RuntimeError("pair_off must have R+1 entries")
    if depth is not None and ray_flat is None:
        raise RuntimeError("depth needs ray_flat")
    if depth is not None and tuple(depth.shape) != (B, h, w):
        raise RuntimeError("depth must be [B,h,w] = %s (got %s)" % ((B, h, w), tuple(depth.shape)))
    if vox_center is not None and tuple(vox_center.shape) != (V, 3):
        raise RuntimeError("vox_center must be [V,3]")
    if ray_pix.shape != (R, 2) or ray_bid.shape != (R,) or (ray_flat is not None and ray_flat.shape != (R,)):
        raise RuntimeError("ray_pix / ray_bid / ray_flat must be [R,2] / [R] / [R]")
    if tuple(ray_dir.shape) != (R, 3):
        raise RuntimeError("ray_dir must be [R,3]")
    if tuple(pair_vox.shape) != (P,) or tuple(pair_t.shape) != (P, 2) or pair_ray.dim() != 1:
        raise RuntimeError("pair_ray / pair_vox / pair_t must be [P] / [P] / [P,2]")

    if not shipped:
        if precision != "f32":
            raise RuntimeError("precision %r is built for the shipped widths" % precision)
        if offsets != "all":
            raise RuntimeError("offsets='selected' is built for the shipped widths")
        from . import generic
        return generic.query(ray_dir, ray_pix, ray_bid, pair_off, pair_ray, pair_vox, pair_t, feat_grid, vox_feat,
                             prob_dec, offset_dec, multires, multires_views, roi_inp_bbox, roi_out_bbox,
                             offset_range, part_size, vox_center, pos_rel, ray_flat, depth, want_rayfeat)
    f32 = dict(dtype=torch.float32, device=dev)
    out = {
        "pred_offset": torch.empty((P, 1), **f32),
        "pred_prob_end": torch.empty((P, 1), **f32),
        "pair_pred_pos": torch.empty((P, 3), **f32),
        "pred_prob_end_softmax": torch.empty((P,), **f32) if want_softmax else None,
        "max_pair_id": torch.empty((R,), dtype=torch.int64, device=dev),
        "pred_pos": torch.empty((R, 3), **f32),
    }
    L = _lib.lib()
    wsb = L.lidf_query_workspace_bytes(R, V, B * 32 * h * w)
    if workspace is None or workspace.numel() < wsb:
        workspace = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    keep = []
    dp = _decoder_struct(prob_dec, keep)
    do = _decoder_struct(offset_dec, keep)
    q = _lib.LidfQueryArgs()
    q.n_rays, q.ray_dir, q.ray_pix, q.ray_bid = R, ray_dir.data_ptr(), ray_pix.data_ptr(), ray_bid.data_ptr()
    q.ray_flat = ray_flat.data_ptr() if ray_flat is not None else None
    q.n_pairs, q.pair_off = P, pair_off.data_ptr()
    q.pair_ray, q.pair_vox, q.pair_t = pair_ray.data_ptr(), pair_vox.data_ptr(), pair_t.data_ptr()
    q.batch, q.height, q.width, q.feat_grid = B, h, w, feat_grid.data_ptr()
    q.n_vox, q.vox_feat = V, vox_feat.data_ptr()
    q.vox_center = vox_center.data_ptr() if vox_center is not None else None
    q.prob, q.off = C.pointer(dp), C.pointer(do)
    q.multires, q.multires_views, q.roi_inp_bbox = multires, multires_views, roi_inp_bbox
    q.pos_rel = 1 if pos_rel else 0
    q.offset_range0, q.offset_range1 = float(offset_range[0]), float(offset_range[1])
    q.part_size = float(part_size)
    q.pred_offset = out["pred_offset"].data_ptr()
    q.pred_prob = out["pred_prob_end"].data_ptr()
    q.pair_pred_pos = out["pair_pred_pos"].data_ptr()
    q.pred_prob_softmax = out["pred_prob_end_softmax"].data_ptr() if want_softmax else None
    q.max_pair_id = out["max_pair_id"].data_ptr()
    q.pred_pos = out["pred_pos"].data_ptr()
    q.depth = depth.data_ptr() if depth is not None else None
    q.workspace, q.workspace_bytes = workspace.data_ptr(), wsb
    if want_rayfeat:  # keep the per-ray feature rows for stage 2 (lidf_refine(..., rayfeat=...))
        out["rayfeat"] = torch.empty((R, 128 + Ed), **f32)
        q.rayfeat_out = out["rayfeat"].data_ptr()
    if precision not in PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(PRECISIONS)}")
    q.precision = PRECISIONS[precision]
    if offsets not in ("all", "selected"):
        raise ValueError("offsets must be 'all' or 'selected'")
    if offsets == "selected":
        if precision != "f32":
            raise RuntimeError("offsets='selected' runs the f32 kernels")
        q.offsets_selected = 1
        out["pred_offset"].fill_(float("nan"))      # rows that are not written must not look like results
        out["pair_pred_pos"].fill_(float("nan"))
    packed = _packed_weights(prob_dec, offset_dec, multires, multires_views, precision, dp, do, dev)
    q.packed = packed.data_ptr()
    with torch.cuda.device(dev):
        if profile_events is not None:  # (hipEvent_t begin, hipEvent_t end): benchmarks only
            _lib.check(L.lidf_query_profile_f32(C.byref(q), profile_events[0], profile_events[1],
                                                _lib.current_stream(dev)))
        else:
            _lib.check(L.lidf_query_f32(C.byref(q), _lib.current_stream(dev)))
    out["workspace"] = workspace
    return out


def ray_features(feat_grid, ray_dir, ray_pix, ray_bid, roi_inp_bbox=8, multires_views=4):
    """Per-ray [ROIAlign 2x2 of the rgb feature map | embed(dir)] (models/pipeline.py:367-397,
    :947-969): [R, 128 + 3 + 6*multires_views]."""
    _lib.require_cuda(feat_grid, ray_dir, ray_pix, ray_bid,
                      names=["feat_grid", "ray_dir", "ray_pix", "ray_bid"])
    _f32(feat_grid, "feat_grid"), _f32(ray_dir, "ray_dir")
    _i32(ray_pix, "ray_pix"), _i32(ray_bid, "ray_bid")
    B, c, h, w = feat_grid.shape
    if c != 32:
        raise RuntimeError("feat_grid must have 32 channels")
    R = ray_dir.shape[0]
    out = torch.empty((R, 128 + 3 + 6 * multires_views), dtype=torch.float32, device=ray_dir.device)
    L = _lib.lib()
    wsb = L.lidf_ray_features_workspace_bytes(B, h, w, R)   # box-sum image: 4 gathers per channel
    ws = torch.empty((max(wsb, 1),), dtype=torch.uint8, device=ray_dir.device)
    with torch.cuda.device(ray_dir.device):
        _lib.check(L.lidf_ray_features_f32(
            _lib.ptr(feat_grid), B, h, w, _lib.ptr(ray_dir), _lib.ptr(ray_pix), _lib.ptr(ray_bid),
            R, roi_inp_bbox, multires_views, _lib.ptr(out), _lib.ptr(ws), wsb,
            _lib.current_stream(ray_dir.device)))
    return out


def lidf_refine(ray_dir, ray_pix, ray_bid, ray_flat, pred_pos, max_pair_id, pair_vox, voxel_bound,
                voxel_bid, rgb_img, feat_grid, valid_inp, valid_vox, pnet_model, offset_dec,
                forward_times=2, multires=8, multires_views=4, roi_inp_bbox=8,
                offset_range=(-0.2, 0.2), pos_rel=False, pnet_pos_rel=True, rayfeat=None,
                precision="f32", pnet_select=None, profile_events=None, roi_out_bbox=2, grid=None):
    """Stage-2 refinement (RefineNet.forward, models/pipeline.py:1032-1041, eval flavour):
    `forward_times` iterations of get_pred_refine through lidf_refine_f32.

    pred_pos [R,3], max_pair_id [R] i64, pair_vox [P] i32 come from lidf_query; voxel_bound [V,6],
    voxel_bid [V] i32; rgb_img [B,3,h,w]; feat_grid [B,32,h,w] (data_dict['full_rgb_feat']);
    valid_inp [Nv,6] = cat(valid_v_rel_coord, valid_v_rgb), valid_vox [Nv] i32 = revidx;
    pnet_model: pointnet.PointNet2Stage (refine), offset_dec: decoders.IEF/IMNet (D = 334).
    pnet_select: None = refine.use_all_pix True (shipped configs); [R] mask (bool / uint8 / float,
    non-zero = selected) = the mask_type 'all', use_all_pix False branch (pipeline.py:987-996): pass
    inp_zero_mask = 1 - valid_mask at the rays' pixels.
    grid (optional): the voxel list as cells of its grid — a dict with get_occ_vox_bound's entries 'xmin'
    (widened lower corner), 'grid_dims', 'part_size' and 'voxel_coord' [V,3] i32. The end voxel of a ray
    (pcl_aabb + scatter max, pipeline.py:939-944) is then looked up in a cell table — the same ids as
    testing every ray against every voxel (LidfRefineArgs.voxel_coord), O(R) instead of O(R V).
    Returns pred_pos_refine [R,3] and the last iteration's end_voxel_id [R] i32."""
    from .pointnet import pointnet_struct
    _refuse_autograd("lidf_refine", "the modules on their own (PointNet2Stage, IEF and get_embedder are "
                     "differentiable; the fused stage-2 call has no backward)",
                     (("pred_pos", pred_pos), ("feat_grid", feat_grid), ("valid_inp", valid_inp),
                      ("rayfeat", rayfeat)), (("pnet_model", pnet_model), ("offset_dec", offset_dec)))
    ts = [ray_dir, ray_pix, ray_bid, ray_flat, pred_pos, max_pair_id, pair_vox, voxel_bound,
          voxel_bid, rgb_img, feat_grid, valid_inp, valid_vox]
    names = ["ray_dir", "ray_pix", "ray_bid", "ray_flat", "pred_pos", "max_pair_id", "pair_vox",
             "voxel_bound", "voxel_bid", "rgb_img", "feat_grid", "valid_inp", "valid_vox"]
    _lib.require_cuda(*ts, names=names)
    for t, n in ((ray_dir, "ray_dir"), (pred_pos, "pred_pos"), (voxel_bound, "voxel_bound"),
                 (rgb_img, "rgb_img"), (valid_inp, "valid_inp")):
        _f32(t, n)
    for t, n in ((ray_bid, "ray_bid"), (ray_flat, "ray_flat"), (pair_vox, "pair_vox"),
                 (voxel_bid, "voxel_bid"), (valid_vox, "valid_vox")):
        _i32(t, n)
    if max_pair_id.dtype != torch.int64:
        raise RuntimeError("max_pair_id must be int64")
    from .decoders import is_shipped
    from .pointnet import is_shipped as pnet_shipped
    E, Ed = 3 + 6 * multires, 3 + 6 * multires_views
    roi_w = feat_grid.shape[1] * roi_out_bbox * roi_out_bbox
    D = pnet_model.point_lin4.out_features + roi_w + E + Ed
    if offset_dec.inp_dim != D:
        raise RuntimeError("refine offset_dec inp_dim must be %d" % D)
    if not (is_shipped(offset_dec) and pnet_shipped(pnet_model) and feat_grid.shape[1] == 32 and roi_out_bbox == 2):
        # widths other than the shipped ones: step by step through the modules (generic.refine), f32 only
        if precision != "f32":
            raise RuntimeError("precision %r is built for the shipped widths" % precision)
        if valid_inp.dim() != 2 or valid_inp.shape[1] != pnet_model.input_channels:
            raise RuntimeError("valid_inp must be [Nv,%d]" % pnet_model.input_channels)
        from . import generic
        ray_rgb = None
        if rayfeat is not None:
            if tuple(rayfeat.shape) != (ray_dir.shape[0], roi_w + Ed):
                raise RuntimeError("rayfeat must be [R,%d]" % (roi_w + Ed))
            ray_rgb = rayfeat[:, :roi_w]
        return generic.refine(ray_dir, _as_i32(ray_pix, "ray_pix"), ray_bid, ray_flat, pred_pos, max_pair_id,
                              pair_vox, voxel_bound, voxel_bid, rgb_img, feat_grid, valid_inp, valid_vox,
                              pnet_model, offset_dec, forward_times, multires, multires_views, roi_inp_bbox,
                              roi_out_bbox, offset_range, pos_rel, pnet_pos_rel, ray_rgb, pnet_select)
    dev = ray_dir.device
    R, P, V, Nv = ray_dir.shape[0], pair_vox.shape[0], voxel_bound.shape[0], valid_inp.shape[0]
    B, _, h, w = rgb_img.shape
    if (tuple(ray_dir.shape) != (R, 3) or tuple(pred_pos.shape) != (R, 3) or tuple(max_pair_id.shape) != (R,)
            or tuple(ray_bid.shape) != (R,) or tuple(ray_flat.shape) != (R,)):
        raise RuntimeError("ray_dir / pred_pos must be [R,3]; ray_bid / ray_flat / max_pair_id [R]")
    if tuple(voxel_bound.shape) != (V, 6) or tuple(voxel_bid.shape) != (V,):
        raise RuntimeError("voxel_bound / voxel_bid must be [V,6] / [V]")
    if tuple(valid_inp.shape) != (Nv, 6) or tuple(valid_vox.shape) != (Nv,):
        raise RuntimeError("valid_inp / valid_vox must be [Nv,6] / [Nv]")
    if rgb_img.dim() != 4 or rgb_img.shape[1] != 3:
        raise RuntimeError("rgb_img must be [B,3,h,w]")
    if rayfeat is None:
        rayfeat = ray_features(feat_grid, ray_dir, _as_i32(ray_pix, "ray_pix"), ray_bid, roi_inp_bbox,
                               multires_views)
    else:   # the rows lidf_query(want_rayfeat=True) kept
        _lib.require_cuda(rayfeat, names=["rayfeat"])
        _f32(rayfeat, "rayfeat")
        if tuple(rayfeat.shape) != (R, 128 + Ed):
            raise RuntimeError("rayfeat must be [R,%d]" % (128 + Ed))
    L = _lib.lib()
    wsb = L.lidf_refine_workspace_bytes(R, Nv, V)
    ws = torch.empty((max(wsb, 1),), dtype=torch.uint8, device=dev)
    keep = []
    pn = pointnet_struct(pnet_model, keep)
    from .pointnet import packed_pointnet
    keep.append(packed_pointnet(pnet_model, pn, dev))   # weight streams packed once per version
    do = _decoder_struct(offset_dec, keep)
    packed = None
    if precision == "f32":   # the IEF's weight streams: kept per module, re-validated on the device
        key = (multires, multires_views, str(dev))
        e = _lib.packed_entry(_lib.PACK_CACHE_REFINE, offset_dec, key,
                              L.lidf_refine_pack_bytes(multires, multires_views), dev)
        frozen = offset_dec in _lib.FROZEN
        if not (frozen and e.frozen_ready):
            with torch.cuda.device(dev):
                _lib.check(L.lidf_refine_pack_guarded_f32(
                    C.byref(do), multires, multires_views, _lib.ptr(e.blob), e.blob.numel(),
                    _lib.ptr(e.guard), _lib.current_stream(dev)))
            e.frozen_ready = frozen
        packed = e.blob
    cur = pred_pos.contiguous()
    end_voxel = torch.empty((R,), dtype=torch.int32, device=dev)
    # the per-ray part of the decoder's layer 1 is the same in every iteration: formed by the first call
    ray_l1 = torch.empty((R, 256), dtype=torch.float32, device=dev) if precision == "f32" and forward_times > 1 else None
    if pnet_select is not None:
        pnet_select = (pnet_select.reshape(-1) != 0).to(torch.uint8).contiguous()
        _lib.require_cuda(pnet_select, names=["pnet_select"])
        if pnet_select.shape[0] != R:
            raise RuntimeError("pnet_select must have one entry per ray")
    cells = _cell_lookup(grid, V, B, dev, voxel_bound)
    for _ in range(forward_times):
        out = torch.empty((R, 3), dtype=torch.float32, device=dev)
        q = _lib.LidfRefineArgs()
        if cells is not None:
            q.voxel_coord, q.cell_table = cells["coord"].data_ptr(), cells["table"].data_ptr()
            q.grid_res, q.grid_xmin, q.grid_part = cells["res"], cells["xmin"], cells["part"]
            q.cell_table_ready = int(_ > 0)
        q.n_rays, q.ray_dir, q.ray_bid, q.ray_flat = R, ray_dir.data_ptr(), ray_bid.data_ptr(), ray_flat.data_ptr()
        q.pred_pos, q.max_pair_id = cur.data_ptr(), max_pair_id.data_ptr()
        q.pair_vox, q.n_pairs = pair_vox.data_ptr(), P
        q.n_vox, q.voxel_bound, q.voxel_bid = V, voxel_bound.data_ptr(), voxel_bid.data_ptr()
        q.rgb_img, q.batch, q.height, q.width = rgb_img.data_ptr(), B, h, w
        q.rayfeat = rayfeat.data_ptr()
        q.n_valid, q.valid_inp, q.valid_vox = Nv, valid_inp.data_ptr(), valid_vox.data_ptr()
        q.pnet, q.off = C.pointer(pn), C.pointer(do)
        q.multires, q.multires_views = multires, multires_views
        q.pos_rel, q.pnet_pos_rel = int(bool(pos_rel)), int(bool(pnet_pos_rel))
        q.offset_range0, q.offset_range1 = float(offset_range[0]), float(offset_range[1])
        q.pred_pos_out, q.end_voxel_id = out.data_ptr(), end_voxel.data_ptr()
        q.workspace, q.workspace_bytes = ws.data_ptr(), wsb
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(PRECISIONS)}")
        q.precision = PRECISIONS[precision]
        q.pnet_select = pnet_select.data_ptr() if pnet_select is not None else None
        q.packed = packed.data_ptr() if packed is not None else None
        q.ray_l1 = ray_l1.data_ptr() if ray_l1 is not None else None
        q.ray_l1_ready = 1 if (ray_l1 is not None and _ > 0) else 0
        with torch.cuda.device(dev):
            if profile_events is not None:   # benchmarks only: forward_times x 4 hipEvent_t (PointNet, IEF)
                ev = profile_events[_]
                _lib.check(L.lidf_refine_profile_f32(C.byref(q), ev[0], ev[1], ev[2], ev[3],
                                                     _lib.current_stream(dev)))
            else:
                _lib.check(L.lidf_refine_f32(C.byref(q), _lib.current_stream(dev)))
        cur = out
    return cur, end_voxel


def _cell_lookup(grid, V, B, dev, voxel_bound=None):
    """LidfRefineArgs' optional cell table from get_occ_vox_bound's entries (None: every voxel is tested).
    Contract: grid['xmin'] is the WIDENED origin get_occ_vox_bound returns (constants.XMIN - part_size / 2),
    grid['part_size'] the cell edge and grid['voxel_coord'][v] the integer cell of voxel v, so that
    voxel_bound[v, :3] == xmin + voxel_coord[v] * part_size. A grid dict that does not describe voxel_bound
    (e.g. the un-widened xmin) would give other end voxels than the every-voxel test: with LIDF_VALIDATE_GRID=1
    in the environment the relation is checked here (one host sync) and a mismatch raises."""
    if grid is None or V == 0:
        return None
    if voxel_bound is not None and os.environ.get("LIDF_VALIDATE_GRID") == "1":
        _validate_grid(grid, voxel_bound)
    coord = grid["voxel_coord"]
    _lib.require_cuda(coord, names=["grid['voxel_coord']"])
    _i32(coord, "grid['voxel_coord']")
    if tuple(coord.shape) != (V, 3):
        raise RuntimeError("grid['voxel_coord'] must be [V,3]")
    res = [int(v) for v in grid["grid_dims"]]
    xmin = grid["xmin"]
    xmin = [float(v) for v in (xmin.tolist() if torch.is_tensor(xmin) else xmin)]
    if len(res) != 3 or len(xmin) != 3 or min(res) <= 0:
        raise RuntimeError("grid['grid_dims'] / grid['xmin'] must hold three entries")
    return {"coord": coord, "res": (C.c_int32 * 3)(*res), "xmin": (C.c_float * 3)(*xmin),
            "part": float(grid["part_size"]),
            "table": torch.empty((B * res[0] * res[1] * res[2],), dtype=torch.int32, device=dev)}


def _validate_grid(grid, voxel_bound):
    xm = grid["xmin"]
    xm = torch.as_tensor(xm.tolist() if torch.is_tensor(xm) else list(xm), dtype=torch.float32,
                         device=voxel_bound.device)
    part = float(grid["part_size"])
    coord = grid["voxel_coord"].to(torch.float32)
    res = torch.as_tensor([int(v) for v in grid["grid_dims"]], dtype=torch.float32, device=voxel_bound.device)
    lo = xm + coord * part
    bad = ((voxel_bound[:, :3] - lo).abs().max() > 1e-4 * max(part, 1.0)) | (coord.min() < 0) | \
        ((coord >= res).any())
    if bool(bad.item()):
        raise RuntimeError("grid dict does not describe voxel_bound: voxel_bound[:, :3] != xmin + voxel_coord * "
                           "part_size (is grid['xmin'] the widened origin get_occ_vox_bound returns?)")


def get_occ_vox_bound(valid_xyz, valid_bid, batch, xmin=(-1.0, -1.0, 0.0), xmax=(1.0, 1.0, 2.0),
                      res=8):
    """Occupied voxels of the valid points (LIDF.get_occ_vox_bound, models/pipeline.py:162-201, with
    utils/point_utils.py:12-76 batch_get_occupied_idx(overlap=False) inside), on the device.

    valid_xyz [N,3] f32, valid_bid [N] i32. xmin/xmax are constants.XMIN/XMAX and res is
    opt.grid.res: part_size = min(xmax-xmin)/res and the grid is widened by half a voxel on every
    side, exactly as the reference does. Returns the reference's data_dict entries:
    part_size, xmin (widened), revidx [Nv] i64, valid_v_pid [Nv] i64, valid_v_rel_coord [Nv,3],
    occ_vox_bid [V] i64, occ_vox_global_coord [V,3] i64, voxel_bound [V,6]; V == 0 is the
    reference's 'No occupied voxel' early exit."""
    _lib.require_cuda(valid_xyz, valid_bid, names=["valid_xyz", "valid_bid"])
    _f32(valid_xyz, "valid_xyz"), _i32(valid_bid, "valid_bid")
    t32 = lambda v: torch.tensor(v, dtype=torch.float32)  # noqa: E731  (the reference's f32 maths)
    lo, hi = t32(xmin), t32(xmax)
    part_size = float(torch.min(hi - lo).item()) / res
    lo = lo - 0.5 * part_size
    hi = hi + 0.5 * part_size
    r = [int(v) for v in torch.ceil((hi - lo) / part_size).tolist()]
    dev = valid_xyz.device
    N = valid_xyz.shape[0]
    ncell = batch * r[0] * r[1] * r[2]
    occ = torch.empty((ncell, 4), dtype=torch.int32, device=dev)
    vb = torch.empty((ncell, 6), dtype=torch.float32, device=dev)
    pid = torch.empty((N,), dtype=torch.int32, device=dev)
    rev = torch.empty((N,), dtype=torch.int32, device=dev)
    rel = torch.empty((N, 3), dtype=torch.float32, device=dev)
    counts = torch.zeros((2,), dtype=torch.int32, device=dev)
    L = _lib.lib()
    wsb = L.lidf_voxelize_workspace_bytes(N, ncell)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    xm = (C.c_float * 3)(*[float(v) for v in lo.tolist()])
    rr = (C.c_int32 * 3)(*r)
    with torch.cuda.device(dev):
        _lib.check(L.lidf_voxelize_f32(_lib.ptr(valid_xyz), _lib.ptr(valid_bid), N, batch, xm, rr,
                                       C.c_float(part_size), _lib.ptr(occ), _lib.ptr(vb),
                                       _lib.ptr(pid), _lib.ptr(rev), _lib.ptr(rel), _lib.ptr(counts),
                                       _lib.ptr(ws), wsb, _lib.current_stream(dev)))
    V, Nv = [int(v) for v in counts.tolist()]  # host needs the sizes, as torch.unique does
    coord32 = occ[:V, 1:].contiguous()   # int32 form for compute_ray_aabb's grid walk
    occ = occ[:V].long()
    return {
        "part_size": part_size, "xmin": lo.to(dev), "revidx": rev[:Nv].long(),
        "valid_v_pid": pid[:Nv].long(), "valid_v_rel_coord": rel[:Nv],
        "occ_vox_bid": occ[:, 0], "occ_vox_global_coord": occ[:, 1:], "voxel_bound": vb[:V],
        # extra entries for compute_ray_aabb's grid walk: cells per axis of the widened grid, int32 coords
        "grid_dims": tuple(r), "voxel_coord": coord32,
    }


METRIC_NAMES = ("a1", "a2", "a3", "rmse", "rmse_log", "log10", "abs_rel", "mae", "sq_rel")


_METRICS_WS = {}   # (device index, stream) -> zeroed workspace of lidf_depth_metrics_f32


def depth_metrics(pred_depth, gt_depth, seg_mask=None, out_size=(144, 256)):
    """The evaluation statistics of LIDF.compute_loss (models/pipeline.py:577-627, the bs == 1
    branch) on the device: cv2.resize(..., (256, 144), INTER_NEAREST) of the predicted depth, the
    ground-truth depth (non-finite -> 0) and the mask, valid = gt > 0 & mask, then a1/a2/a3, rmse,
    rmse_log, log10, abs_rel, mae, sq_rel. pred_depth / gt_depth [h,w] f32, seg_mask [h,w]
    bool/uint8 or None. out_size=None keeps the source resolution (the bs != 1 statistics on
    already-selected pixels). Returns a dict of 0-d device tensors plus "count"."""
    _lib.require_cuda(pred_depth, gt_depth, names=["pred_depth", "gt_depth"])
    _f32(pred_depth, "pred_depth"), _f32(gt_depth, "gt_depth")
    if pred_depth.dim() != 2 or pred_depth.shape != gt_depth.shape:
        raise RuntimeError("pred_depth and gt_depth must be [h,w] tensors of the same shape")
    h, w = pred_depth.shape
    seg_dtype = 0
    if seg_mask is not None:
        if seg_mask.shape != pred_depth.shape:
            raise RuntimeError("seg_mask must have the shape of the depth maps")
        # the reference's float corrupt_mask is read as it lies (cast per pixel as its astype(np.uint8),
        # models/pipeline.py:588); bool / uint8 masks byte-wise; other integer types are converted
        if seg_mask.dtype == torch.float32:
            seg_dtype = 2
        elif seg_mask.dtype in (torch.bool, torch.uint8):
            seg_dtype = 1
            if seg_mask.dtype == torch.bool:
                seg_mask = seg_mask.view(torch.uint8)
        else:
            seg_mask, seg_dtype = seg_mask.to(torch.uint8), 1
        seg_mask = seg_mask.contiguous()
        _lib.require_cuda(seg_mask, names=["seg_mask"])
    dh, dw = (h, w) if out_size is None else out_size
    dev = pred_depth.device
    out = torch.empty((10,), dtype=torch.float32, device=dev)
    L = _lib.lib()
    # partial sums + arrival ticket of the statistics kernel: zero-filled once per (device, stream); every
    # call leaves the buffer as it found it
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _METRICS_WS.get(key)
    if ws is None:
        ws = _METRICS_WS[key] = torch.zeros((L.lidf_depth_metrics_workspace_bytes(),), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L.lidf_depth_metrics_f32(
            _lib.ptr(pred_depth), _lib.ptr(gt_depth), _lib.ptr(seg_mask) if seg_mask is not None else None,
            seg_dtype, h, w, dh, dw, _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.current_stream(dev)))
    res = {k: out[i] for i, k in enumerate(METRIC_NAMES)}
    res["count"] = out[9]
    return res


# ------------------------------------------------------------------------------------------------
# Training path of the query (gradients to vox_feat, feat_grid and the decoders' parameters)
# ------------------------------------------------------------------------------------------------
class _RayFeaturesFn(torch.autograd.Function):
    """ray_features with RoIAlign backward to the feature map (lidf_ray_features_backward_f32)."""

    @staticmethod
    def forward(ctx, feat_grid, ray_dir, ray_pix, ray_bid, roi_inp_bbox, multires_views):
        fg = feat_grid.detach().contiguous()
        out = ray_features(fg, ray_dir, ray_pix, ray_bid, roi_inp_bbox, multires_views)
        ctx.save_for_backward(ray_pix, ray_bid)
        ctx.shape, ctx.bbox, ctx.lv = tuple(fg.shape), roi_inp_bbox, multires_views
        return out

    @staticmethod
    def backward(ctx, g):
        ray_pix, ray_bid = ctx.saved_tensors
        B, _, h, w = ctx.shape
        g = g.detach().contiguous().float()
        d_feat = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        wsb = B * 129 * h * w * 4    # scratch image of the parked gradients + rays-per-pixel table
        ws = torch.empty((wsb,), dtype=torch.uint8, device=g.device)
        with torch.cuda.device(g.device):
            _lib.check(_lib.lib().lidf_ray_features_backward_f32(
                _lib.ptr(g), _lib.ptr(ray_pix), _lib.ptr(ray_bid), g.shape[0], B, h, w, ctx.bbox,
                ctx.lv, _lib.ptr(d_feat), _lib.ptr(ws), wsb, _lib.current_stream(g.device)))
        return d_feat, None, None, None, None, None


class _BuildRowsFn(torch.autograd.Function):
    """Decoder input rows (LIDF.get_embedding's concat, models/pipeline.py:399-420) with the
    gradient reduced back to vox_feat and the per-ray features (lidf_rows_backward_f32)."""

    @staticmethod
    def forward(ctx, vox_feat, rayfeat, pair_off, pair_ray, pair_vox, pair_t, ray_dir, vox_center,
                pos_rel, multires, multires_views):
        vf, rf = vox_feat.detach().contiguous(), rayfeat.detach().contiguous()
        P = pair_ray.shape[0]
        D = 256 + 2 * (3 + 6 * multires) + 3 + 6 * multires_views
        rows = torch.empty((P, D), dtype=torch.float32, device=vf.device)
        with torch.cuda.device(vf.device):
            _lib.check(_lib.lib().lidf_build_rows_f32(
                _lib.ptr(pair_ray), _lib.ptr(pair_vox), _lib.ptr(pair_t), _lib.ptr(ray_dir),
                _lib.ptr(vox_center) if vox_center is not None else None, 1 if pos_rel else 0,
                _lib.ptr(vf), _lib.ptr(rf), multires, multires_views, P, _lib.ptr(rows),
                _lib.current_stream(vf.device)))
        ctx.save_for_backward(pair_off, pair_vox)
        ctx.dims = (vf.shape[0], rf.shape[0], P, multires, multires_views)
        return rows

    @staticmethod
    def backward(ctx, g):
        pair_off, pair_vox = ctx.saved_tensors
        V, R, P, L, Lv = ctx.dims
        g = g.detach().contiguous().float()
        f32 = dict(dtype=torch.float32, device=g.device)
        d_vox = torch.empty((V, 128), **f32) if ctx.needs_input_grad[0] else None
        d_ray = torch.zeros((R, 128 + 3 + 6 * Lv), **f32) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(g.device):
            _lib.check(_lib.lib().lidf_rows_backward_f32(
                _lib.ptr(g), _lib.ptr(pair_off), _lib.ptr(pair_vox), R, P, V, L, Lv,
                _lib.ptr(d_vox), _lib.ptr(d_ray), _lib.current_stream(g.device)))
        return (d_vox, d_ray) + (None,) * 9


class _QueryDecoderFn(torch.autograd.Function):
    """One decoder of the query in the factorised form (lidf_query_decoder_forward_train_f32 /
    lidf_query_decoder_backward_f32): inputs vox_feat [V,128] and rayfeat [R,128+Ed], no [P,385]
    rows; gradients for both and for every parameter."""

    @staticmethod
    def forward(ctx, mod, vox_feat, rayfeat, pe, pair_off, pair_ray, pair_vox, multires,
                multires_views, *params):
        from . import decoders as _dec
        vf, rf = vox_feat.detach().contiguous(), rayfeat.detach().contiguous()
        P, R, V = pair_ray.shape[0], rf.shape[0], vf.shape[0]
        keep = []
        dec = _dec._decoder_struct(mod, keep)
        a = _lib.LidfQueryTrainArgs()
        a.n_pairs, a.n_rays, a.n_vox = P, R, V
        a.pair_off, a.pair_ray, a.pair_vox = pair_off.data_ptr(), pair_ray.data_ptr(), pair_vox.data_ptr()
        a.pe, a.multires, a.multires_views = pe.data_ptr(), multires, multires_views
        a.vox_feat, a.rayfeat, a.dec = vf.data_ptr(), rf.data_ptr(), C.pointer(dec)
        L = _lib.lib()
        n_pass = int(mod.n_iter) if isinstance(mod, _dec.IEF) else 1
        f32 = dict(dtype=torch.float32, device=vf.device)
        act = torch.empty((max(L.lidf_query_decoder_act_floats(P, R, V, n_pass), 1),), **f32)
        wsb = L.lidf_query_decoder_workspace_bytes(P, R, V)
        ws = torch.empty((max(wsb, 1),), dtype=torch.uint8, device=vf.device)
        out = torch.empty((P, 1), **f32)
        with torch.cuda.device(vf.device):
            _lib.check(L.lidf_query_decoder_forward_train_f32(
                C.byref(a), _lib.ptr(out), _lib.ptr(act), _lib.ptr(ws), wsb,
                _lib.current_stream(vf.device)))
        ctx.mod, ctx.ws = mod, ws
        ctx.cfg = (multires, multires_views, wsb)
        ctx.names = [k for k in _dec._PARAM_ORDER if _dec._has(mod, k)]
        ctx.save_for_backward(vf, rf, pe, pair_off, pair_ray, pair_vox, act, *params)
        return out

    @staticmethod
    def backward(ctx, g_out):
        from . import decoders as _dec
        mod, ws = ctx.mod, ctx.ws
        vf, rf, pe, pair_off, pair_ray, pair_vox, act = ctx.saved_tensors[:7]
        saved = dict(zip(ctx.names, ctx.saved_tensors[7:]))
        multires, multires_views, wsb = ctx.cfg
        keep = []
        dec = _dec._decoder_struct(mod, keep, saved)
        a = _lib.LidfQueryTrainArgs()
        a.n_pairs, a.n_rays, a.n_vox = pair_ray.shape[0], rf.shape[0], vf.shape[0]
        a.pair_off, a.pair_ray, a.pair_vox = pair_off.data_ptr(), pair_ray.data_ptr(), pair_vox.data_ptr()
        a.pe, a.multires, a.multires_views = pe.data_ptr(), multires, multires_views
        a.vox_feat, a.rayfeat, a.dec = vf.data_ptr(), rf.data_ptr(), C.pointer(dec)
        f32 = dict(dtype=torch.float32, device=vf.device)
        g = g_out.detach().reshape(-1).contiguous().float()
        names = ctx.names
        gt = {k: torch.empty_like(saved[k], **f32).contiguous() for k in names}
        gs = _lib.LidfDecoderGrads()
        for field, k in zip(("w1", "b1", "w2", "b2", "w3", "b3", "w4", "b4", "wenc", "benc"), _dec._PARAM_ORDER):
            setattr(gs, field, gt[k].data_ptr() if k in gt else None)
        d_vox = torch.empty_like(vf) if ctx.needs_input_grad[1] else None
        d_ray = torch.empty_like(rf) if ctx.needs_input_grad[2] else None
        with torch.cuda.device(vf.device):
            _lib.check(_lib.lib().lidf_query_decoder_backward_f32(
                C.byref(a), _lib.ptr(act), _lib.ptr(g), _lib.ptr(d_vox), _lib.ptr(d_ray), 0,
                C.byref(gs), _lib.ptr(ws), wsb, _lib.current_stream(vf.device)))
        return (None, d_vox, d_ray) + (None,) * 6 + tuple(
            gt[k] if ctx.needs_input_grad[9 + i] else None for i, k in enumerate(names))


class _QueryDecodersFn(torch.autograd.Function):
    """Both decoders of the query in the factorised form: forward = ONE launch of the per-point
    kernel with the activations kept (lidf_query_forward_train_f32; the positional encodings are
    formed in registers), backward = lidf_query_decoder_backward_f32 per decoder, the second one
    adding into the first one's d vox_feat / d rayfeat."""

    @staticmethod
    def forward(ctx, prob_mod, off_mod, vox_feat, rayfeat, pe, pair_off, pair_ray, pair_vox, pair_t,
                ray_dir, vox_center, pos_rel, multires, multires_views, n_prob, *params):
        from . import decoders as _dec
        vf, rf = vox_feat.detach().contiguous(), rayfeat.detach().contiguous()
        P, R, V = pair_ray.shape[0], rf.shape[0], vf.shape[0]
        keep = []
        dp, do = _dec._decoder_struct(prob_mod, keep), _dec._decoder_struct(off_mod, keep)
        a = _lib.LidfQueryTrainArgs()
        a.n_pairs, a.n_rays, a.n_vox = P, R, V
        a.pair_off, a.pair_ray, a.pair_vox = pair_off.data_ptr(), pair_ray.data_ptr(), pair_vox.data_ptr()
        a.pe, a.multires, a.multires_views = pe.data_ptr(), multires, multires_views
        a.vox_feat, a.rayfeat, a.dec = vf.data_ptr(), rf.data_ptr(), C.pointer(dp)
        L = _lib.lib()
        f32 = dict(dtype=torch.float32, device=vf.device)
        passes = [int(m.n_iter) if isinstance(m, _dec.IEF) else 1 for m in (prob_mod, off_mod)]
        acts = [torch.empty((max(L.lidf_query_decoder_act_floats(P, R, V, n), 1),), **f32) for n in passes]
        wsb = L.lidf_query_forward_train_workspace_bytes(R, V)
        ws = torch.empty((max(wsb, 1),), dtype=torch.uint8, device=vf.device)
        outs = [torch.empty((P, 1), **f32) for _ in range(2)]
        with torch.cuda.device(vf.device):
            _lib.check(L.lidf_query_forward_train_f32(
                C.byref(a), C.byref(do), _lib.ptr(pair_t), _lib.ptr(ray_dir),
                _lib.ptr(vox_center) if vox_center is not None else None, 1 if pos_rel else 0,
                _lib.ptr(outs[0]), _lib.ptr(outs[1]), _lib.ptr(acts[0]), _lib.ptr(acts[1]), _lib.ptr(ws), wsb,
                _lib.current_stream(vf.device)))
        ctx.mods = (prob_mod, off_mod)
        ctx.cfg = (multires, multires_views, n_prob)
        ctx.names = [[k for k in _dec._PARAM_ORDER if _dec._has(m, k)] for m in (prob_mod, off_mod)]
        ctx.save_for_backward(vf, rf, pe, pair_off, pair_ray, pair_vox, acts[0], acts[1], *params)
        return outs[0], outs[1]

    @staticmethod
    def backward(ctx, g_prob, g_off):
        from . import decoders as _dec
        vf, rf, pe, pair_off, pair_ray, pair_vox, act_p, act_o = ctx.saved_tensors[:8]
        params = ctx.saved_tensors[8:]
        multires, multires_views, n_prob = ctx.cfg
        P, R, V = pair_ray.shape[0], rf.shape[0], vf.shape[0]
        f32 = dict(dtype=torch.float32, device=vf.device)
        L = _lib.lib()
        wsb = L.lidf_query_decoder_workspace_bytes(P, R, V)
        ws = torch.empty((max(wsb, 1),), dtype=torch.uint8, device=vf.device)
        d_vox = torch.empty_like(vf) if ctx.needs_input_grad[2] else None
        d_ray = torch.empty_like(rf) if ctx.needs_input_grad[3] else None
        g_pred = g_pred.contiguous().float() if g_pred is not None else None
        grads_out = []
        for i, (mod, act, g_out, names, ps) in enumerate((
                (ctx.mods[0], act_p, g_prob, ctx.names[0], params[:n_prob]),
                (ctx.mods[1], act_o, g_off, ctx.names[1], params[n_prob:]))):
            saved = dict(zip(names, ps))
            keep = []
            dec = _dec._decoder_struct(mod, keep, saved)
            a = _lib.LidfQueryTrainArgs()
            a.n_pairs, a.n_rays, a.n_vox = P, R, V
            a.pair_off, a.pair_ray, a.pair_vox = pair_off.data_ptr(), pair_ray.data_ptr(), pair_vox.data_ptr()
            a.pe, a.multires, a.multires_views = pe.data_ptr(), multires, multires_views
            a.vox_feat, a.rayfeat, a.dec = vf.data_ptr(), rf.data_ptr(), C.pointer(dec)
            g = (g_out if g_out is not None else torch.zeros((P, 1), **f32)).detach().reshape(-1).contiguous().float()
            gt = {k: torch.empty_like(saved[k], **f32).contiguous() for k in names}
            gs = _lib.LidfDecoderGrads()
            for field, k in zip(("w1", "b1", "w2", "b2", "w3", "b3", "w4", "b4", "wenc", "benc"), _dec._PARAM_ORDER):
                setattr(gs, field, gt[k].data_ptr() if k in gt else None)
            with torch.cuda.device(vf.device):
                _lib.check(L.lidf_query_decoder_backward_f32(
                    C.byref(a), _lib.ptr(act), _lib.ptr(g), _lib.ptr(d_vox), _lib.ptr(d_ray), 1 if i else 0,
                    C.byref(gs), _lib.ptr(ws), wsb, _lib.current_stream(vf.device)))
            grads_out += [gt[k] for k in names]
        base = 15
        return (None, None, d_vox, d_ray) + (None,) * 11 + tuple(
            g if ctx.needs_input_grad[base + i] else None for i, g in enumerate(grads_out))


def _query_decoders(prob_mod, off_mod, vox_feat, rayfeat, pe, pair_off, pair_ray, pair_vox, pair_t,
                    ray_dir, vox_center, pos_rel, multires, multires_views):
    from . import decoders as _dec
    _dec._check_supported(prob_mod), _dec._check_supported(off_mod)
    pp = [_dec._get(prob_mod, k) for k in _dec._PARAM_ORDER if _dec._has(prob_mod, k)]
    po = [_dec._get(off_mod, k) for k in _dec._PARAM_ORDER if _dec._has(off_mod, k)]
    return _QueryDecodersFn.apply(prob_mod, off_mod, vox_feat, rayfeat, pe, pair_off, pair_ray, pair_vox,
                                  pair_t, ray_dir, vox_center, pos_rel, multires, multires_views, len(pp),
                                  *pp, *po)


def _query_decoder(mod, vox_feat, rayfeat, pe, pair_off, pair_ray, pair_vox, multires, multires_views):
    from . import decoders as _dec
    _dec._check_supported(mod)
    return _QueryDecoderFn.apply(mod, vox_feat, rayfeat, pe, pair_off, pair_ray, pair_vox, multires,
                                 multires_views, *[_dec._get(mod, k) for k in _dec._PARAM_ORDER if _dec._has(mod, k)])


class _QueryTailFn(torch.autograd.Function):
    """pair_pred_pos, per-ray softmax / arg-max / select of LIDF.get_pred (models/pipeline.py:437-454)
    through lidf_query_tail_f32, adjoint lidf_query_tail_backward_f32. The logits are detached as in
    the reference (:442): the only gradient is d pred_offset."""

    @staticmethod
    def forward(ctx, pred_offset, pred_prob, pair_off, pair_ray, pair_t, ray_dir, r0, r1, part, mid_in):
        dev = ray_dir.device
        R, P = ray_dir.shape[0], pair_ray.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        off = pred_offset.detach().reshape(-1).contiguous()
        prob = pred_prob.detach().reshape(-1).contiguous()
        pos = torch.empty((P, 3), **f32)
        sm = torch.empty((P,), **f32)
        mid = torch.empty((R,), dtype=torch.int64, device=dev)
        pred = torch.empty((R, 3), **f32)
        if mid_in is not None:
            mid_in = mid_in.to(torch.int64).contiguous()
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().lidf_query_tail_f32(
                _lib.ptr(off), _lib.ptr(prob), _lib.ptr(pair_off), _lib.ptr(pair_ray), _lib.ptr(pair_t),
                _lib.ptr(ray_dir), R, P, r0, r1, part, _lib.ptr(mid_in), _lib.ptr(pos), _lib.ptr(sm),
                _lib.ptr(mid), _lib.ptr(pred), _lib.current_stream(dev)))
        sel = mid_in if mid_in is not None else mid
        ctx.save_for_backward(sel, pair_ray, ray_dir)
        ctx.cfg = (r0, r1, part, R, P, tuple(pred_offset.shape))
        # (exactly the returned tensors are marked; with a caller-supplied selection the output is a copy, never
        # the caller's own tensor aliased as an output of the node)
        mid_out = mid if mid_in is None else sel.clone()
        ctx.mark_non_differentiable(sm, mid_out)
        return pos, sm, mid_out, pred

    @staticmethod
    def backward(ctx, g_pos, g_sm, g_mid, g_pred):
        sel, pair_ray, ray_dir = ctx.saved_tensors
        r0, r1, part, R, P, shape = ctx.cfg
        d_off = torch.empty((P,), dtype=torch.float32, device=ray_dir.device)
        g_pos = g_pos.contiguous().float() if g_pos is not None else None
        g_pred = g_pred.contiguous().float() if g_pred is not None else None
        with torch.cuda.device(ray_dir.device):
            _lib.check(_lib.lib().lidf_query_tail_backward_f32(
                _lib.ptr(g_pos), _lib.ptr(g_pred), _lib.ptr(sel), _lib.ptr(pair_ray), _lib.ptr(ray_dir),
                R, P, r0, r1, part, _lib.ptr(d_off), _lib.current_stream(ray_dir.device)))
        return (d_off.reshape(shape),) + (None,) * 9


_TRAIN_SIDE = {}


def _train_side_stream(dev):
    """One side stream per device for the training steps' independent branches (created on first use)."""
    s = _TRAIN_SIDE.get(dev.index)
    if s is None:
        s = _TRAIN_SIDE[dev.index] = torch.cuda.Stream(dev)
    return s


class _QueryTrainFn(torch.autograd.Function):
    """get_pred of the training step as ONE autograd node: both decoders (lidf_query_forward_train_f32, one launch
    of the per-point kernel with the activations kept) and the per-pair / per-ray tail (lidf_query_tail_f32).
    The node sees which of its outputs the loss used. The reference's losses reach offset_dec through
    pred_pos = pair_pred_pos[max_pair_id] alone (models/pipeline.py:437-454, 468-476): when neither pred_offset
    nor pair_pred_pos received a gradient, dL/d pred_offset is non-zero at the selected pair of every ray only,
    and offset_dec's backward runs over those R rows (lidf_query_decoder_backward_rows_f32) — the same gradients,
    an eighth of the rows at 8 pairs per ray. A loss that touches pred_offset or pair_pred_pos takes the dense
    backward (lidf_query_tail_backward_f32 + lidf_query_decoder_backward_f32)."""

    @staticmethod
    def forward(ctx, prob_mod, off_mod, vox_feat, rayfeat, pe, pair_off, pair_ray, pair_vox, pair_t,
                ray_dir, vox_center, pos_rel, multires, multires_views, r0, r1, part, mid_in, selected, n_prob,
                *params):
        from . import decoders as _dec
        vf, rf = vox_feat.detach().contiguous(), rayfeat.detach().contiguous()
        dev = vf.device
        P, R, V = pair_ray.shape[0], rf.shape[0], vf.shape[0]
        if selected:
            return _QueryTrainFn._forward_selected(ctx, prob_mod, off_mod, vf, rf, pe, pair_off, pair_ray, pair_vox,
                                                   pair_t, ray_dir, vox_center, pos_rel, multires, multires_views,
                                                   r0, r1, part, mid_in, n_prob, params)
        keep = []
        dp, do = _dec._decoder_struct(prob_mod, keep), _dec._decoder_struct(off_mod, keep)
        a = _lib.LidfQueryTrainArgs()
        a.n_pairs, a.n_rays, a.n_vox = P, R, V
        a.pair_off, a.pair_ray, a.pair_vox = pair_off.data_ptr(), pair_ray.data_ptr(), pair_vox.data_ptr()
        a.pe, a.multires, a.multires_views = pe.data_ptr(), multires, multires_views
        a.vox_feat, a.rayfeat, a.dec = vf.data_ptr(), rf.data_ptr(), C.pointer(dp)
        L = _lib.lib()
        f32 = dict(dtype=torch.float32, device=dev)
        passes = [int(m.n_iter) if isinstance(m, _dec.IEF) else 1 for m in (prob_mod, off_mod)]
        acts = [torch.empty((max(L.lidf_query_decoder_act_floats(P, R, V, n), 1),), **f32) for n in passes]
        wsb = L.lidf_query_forward_train_workspace_bytes(R, V)
        ws = torch.empty((max(wsb, 1),), dtype=torch.uint8, device=dev)
        prob, off = torch.empty((P, 1), **f32), torch.empty((P, 1), **f32)
        pos = torch.empty((P, 3), **f32)
        sm = torch.empty((P,), **f32)
        mid = torch.empty((R,), dtype=torch.int64, device=dev)
        pred = torch.empty((R, 3), **f32)
        if mid_in is not None:
            mid_in = mid_in.to(torch.int64).contiguous()
        with torch.cuda.device(dev):
            st = _lib.current_stream(dev)
            _lib.check(L.lidf_query_forward_train_f32(
                C.byref(a), C.byref(do), _lib.ptr(pair_t), _lib.ptr(ray_dir),
                _lib.ptr(vox_center) if vox_center is not None else None, 1 if pos_rel else 0,
                _lib.ptr(prob), _lib.ptr(off), _lib.ptr(acts[0]), _lib.ptr(acts[1]), _lib.ptr(ws), wsb, st))
            _lib.check(L.lidf_query_tail_f32(
                _lib.ptr(off), _lib.ptr(prob), _lib.ptr(pair_off), _lib.ptr(pair_ray), _lib.ptr(pair_t),
                _lib.ptr(ray_dir), R, P, r0, r1, part, _lib.ptr(mid_in), _lib.ptr(pos), _lib.ptr(sm),
                _lib.ptr(mid), _lib.ptr(pred), st))
        sel = mid_in if mid_in is not None else mid
        ctx.mods = (prob_mod, off_mod)
        ctx.cfg = (multires, multires_views, n_prob, r0, r1, part, passes, False)
        ctx.names = [[k for k in _dec._PARAM_ORDER if _dec._has(m, k)] for m in (prob_mod, off_mod)]
        ctx.save_for_backward(vf, rf, pe, pair_off, pair_ray, pair_vox, ray_dir, sel, acts[0], acts[1], *params)
        ctx.set_materialize_grads(False)
        mid_out = mid if mid_in is None else sel.clone()   # (a copy: never the caller's tensor as an output)
        ctx.mark_non_differentiable(sm, mid_out)
        return prob, off, pos, sm, mid_out, pred

    @staticmethod
    def _forward_selected(ctx, prob_mod, off_mod, vf, rf, pe, pair_off, pair_ray, pair_vox, pair_t, ray_dir,
                          vox_center, pos_rel, multires, multires_views, r0, r1, part, mid_in, n_prob, params):
        """offsets="selected": offset_dec runs on the selected pair of every ray only, forward and backward
        (lidf_query_forward_train_selected_f32); pred_offset / pair_pred_pos are zero at every other pair."""
        from . import decoders as _dec
        dev = vf.device
        P, R, V = pair_ray.shape[0], rf.shape[0], vf.shape[0]
        keep = []
        dp, do = _dec._decoder_struct(prob_mod, keep), _dec._decoder_struct(off_mod, keep)
        a = _lib.LidfQueryTrainArgs()
        a.n_pairs, a.n_rays, a.n_vox = P, R, V
        a.pair_off, a.pair_ray, a.pair_vox = pair_off.data_ptr(), pair_ray.data_ptr(), pair_vox.data_ptr()
        a.pe, a.multires, a.multires_views = pe.data_ptr(), multires, multires_views
        a.vox_feat, a.rayfeat, a.dec = vf.data_ptr(), rf.data_ptr(), C.pointer(dp)
        L = _lib.lib()
        f32 = dict(dtype=torch.float32, device=dev)
        passes = [int(m.n_iter) if isinstance(m, _dec.IEF) else 1 for m in (prob_mod, off_mod)]
        act_p = torch.empty((max(L.lidf_query_decoder_act_floats(P, R, V, passes[0]), 1),), **f32)
        act_o = torch.empty((max(L.lidf_query_decoder_act_floats(R, R, V, passes[1]), 1),), **f32)
        wsb = L.lidf_query_forward_train_workspace_bytes(R, V)
        ws = torch.empty((max(wsb, 1),), dtype=torch.uint8, device=dev)
        prob = torch.empty((P, 1), **f32)
        off, pos = torch.zeros((P, 1), **f32), torch.zeros((P, 3), **f32)
        sm = torch.empty((P,), **f32)
        mid = torch.empty((R,), dtype=torch.int64, device=dev)
        pred = torch.empty((R, 3), **f32)
        if mid_in is not None:
            mid_in = mid_in.to(torch.int64).contiguous()
        with torch.cuda.device(dev):
            _lib.check(L.lidf_query_forward_train_selected_f32(
                C.byref(a), C.byref(do), _lib.ptr(pair_t), _lib.ptr(ray_dir),
                _lib.ptr(vox_center) if vox_center is not None else None, 1 if pos_rel else 0, r0, r1, part,
                _lib.ptr(mid_in), _lib.ptr(prob), _lib.ptr(sm), _lib.ptr(mid), _lib.ptr(off), _lib.ptr(pos),
                _lib.ptr(pred), _lib.ptr(act_p), _lib.ptr(act_o), _lib.ptr(ws), wsb, _lib.current_stream(dev)))
        sel = mid_in if mid_in is not None else mid
        ctx.mods = (prob_mod, off_mod)
        ctx.cfg = (multires, multires_views, n_prob, r0, r1, part, passes, True)
        ctx.names = [[k for k in _dec._PARAM_ORDER if _dec._has(m, k)] for m in (prob_mod, off_mod)]
        ctx.save_for_backward(vf, rf, pe, pair_off, pair_ray, pair_vox, ray_dir, sel, act_p, act_o, *params)
        ctx.set_materialize_grads(False)
        mid_out = mid if mid_in is None else sel.clone()   # (a copy: never the caller's tensor as an output)
        ctx.mark_non_differentiable(sm, mid_out)
        return prob, off, pos, sm, mid_out, pred

    @staticmethod
    def backward(ctx, g_prob, g_off, g_pos, g_sm, g_mid, g_pred):
        from . import decoders as _dec
        vf, rf, pe, pair_off, pair_ray, pair_vox, ray_dir, sel, act_p, act_o = ctx.saved_tensors[:10]
        params = ctx.saved_tensors[10:]
        multires, multires_views, n_prob, r0, r1, part, passes, selected = ctx.cfg
        P, R, V = pair_ray.shape[0], rf.shape[0], vf.shape[0]
        dev = vf.device
        f32 = dict(dtype=torch.float32, device=dev)
        L = _lib.lib()
        # (autograd hands expanded stride-0 gradients to a node for losses such as pred_pos.sum(): the kernels
        # read [R,3] / [P,3] / [P,1] floats through raw pointers, so every incoming gradient is made dense first)
        g_pred = g_pred.contiguous().float() if g_pred is not None else None
        g_pos = g_pos.contiguous().float() if g_pos is not None else None
        g_off = g_off.contiguous().float() if g_off is not None else None
        g_prob = g_prob.contiguous().float() if g_prob is not None else None
        # offset_dec is reached through pred_pos alone — or ran on the selected pairs only in the forward
        rows_only = selected or (g_off is None and g_pos is None)
        g_off_rows = None
        if selected and P > 0 and (g_off is not None or g_pos is not None):
            # (gradients the loss put on the selected pairs' own slots of pair_pred_pos / pred_offset; the other
            # entries of those outputs are constants)
            has = (sel >= 0) & (sel < P)
            at = sel.clamp(0, P - 1)
            if g_pos is not None:
                extra = torch.where(has[:, None], g_pos.reshape(P, 3)[at], torch.zeros((), **f32))
                g_pred = extra if g_pred is None else g_pred + extra
            if g_off is not None:
                g_off_rows = torch.where(has, g_off.reshape(-1)[at], torch.zeros((), **f32)).contiguous().float()
        wsb = L.lidf_query_decoder_workspace_bytes(P, R, V)
        if rows_only:
            wsb = max(wsb, L.lidf_query_decoder_rows_workspace_bytes(R, V, multires, passes[1]))
        ws = torch.empty((max(wsb, 1),), dtype=torch.uint8, device=dev)
        d_vox = torch.empty_like(vf) if ctx.needs_input_grad[2] else None
        d_ray = torch.empty_like(rf) if ctx.needs_input_grad[3] else None
        # offset_dec's backward over the selected rows is ~40 launches on R rows (76,800 where prob_dec's walk
        # 614,400): latency, not work. The two backwards are independent until their input gradients are added, so
        # that one runs on a side stream beside prob_dec's (its own workspace and d_vox / d_ray, added at the join;
        # every sum keeps its fixed order). LIDF_TRAIN_STREAMS=1 keeps everything on the caller's stream.
        two = (rows_only and P > 0 and os.environ.get("LIDF_TRAIN_STREAMS", "2") != "1"
               and not torch.cuda.is_current_stream_capturing())
        side = _train_side_stream(dev) if two else None
        ws2 = d_vox2 = d_ray2 = None
        if two:
            ws2 = torch.empty((max(wsb, 1),), dtype=torch.uint8, device=dev)
            d_vox2 = torch.empty_like(vf) if d_vox is not None else None
            d_ray2 = torch.empty_like(rf) if d_ray is not None else None
            side.wait_stream(torch.cuda.current_stream(dev))
        grads_out = [None, None]
        with torch.cuda.device(dev):
            st = _lib.current_stream(dev)
            order = ((1, ctx.mods[1], act_o, ctx.names[1], params[n_prob:]),
                     (0, ctx.mods[0], act_p, ctx.names[0], params[:n_prob]))
            if not two:
                order = order[::-1]
            for i, mod, act, names, ps in order:
                saved = dict(zip(names, ps))
                keep = []
                dec = _dec._decoder_struct(mod, keep, saved)
                a = _lib.LidfQueryTrainArgs()
                a.n_pairs, a.n_rays, a.n_vox = P, R, V
                a.pair_off, a.pair_ray, a.pair_vox = pair_off.data_ptr(), pair_ray.data_ptr(), pair_vox.data_ptr()
                a.pe, a.multires, a.multires_views = pe.data_ptr(), multires, multires_views
                a.vox_feat, a.rayfeat, a.dec = vf.data_ptr(), rf.data_ptr(), C.pointer(dec)
                gt = {k: torch.empty_like(saved[k], **f32).contiguous() for k in names}
                gs = _lib.LidfDecoderGrads()
                for field, k in zip(("w1", "b1", "w2", "b2", "w3", "b3", "w4", "b4", "wenc", "benc"), _dec._PARAM_ORDER):
                    setattr(gs, field, gt[k].data_ptr() if k in gt else None)
                if i == 1 and rows_only:
                    with torch.cuda.stream(side) if two else contextlib.nullcontext():
                        _lib.check(L.lidf_query_decoder_backward_rows_f32(
                            C.byref(a), _lib.ptr(act), 1 if selected else 0, _lib.ptr(sel), _lib.ptr(g_pred),
                            _lib.ptr(g_off_rows), _lib.ptr(ray_dir), float((r1 - r0) * 1.7320508075688772 * part),
                            _lib.ptr(d_vox2 if two else d_vox), _lib.ptr(d_ray2 if two else d_ray), 0 if two else 1,
                            C.byref(gs), _lib.ptr(ws2 if two else ws), wsb,
                            _lib.current_stream(dev) if two else st))
                else:
                    if i == 0:
                        g = g_prob if g_prob is not None else torch.zeros((P, 1), **f32)
                    else:   # the tail's adjoint for every pair, + what the loss put on pred_offset itself
                        g = torch.empty((P,), **f32)
                        gpos = g_pos.contiguous().float() if g_pos is not None else None
                        _lib.check(L.lidf_query_tail_backward_f32(
                            _lib.ptr(gpos), _lib.ptr(g_pred), _lib.ptr(sel), _lib.ptr(pair_ray), _lib.ptr(ray_dir),
                            R, P, r0, r1, part, _lib.ptr(g), st))
                        if g_off is not None:
                            g = g + g_off.reshape(-1)
                    g = g.detach().reshape(-1).contiguous().float()
                    _lib.check(L.lidf_query_decoder_backward_f32(
                        C.byref(a), _lib.ptr(act), _lib.ptr(g), _lib.ptr(d_vox), _lib.ptr(d_ray), 1 if i else 0,
                        C.byref(gs), _lib.ptr(ws), wsb, st))
                grads_out[i] = [gt[k] for k in names]
        if two:   # join: offset_dec's share of the input gradients
            torch.cuda.current_stream(dev).wait_stream(side)
            if d_vox is not None:
                d_vox += d_vox2
            if d_ray is not None:
                d_ray += d_ray2
        grads_out = grads_out[0] + grads_out[1]
        base = 20
        return (None, None, d_vox, d_ray) + (None,) * 16 + tuple(
            g if ctx.needs_input_grad[base + i] else None for i, g in enumerate(grads_out))


def lidf_query_train(ray_dir, ray_pix, ray_bid, pair_off, pair_ray, pair_vox, pair_t, feat_grid,
                     vox_feat, prob_dec, offset_dec, multires=8, multires_views=4, roi_inp_bbox=8,
                     offset_range=(0.0, 1.0), part_size=0.25, vox_center=None, pos_rel=False,
                     factorised=True, max_pair_id=None, offsets="all"):
    """Differentiable get_embedding + get_pred (models/pipeline.py:338-466) for training
    (train_lidf.py:393-396): gradients reach feat_grid (through RoIAlign), vox_feat and every
    decoder parameter, all through liblidf_hip — ROI pooling, the decoder input rows, the decoders'
    forward that keeps activations and their backward, and the per-pair / per-ray tail
    (pair_pred_pos, softmax over a ray of the detached logits, arg-max, select) with its adjoint.
    max_pair_id [R] int64 overrides the arg-max selection: the reference selects by ground-truth
    labels while epoch < maxpool_label_epo (pipeline.py:444-446).
    Same arguments as lidf_query; the decoders must be in autograd mode (parameters requiring grad).
    factorised=True (default) keeps the layer-1 rewrite of the inference kernel — per-voxel and
    per-ray partial products, only the positional encodings as per-pair rows, the layer-1 gradient
    reduced per voxel / per ray before it meets a weight; factorised=False materialises the
    reference's [P,385] decoder input rows (same results, more memory and work).
    offsets="selected" (opt-in, as in lidf_query): offset_dec runs on the selected pair of every ray only, in the
    forward as well — pred_offset / pair_pred_pos are zero at every other pair, which nothing in the reference reads
    (pred_offset is a local of get_pred, pair_pred_pos is stored at pipeline.py:461 and never used); pred_pos, the
    logits and every gradient of a loss on them are those of offsets="all".
    Returns pred_offset, pred_prob_end [P,1], pair_pred_pos [P,3], pred_prob_end_softmax [P],
    max_pair_id [R] (P for an empty ray) and pred_pos [R,3]."""
    if offsets not in ("all", "selected"):
        raise ValueError("offsets must be 'all' or 'selected'")
    if offsets == "selected" and not factorised:
        raise RuntimeError("offsets='selected' needs the factorised path")
    ray_pix, ray_bid = _as_i32(ray_pix, "ray_pix"), _as_i32(ray_bid, "ray_bid")   # int64 accepted
    _lib.require_cuda(ray_dir, ray_pix, ray_bid, pair_off, pair_ray, pair_vox, pair_t, feat_grid,
                      vox_feat, vox_center,
                      names=["ray_dir", "ray_pix", "ray_bid", "pair_off", "pair_ray", "pair_vox",
                             "pair_t", "feat_grid", "vox_feat", "vox_center"])
    for t, n in ((ray_dir, "ray_dir"), (pair_t, "pair_t"), (feat_grid, "feat_grid"),
                 (vox_feat, "vox_feat"), (vox_center, "vox_center")):
        if t is not None:
            _f32(t, n)
    for t, n in ((pair_off, "pair_off"), (pair_ray, "pair_ray"), (pair_vox, "pair_vox")):
        _i32(t, n)
    if feat_grid.dim() != 4 or feat_grid.shape[1] != 32:
        raise RuntimeError("feat_grid must be [B,32,h,w] (rgb_out=32)")
    if vox_feat.dim() != 2 or vox_feat.shape[1] != 128:
        raise RuntimeError("vox_feat must be [V,128] (pnet_out=128)")
    R, P = ray_dir.shape[0], pair_ray.shape[0]
    if pair_off.shape[0] != R + 1 or ray_pix.shape != (R, 2) or ray_bid.shape != (R,):
        raise RuntimeError("pair_off / ray_pix / ray_bid must be [R+1] / [R,2] / [R]")
    dev = ray_dir.device
    two = (factorised and P > 0 and os.environ.get("LIDF_TRAIN_STREAMS", "2") != "1"
           and not torch.cuda.is_current_stream_capturing())
    if factorised:
        # the pairs' position-embedding rows (one sweep over P x 102 floats) beside the per-ray RoIAlign features
        # (three light launches over R rays): independent, so the rows go to the training side stream
        pe = torch.empty((P, 2 * (3 + 6 * multires)), dtype=torch.float32, device=dev)
        side = _train_side_stream(dev) if two else None
        if two:
            side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.device(dev), (torch.cuda.stream(side) if two else contextlib.nullcontext()):
            _lib.check(_lib.lib().lidf_pe_rows_f32(
                _lib.ptr(pair_ray), _lib.ptr(pair_vox), _lib.ptr(pair_t), _lib.ptr(ray_dir),
                _lib.ptr(vox_center) if vox_center is not None else None, 1 if pos_rel else 0,
                multires, P, _lib.ptr(pe), _lib.current_stream(dev)))
    rayfeat = _RayFeaturesFn.apply(feat_grid, ray_dir, ray_pix, ray_bid, roi_inp_bbox, multires_views)
    if factorised:
        if two:
            torch.cuda.current_stream(dev).wait_stream(side)
        from . import decoders as _dec
        _dec._check_supported(prob_dec), _dec._check_supported(offset_dec)
        pp = [_dec._get(prob_dec, k) for k in _dec._PARAM_ORDER if _dec._has(prob_dec, k)]
        po = [_dec._get(offset_dec, k) for k in _dec._PARAM_ORDER if _dec._has(offset_dec, k)]
        pred_prob, pred_offset, pair_pred_pos, sm, mid, pred_pos = _QueryTrainFn.apply(
            prob_dec, offset_dec, vox_feat, rayfeat, pe, pair_off, pair_ray, pair_vox, pair_t, ray_dir, vox_center,
            pos_rel, multires, multires_views, float(offset_range[0]), float(offset_range[1]), float(part_size),
            max_pair_id, offsets == "selected", len(pp), *pp, *po)
        return {"pred_offset": pred_offset, "pred_prob_end": pred_prob, "pair_pred_pos": pair_pred_pos,
                "pred_prob_end_softmax": sm, "max_pair_id": mid, "pred_pos": pred_pos}
    else:
        rows = _BuildRowsFn.apply(vox_feat, rayfeat, pair_off, pair_ray, pair_vox, pair_t, ray_dir,
                                  vox_center, pos_rel, multires, multires_views)
        pred_prob = prob_dec(rows)
        pred_offset = offset_dec(rows)
    pair_pred_pos, sm, mid, pred_pos = _QueryTailFn.apply(
        pred_offset, pred_prob, pair_off, pair_ray, pair_t, ray_dir, float(offset_range[0]),
        float(offset_range[1]), float(part_size), max_pair_id)
    return {"pred_offset": pred_offset, "pred_prob_end": pred_prob, "pair_pred_pos": pair_pred_pos,
            "pred_prob_end_softmax": sm, "max_pair_id": mid, "pred_pos": pred_pos}



# ------------------------------------------------------------------------------------------------
# Training path of stage 2 (train_refine.py:393-399: loss.backward() through RefineNet.forward)
# ------------------------------------------------------------------------------------------------
def refine_perturb_noise(perturb_prob=0.8):
    """The train-only perturbation of RefineNet.get_pred_refine (models/pipeline.py:925-937), drawn
    with the reference's own np.random.random() calls in the reference's order (so a run seeded like
    the reference perturbs identically): returns the scalar `noise` (pred_pos += noise * ray_dir in
    the first iteration) or None when the draw says 'no perturbation'."""
    import numpy as np
    if not (np.random.random() < perturb_prob):
        return None
    prob = np.random.random()
    if prob < 0.5:
        return np.random.random() * (0 + 0.05) - 0.05
    if prob < 0.8:
        return np.random.random() * (0.05 - 0)
    if prob < 0.9:
        return np.random.random() * (-0.05 + 0.1) - 0.1
    return np.random.random() * (0.1 - 0.05) + 0.05


def _refine_train_args(cfg, x, rf, pn, do, out, end_voxel, ws, cells):
    """LidfRefineArgs of one stage-2 training call from plain tensors (forward: out / end_voxel are the call's
    outputs; backward: None — lidf_refine_train_backward_f32 writes neither)."""
    t = cfg["tensors"]
    B, _, h, w = t["rgb_img"].shape
    q = _lib.LidfRefineArgs()
    q.n_rays, q.ray_dir, q.ray_bid, q.ray_flat = x.shape[0], t["ray_dir"].data_ptr(), t["ray_bid"].data_ptr(), \
        t["ray_flat"].data_ptr()
    q.pred_pos, q.max_pair_id = x.data_ptr(), t["max_pair_id"].data_ptr()
    q.pair_vox, q.n_pairs = t["pair_vox"].data_ptr(), t["pair_vox"].shape[0]
    q.n_vox, q.voxel_bound, q.voxel_bid = t["voxel_bound"].shape[0], t["voxel_bound"].data_ptr(), \
        t["voxel_bid"].data_ptr()
    q.rgb_img, q.batch, q.height, q.width = t["rgb_img"].data_ptr(), B, h, w
    q.rayfeat = rf.data_ptr()
    q.n_valid, q.valid_inp, q.valid_vox = t["valid_inp"].shape[0], t["valid_inp"].data_ptr(), t["valid_vox"].data_ptr()
    q.pnet, q.off = C.pointer(pn), C.pointer(do)
    q.multires, q.multires_views = cfg["multires"], cfg["multires_views"]
    q.pos_rel, q.pnet_pos_rel = int(bool(cfg["pos_rel"])), int(bool(cfg["pnet_pos_rel"]))
    q.offset_range0, q.offset_range1 = float(cfg["offset_range"][0]), float(cfg["offset_range"][1])
    q.pred_pos_out = out.data_ptr() if out is not None else None
    q.end_voxel_id = end_voxel.data_ptr() if end_voxel is not None else None
    q.workspace, q.workspace_bytes = ws.data_ptr(), ws.numel()
    q.precision = PRECISIONS["f32"]
    if cells is not None:
        q.voxel_coord, q.cell_table = cells["coord"].data_ptr(), cells["table"].data_ptr()
        q.grid_res, q.grid_xmin, q.grid_part = cells["res"], cells["xmin"], cells["part"]
    return q


class _RefineTrainFn(torch.autograd.Function):
    """RefineNet.forward 'train' as ONE autograd node: lidf_refine_train_forward_f32 keeps what
    lidf_refine_train_backward_f32 needs in one `act` buffer (weight streams of the step, the per-ray layer-1
    table, per iteration the end voxels, PointNet rows / activations / arg rows, embed(pos) rows and the
    decoder's activations); the backward returns the gradient of every PointNet2Stage / decoder parameter
    (summed over the iterations inside the call), of pred_pos and of the per-ray features."""

    @staticmethod
    def forward(ctx, pred_pos, rayfeat, cfg, *params):
        from .decoders import _decoder_struct
        from .pointnet import _pn_struct_from
        t = cfg["tensors"]
        dev = pred_pos.device
        R, P, V, Nv = pred_pos.shape[0], t["pair_vox"].shape[0], t["voxel_bound"].shape[0], t["valid_inp"].shape[0]
        B, _, h, w = t["rgb_img"].shape
        T, L_, Lv = cfg["forward_times"], cfg["multires"], cfg["multires_views"]
        dec = cfg["offset_dec"]
        npn = 12
        pn_params, dec_params = params[:npn], params[npn:]
        keep = []
        pn = _pn_struct_from(pn_params, keep)
        do = _decoder_struct(dec, keep, tensors=dict(zip(cfg["dec_names"], dec_params)))
        L = _lib.lib()
        npass = dec.n_iter if do.is_ief else 1
        act = torch.empty((max(L.lidf_refine_train_act_bytes(R, Nv, V, L_, Lv, npass, T), 1),), dtype=torch.uint8,
                          device=dev)
        ws = torch.empty((max(L.lidf_refine_train_workspace_bytes(R, Nv, V, L_), 1),), dtype=torch.uint8, device=dev)
        out = torch.empty((R, 3), dtype=torch.float32, device=dev)
        end_voxel = torch.empty((R,), dtype=torch.int32, device=dev)
        x = pred_pos.detach().contiguous()
        rf = rayfeat.detach().contiguous()
        cells = _cell_lookup(cfg["grid"], V, B, dev, t["voxel_bound"])
        q = _refine_train_args(cfg, x, rf, pn, do, out, end_voxel, ws, cells)
        with torch.cuda.device(dev):
            _lib.check(L.lidf_refine_train_forward_f32(C.byref(q), T, _lib.ptr(act), act.numel(),
                                                       _lib.current_stream(dev)))
        # (nothing that references `out` is kept on ctx: a ctx-held closure over the output would close the cycle
        # out -> grad_fn -> ctx -> closure -> out and keep ~1 GB of workspace alive until Python's cycle collector
        # runs; the backward rebuilds the argument block from the saved tensors and allocates its own scratch)
        ctx.cfg, ctx.cells = cfg, cells
        ctx.mark_non_differentiable(end_voxel)
        ctx.save_for_backward(x, rf, act, *params)   # (saved: an in-place update before backward raises)
        return out, end_voxel

    @staticmethod
    def backward(ctx, g_pos, _g_end):
        from .decoders import _decoder_struct
        from .pointnet import _PN_FIELDS, _pn_struct_from
        cfg = ctx.cfg
        x, rf, act = ctx.saved_tensors[:3]
        params = ctx.saved_tensors[3:]
        npn = 12
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        grads = [torch.empty(p.shape, **f32) for p in params]
        gp = _lib.LidfPointNetGrads()
        for i, f in enumerate(_PN_FIELDS):
            setattr(gp, "w_" + f, grads[2 * i].data_ptr())
            setattr(gp, "b_" + f, grads[2 * i + 1].data_ptr())
        gd = _lib.LidfDecoderGrads()
        _map = {"linear_1.weight": "w1", "linear_1.bias": "b1", "linear_2.weight": "w2", "linear_2.bias": "b2",
                "linear_3.weight": "w3", "linear_3.bias": "b3", "linear_4.weight": "w4", "linear_4.bias": "b4",
                "offset_enc.weight": "wenc", "offset_enc.bias": "benc"}
        for name, gr in zip(cfg["dec_names"], grads[npn:]):
            setattr(gd, _map[name], gr.data_ptr())
        d_pos = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        d_rf = torch.empty_like(rf) if ctx.needs_input_grad[1] else None
        g = g_pos.detach().contiguous().float()
        # (the argument block is rebuilt from the SAVED tensors: same buffers as the forward's)
        keep = []
        pn = _pn_struct_from(params[:npn], keep)
        do = _decoder_struct(cfg["offset_dec"], keep, tensors=dict(zip(cfg["dec_names"], params[npn:])))
        R, Nv, V = x.shape[0], cfg["tensors"]["valid_inp"].shape[0], cfg["tensors"]["voxel_bound"].shape[0]
        ws = torch.empty((max(_lib.lib().lidf_refine_train_workspace_bytes(R, Nv, V, cfg["multires"]), 1),),
                         dtype=torch.uint8, device=dev)
        q = _refine_train_args(cfg, x, rf, pn, do, None, None, ws, ctx.cells)
        q.cell_table_ready = 1 if ctx.cells is not None else 0
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().lidf_refine_train_backward_f32(
                C.byref(q), cfg["forward_times"], _lib.ptr(act), act.numel(), _lib.ptr(g), _lib.ptr(d_pos),
                _lib.ptr(d_rf), C.byref(gp), C.byref(gd), _lib.current_stream(dev)))
        return (d_pos, d_rf, None) + tuple(gr if ctx.needs_input_grad[3 + i] else None for i, gr in enumerate(grads))


def lidf_refine_train(ray_dir, ray_pix, ray_bid, ray_flat, pred_pos, max_pair_id, pair_vox, voxel_bound,
                      voxel_bid, rgb_img, feat_grid, valid_inp, valid_vox, pnet_model, offset_dec,
                      forward_times=2, multires=8, multires_views=4, roi_inp_bbox=8,
                      offset_range=(-0.2, 0.2), pos_rel=False, pnet_pos_rel=True, perturb_noise=None, grid=None):
    """Differentiable RefineNet.forward (models/pipeline.py:922-1041, exp_type 'train'): the same
    arguments as lidf_refine; gradients reach every parameter of pnet_model (PointNet2Stage) and
    offset_dec (IEF / IMNet) — through the second and later iterations also via the refined position
    (PointNet input, embed(pos) and the additive term, as in the reference's autograd graph) — and
    feat_grid when it requires grad (RoIAlign backward). Stage 1 is frozen in train_refine.py:73, so
    pred_pos / max_pair_id arrive detached (a pred_pos that requires grad receives its gradient all the same).

    The whole step is two library calls (lidf_refine_train_forward_f32 / _backward_f32, one autograd node):
    one multi-pack of every weight stream per step, the decoder in its factorised form (per-voxel / per-ray /
    embed(pos) parts of layer 1; no [R, 334] row or input gradient is formed), parameter gradients summed
    over the iterations inside the call, no torch op between the launches. Widths other than the shipped
    ones (and decoders that are neither IEF nor IMNet of gf_dim 64) take the composed path
    (_lidf_refine_train_composed: the modules' own autograd functions joined by torch ops).
    perturb_noise: the scalar of refine_perturb_noise() — or None — applied in the first iteration
    (pipeline.py:937). grid: as lidf_refine (the end voxels through the cell table).
    Returns pred_pos_refine [R,3] (with grad) and the last end_voxel_id [R] i32."""
    from .decoders import is_shipped
    from .pointnet import _PN_ORDER, is_shipped as pnet_shipped
    E, Ed = 3 + 6 * multires, 3 + 6 * multires_views
    if not (is_shipped(offset_dec) and pnet_shipped(pnet_model) and feat_grid.shape[1] == 32
            and multires > 0 and offset_dec.inp_dim == 256 + E + Ed):
        return _lidf_refine_train_composed(ray_dir, ray_pix, ray_bid, ray_flat, pred_pos, max_pair_id, pair_vox,
                                           voxel_bound, voxel_bid, rgb_img, feat_grid, valid_inp, valid_vox,
                                           pnet_model, offset_dec, forward_times, multires, multires_views,
                                           roi_inp_bbox, offset_range, pos_rel, pnet_pos_rel, perturb_noise)
    ray_pix, ray_bid = _as_i32(ray_pix, "ray_pix"), _as_i32(ray_bid, "ray_bid")
    ray_flat, voxel_bid = _as_i32(ray_flat, "ray_flat"), _as_i32(voxel_bid, "voxel_bid")
    pair_vox, valid_vox = _as_i32(pair_vox, "pair_vox"), _as_i32(valid_vox, "valid_vox")
    ts = dict(ray_dir=ray_dir, ray_bid=ray_bid, ray_flat=ray_flat, max_pair_id=max_pair_id, pair_vox=pair_vox,
              voxel_bound=voxel_bound, voxel_bid=voxel_bid, rgb_img=rgb_img, valid_inp=valid_inp, valid_vox=valid_vox)
    _lib.require_cuda(pred_pos, feat_grid, *ts.values(), names=["pred_pos", "feat_grid"] + list(ts))
    for n in ("ray_dir", "voxel_bound", "rgb_img", "valid_inp"):
        _f32(ts[n], n)
    _f32(pred_pos, "pred_pos")
    if max_pair_id.dtype != torch.int64:
        raise RuntimeError("max_pair_id must be int64")
    R, V, Nv = ray_dir.shape[0], voxel_bound.shape[0], valid_inp.shape[0]
    if (tuple(ray_dir.shape) != (R, 3) or tuple(pred_pos.shape) != (R, 3) or tuple(max_pair_id.shape) != (R,)
            or tuple(ray_bid.shape) != (R,) or tuple(ray_flat.shape) != (R,)):
        raise RuntimeError("ray_dir / pred_pos must be [R,3]; ray_bid / ray_flat / max_pair_id [R]")
    if tuple(voxel_bound.shape) != (V, 6) or tuple(voxel_bid.shape) != (V,):
        raise RuntimeError("voxel_bound / voxel_bid must be [V,6] / [V]")
    if tuple(valid_inp.shape) != (Nv, 6) or tuple(valid_vox.shape) != (Nv,):
        raise RuntimeError("valid_inp / valid_vox must be [Nv,6] / [Nv]")
    if rgb_img.dim() != 4 or rgb_img.shape[1] != 3:
        raise RuntimeError("rgb_img must be [B,3,h,w]")
    if valid_inp.requires_grad:
        raise RuntimeError("lidf_refine_train: valid_inp carries no gradient (the valid points are inputs of the "
                           "frozen stage 1, trainers/train_refine.py:73)")
    # per-ray [ROI | embed(dir)] rows: constants of the iterations (RoIAlign backward when feat_grid trains)
    if torch.is_grad_enabled() and feat_grid.requires_grad:
        rayfeat = _RayFeaturesFn.apply(feat_grid, ray_dir, ray_pix, ray_bid, roi_inp_bbox, multires_views)
    else:
        rayfeat = ray_features(feat_grid.detach(), ray_dir, ray_pix, ray_bid, roi_inp_bbox, multires_views)
    cur = pred_pos
    if perturb_noise is not None:
        cur = cur + float(perturb_noise) * ray_dir
    dec_names = [n for n, _ in offset_dec.named_parameters()]
    params = [t for name in _PN_ORDER for t in (getattr(pnet_model, name).weight, getattr(pnet_model, name).bias)]
    params += [p for _, p in offset_dec.named_parameters()]
    cfg = dict(tensors={k: v.detach() for k, v in ts.items()}, forward_times=int(forward_times), multires=multires,
               multires_views=multires_views, pos_rel=pos_rel, pnet_pos_rel=pnet_pos_rel, offset_range=offset_range,
               offset_dec=offset_dec, dec_names=dec_names, grid=grid)
    pos, end_voxel = _RefineTrainFn.apply(cur, rayfeat, cfg, *params)
    return pos, end_voxel


def _lidf_refine_train_composed(ray_dir, ray_pix, ray_bid, ray_flat, pred_pos, max_pair_id, pair_vox, voxel_bound,
                      voxel_bid, rgb_img, feat_grid, valid_inp, valid_vox, pnet_model, offset_dec,
                      forward_times=2, multires=8, multires_views=4, roi_inp_bbox=8,
                      offset_range=(-0.2, 0.2), pos_rel=False, pnet_pos_rel=True, perturb_noise=None):
    """lidf_refine_train composed from the modules' own autograd functions (rounds 3-4; the path for widths
    other than the shipped ones, and the definition tests/test_train_gpu.py compares the fused step with).
    Differentiable RefineNet.forward (models/pipeline.py:922-1041, exp_type 'train'): the same
    arguments as lidf_refine; gradients reach every parameter of pnet_model (PointNet2Stage) and
    offset_dec (IEF / IMNet) — through the second and later iterations also via the refined position
    (PointNet input, embed(pos) and the additive term, as in the reference's autograd graph) — and
    feat_grid when it requires grad (RoIAlign backward). Stage 1 is frozen in train_refine.py:73, so
    pred_pos / max_pair_id arrive detached.

    Every differentiable stage runs liblidf_hip's training kernels (PointNet2Stage forward/backward
    with arg-routed max-pool, the embedder's backward, the decoder's forward that keeps activations +
    its backward, RoIAlign backward); the end-voxel lookup is the inference kernel (no gradient flows
    through an index). torch joins them (cat / row gather and their adjoints).
    perturb_noise: the scalar of refine_perturb_noise() — or None — applied in the first iteration
    (pipeline.py:937).  Returns pred_pos_refine [R,3] (with grad) and the last end_voxel_id [R] i32."""
    from .decoders import get_embedder
    from .extensions import pcl_aabb
    ray_pix, ray_bid = _as_i32(ray_pix, "ray_pix"), _as_i32(ray_bid, "ray_bid")
    ray_flat, voxel_bid = _as_i32(ray_flat, "ray_flat"), _as_i32(voxel_bid, "voxel_bid")
    pair_vox, valid_vox = _as_i32(pair_vox, "pair_vox"), _as_i32(valid_vox, "valid_vox")
    _lib.require_cuda(ray_dir, pred_pos, max_pair_id, voxel_bound, rgb_img, feat_grid, valid_inp,
                      names=["ray_dir", "pred_pos", "max_pair_id", "voxel_bound", "rgb_img", "feat_grid",
                             "valid_inp"])
    R, P, V = ray_dir.shape[0], pair_vox.shape[0], voxel_bound.shape[0]
    B, _, h, w = rgb_img.shape
    E, Ed = 3 + 6 * multires, 3 + 6 * multires_views
    if offset_dec.inp_dim != 256 + E + Ed:
        raise RuntimeError("refine offset_dec inp_dim must be %d" % (256 + E + Ed))
    embed_pos, _ = get_embedder(multires) if multires > 0 else (torch.nn.Identity(), 3)
    # per-ray [ROI | embed(dir)] rows: constants of the iterations (RoIAlign backward when feat_grid trains)
    if torch.is_grad_enabled() and feat_grid.requires_grad:
        rayfeat = _RayFeaturesFn.apply(feat_grid, ray_dir, ray_pix, ray_bid, roi_inp_bbox, multires_views)
    else:
        rayfeat = ray_features(feat_grid.detach(), ray_dir, ray_pix, ray_bid, roi_inp_bbox, multires_views)
    roi, dir_embed = rayfeat[:, :128], rayfeat[:, 128:]
    miss_rgb = rgb_img.permute(0, 2, 3, 1).reshape(-1, 3).index_select(
        0, ray_bid.long() * (h * w) + ray_flat.long())
    # voxel of the arg-max pair; the dummy row (a ray without pairs) selects voxel 0 (pipeline.py:941-943)
    pv = torch.cat((pair_vox.long(), torch.zeros(1, dtype=torch.long, device=pair_vox.device)))
    arg_vox = pv[max_pair_id.clamp(max=P)]
    r0, r1 = float(offset_range[0]), float(offset_range[1])
    cur = pred_pos
    if perturb_noise is not None:
        cur = cur + float(perturb_noise) * ray_dir
    end_voxel = None
    for _ in range(forward_times):
        with torch.no_grad():   # index selection: pcl_aabb + scatter max (pipeline.py:939-944)
            last = pcl_aabb.last_voxel(cur.detach().contiguous(), voxel_bound, ray_bid, voxel_bid).long()
            end_voxel = torch.maximum(arg_vox, last)
            vb = voxel_bound.index_select(0, end_voxel)
            center = (vb[:, :3] + vb[:, 3:]) / 2.0
        pred_inp = torch.cat(((cur - center) if pnet_pos_rel else cur, miss_rgb), 1)
        pn_inp = torch.cat((valid_inp, pred_inp), 0)
        pn_vox = torch.cat((valid_vox, end_voxel.to(torch.int32)), 0)
        occ_voxel_feat = pnet_model(pn_inp, pn_vox, n_vox=V)
        enter = (cur - center) if pos_rel else cur
        inp_embed = torch.cat((occ_voxel_feat.index_select(0, end_voxel), roi, embed_pos(enter), dir_embed), -1)
        off = offset_dec(inp_embed)
        cur = cur + (off * (r1 - r0) + r0) * ray_dir
    return cur, end_voxel.to(torch.int32)
