"""pipeline.py — the whole evaluation path of the reference's LIDF / RefineNet, chained on the device.

Reference counterparts (paths relative to /root/reference/src, eval flavour `exp_type='test'`):
  prepare_data        <- LIDF.prepare_data          models/pipeline.py:91-133
  get_valid_points    <- LIDF.get_valid_points      models/pipeline.py:135-160
  lidf_forward        <- LIDF.forward               models/pipeline.py:652-717
                         (get_occ_vox_bound :162-201, get_miss_ray :203-269, compute_ray_aabb
                          :271-296, get_embedding :338-425, get_pred :427-466, depth map :593-596)
  refine_forward      <- RefineNet.forward          models/pipeline.py:1032-1041
  eval_metrics        <- the bs == 1 statistics of LIDF.compute_loss   models/pipeline.py:577-627

Every compute step is a call into liblidf_hip.so through implicit_depth_amd.query / .pointnet;
what is written here in torch is plumbing only (views, index_select of inputs, dict handling).
The ResNet that produces `full_rgb_feat` is upstream of the path (out of scope): the
caller passes its output. Keys of the returned data_dict are the reference's.
"""
import torch

from . import _lib
from . import query as Q


class LidfOptions:
    """The options of the shipped configs that the path reads (experiments/implicit_depth/
    default_config.yaml + test_lidf.yaml / test_refine.yaml)."""

    def __init__(self, **kw):
        self.mask_type = "all"                 # 'all' | 'pred'
        self.multires, self.multires_views = 8, 4
        self.roi_inp_bbox = 8
        self.roi_out_bbox = 2                  # model.roi_out_bbox (2 in every shipped config)
        self.intersect_pos_type = "abs"
        self.offset_range = (0.0, 1.0)
        self.grid_res = 8
        self.xmin, self.xmax = (-1.0, -1.0, 0.0), (1.0, 1.0, 2.0)   # utils/constants.py:15-16
        self.valid_stride = None               # None = every valid point (valid_sample_num == -1)
        # stage 2 (test_refine.yaml)
        self.refine_forward_times = 2
        self.refine_offset_range = (-0.2, 0.2)
        self.refine_use_all_pix = True
        self.refine_pnet_pos_type = "rel"
        self.refine_intersect_pos_type = "abs"
        for k, v in kw.items():
            if not hasattr(self, k):
                raise TypeError("unknown option %s" % k)
            setattr(self, k, v)


def prepare_data(batch, opt, pred_mask=None):
    """LIDF.prepare_data, exp_type != 'train' (models/pipeline.py:91-133)."""
    rgb_img = batch["rgb"]
    bs, _, h, w = rgb_img.shape
    corrupt_mask = batch["corrupt_mask"].squeeze(1)
    valid_mask = batch["valid_mask"].squeeze(1) if "valid_mask" in batch else 1 - corrupt_mask
    dd = {
        "bs": bs, "h": h, "w": w, "rgb_img": rgb_img, "corrupt_mask": corrupt_mask,
        "valid_mask": valid_mask,
        "xyz_flat": batch["xyz"].permute(0, 2, 3, 1).contiguous().reshape(bs, -1, 3),
        "xyz_corrupt_flat": batch["xyz_corrupt"].permute(0, 2, 3, 1).contiguous().reshape(bs, -1, 3),
        "fx": batch["fx"].float(), "fy": batch["fy"].float(),
        "cx": batch["cx"].float(), "cy": batch["cy"].float(),
        "item_path": batch.get("item_path"),
    }
    if opt.mask_type == "pred":
        dd["pred_mask"] = pred_mask
        dd["valid_mask"] = 1 - pred_mask
    elif opt.mask_type == "all":
        dd["pred_mask"] = torch.ones_like(corrupt_mask)
        dd["valid_mask"] = 1 - (batch["depth_corrupt"] == 0).squeeze(1).float()
    else:
        raise NotImplementedError("mask_type %s" % opt.mask_type)
    return dd


def get_valid_points(dd, opt, valid_idx=None):
    """LIDF.get_valid_points (models/pipeline.py:135-160). valid_idx [Nv,2] (image, flat pixel)
    may be supplied (the reference's random block sampler, utils/point_utils.py:79-120, is host
    code upstream of the path); otherwise every valid pixel, optionally every opt.valid_stride-th."""
    bs = dd["bs"]
    if valid_idx is None:
        nz = Q.nonzero_pixels(dd["valid_mask"])
        valid_bid, valid_flat = nz["bid"], nz["flat"]
        if opt.valid_stride and opt.valid_stride > 1:
            valid_bid, valid_flat = valid_bid[::opt.valid_stride], valid_flat[::opt.valid_stride]
    else:
        valid_bid, valid_flat = valid_idx[:, 0].long(), valid_idx[:, 1].long()
    lin = valid_bid * (dd["h"] * dd["w"]) + valid_flat
    rgb_flat = dd["rgb_img"].permute(0, 2, 3, 1).contiguous().reshape(-1, 3)
    dd.update({
        "valid_bid": valid_bid, "valid_flat_img_id": valid_flat,
        "valid_xyz": dd["xyz_corrupt_flat"].reshape(-1, 3).index_select(0, lin),
        "valid_rgb": rgb_flat.index_select(0, lin),
    })
    return dd


def _mark(marks, name):
    """Benchmarks only: record a CUDA event named `name` on the current stream."""
    if marks is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        marks.append((name, ev))


def lidf_forward(batch, full_rgb_feat, pnet_model, prob_dec, offset_dec, opt=None, pred_mask=None,
                 valid_idx=None, precision="f32", workspace=None, marks=None, offsets="all"):
    """LIDF.forward for evaluation (models/pipeline.py:652-717): returns (success, data_dict).
    success False = one of the reference's early exits (no occupied voxel / no miss ray / no
    intersecting pair); data_dict then holds what was computed up to that point."""
    opt = opt or LidfOptions()
    Q._refuse_autograd("pipeline.lidf_forward", "query.lidf_query_train (stage-1 training step)",
                       (("full_rgb_feat", full_rgb_feat),),
                       (("pnet_model", pnet_model), ("prob_dec", prob_dec), ("offset_dec", offset_dec)))
    _mark(marks, "start")
    dd = prepare_data(batch, opt, pred_mask)
    get_valid_points(dd, opt, valid_idx)
    _mark(marks, "valid_points")
    bs, h, w = dd["bs"], dd["h"], dd["w"]
    # occupied voxels (get_occ_vox_bound)
    occ = Q.get_occ_vox_bound(dd["valid_xyz"].contiguous(), dd["valid_bid"].to(torch.int32).contiguous(),
                              bs, opt.xmin, opt.xmax, opt.grid_res)
    dd.update(occ)
    V = occ["voxel_bound"].shape[0]
    _mark(marks, "occupied_voxels")
    if V == 0:
        return False, dd
    # miss rays
    dd.update(Q.get_miss_ray(dd["pred_mask"], dd["fx"], dd["fy"], dd["cx"], dd["cy"]))
    _mark(marks, "miss_rays")
    if dd["total_miss_sample_num"] == 0:
        return False, dd
    # ray / voxel pairs (compact, ray-major)
    vox_bid = occ["occ_vox_bid"].to(torch.int32).contiguous()
    # densely occupied grids take the cell walk (bit-identical; measured on MI355X, 76,800 rays per
    # frame: 0.34 -> 0.22 ms at 729 voxels per frame, but 0.066 -> 0.082 ms at 74, where the
    # voxel-by-voxel loop is already shorter than the walk's two extra launches)
    grid = dict(voxel_coord=occ["voxel_coord"], grid_dims=occ["grid_dims"], batch=bs) if V > 256 * bs else {}
    pair_off, pair_ray, pair_vox, pair_t = Q.compute_ray_aabb(
        dd["miss_ray_dir"], occ["voxel_bound"], dd["ray_bid"], vox_bid, **grid)
    dd.update({"pair_off": pair_off, "pair_ray": pair_ray, "pair_vox": pair_vox, "pair_t": pair_t,
               "voxel_bid": vox_bid})
    _mark(marks, "ray_aabb")
    if pair_ray.shape[0] == 0:
        return False, dd
    # voxel embedding: PointNet over the valid points of every occupied voxel
    valid_v_rgb = dd["valid_rgb"].index_select(0, occ["valid_v_pid"])
    pnet_inp = torch.cat((occ["valid_v_rel_coord"], valid_v_rgb), -1)
    dd["pnet_inp"] = pnet_inp
    dd["occ_voxel_feat"] = pnet_model(pnet_inp, occ["revidx"], n_vox=V)
    dd["full_rgb_feat"] = full_rgb_feat
    _mark(marks, "pointnet")
    # get_embedding + get_pred + depth map (pred_xyz = xyz_corrupt with the rays' pixels replaced)
    depth = dd["xyz_corrupt_flat"][:, :, 2].reshape(bs, h, w).clone()
    vox_center = None
    if opt.intersect_pos_type == "rel":
        vb = occ["voxel_bound"]
        vox_center = ((vb[:, :3] + vb[:, 3:]) / 2.0).contiguous()
    out = Q.lidf_query(dd["miss_ray_dir"], dd["ray_pix"], dd["ray_bid"], pair_off, pair_ray, pair_vox,
                       pair_t, full_rgb_feat, dd["occ_voxel_feat"], prob_dec, offset_dec,
                       multires=opt.multires, multires_views=opt.multires_views,
                       roi_inp_bbox=opt.roi_inp_bbox, roi_out_bbox=opt.roi_out_bbox, offset_range=opt.offset_range,
                       part_size=occ["part_size"], vox_center=vox_center,
                       pos_rel=opt.intersect_pos_type == "rel", ray_flat=dd["ray_flat"], depth=depth,
                       want_rayfeat=True, precision=precision, workspace=workspace, offsets=offsets)
    dd.update(out)
    dd["pred_depth"] = depth
    _mark(marks, "query")
    return True, dd


def refine_forward(dd, pnet_model_refine, offset_dec_refine, opt=None, precision="f32", cell_lookup=True):
    """RefineNet.forward for evaluation (models/pipeline.py:1032-1041) on lidf_forward's data_dict:
    opt.refine_forward_times x get_pred_refine; adds pred_pos_refine and pred_depth_refine.
    cell_lookup: the end voxel of a ray through the cell table of get_occ_vox_bound's grid (the same ids as
    the reference's every-ray x every-voxel pcl_aabb, lidf_refine(grid=)); False tests every voxel."""
    opt = opt or LidfOptions()
    Q._refuse_autograd("pipeline.refine_forward", "the modules on their own (the fused stage-2 call has "
                       "no backward)", (("pred_pos", dd.get("pred_pos")),),
                       (("pnet_model_refine", pnet_model_refine), ("offset_dec_refine", offset_dec_refine)))
    bs, h, w = dd["bs"], dd["h"], dd["w"]
    occ_rev = dd["revidx"].to(torch.int32).contiguous()
    sel = None
    if opt.mask_type == "all" and not opt.refine_use_all_pix:
        inp_zero = (1 - dd["valid_mask"]).reshape(-1)
        sel = inp_zero.index_select(0, dd["miss_bid"] * (h * w) + dd["miss_flat_img_id"])
    valid_inp = dd["pnet_inp"]                     # cat(valid_v_rel_coord, valid_v_rgb), pipeline.py:1000
    if opt.refine_pnet_pos_type == "abs":          # :1001-1003
        valid_inp = torch.cat((dd["valid_xyz"].index_select(0, dd["valid_v_pid"]), dd["pnet_inp"][:, 3:]), -1)
    pos, end_voxel = Q.lidf_refine(
        dd["miss_ray_dir"], dd["ray_pix"], dd["ray_bid"], dd["ray_flat"], dd["pred_pos"],
        dd["max_pair_id"], dd["pair_vox"], dd["voxel_bound"], dd["voxel_bid"], dd["rgb_img"],
        dd["full_rgb_feat"], valid_inp.contiguous(), occ_rev, pnet_model_refine, offset_dec_refine,
        forward_times=opt.refine_forward_times, multires=opt.multires,
        multires_views=opt.multires_views, roi_inp_bbox=opt.roi_inp_bbox, roi_out_bbox=opt.roi_out_bbox,
        offset_range=opt.refine_offset_range, pos_rel=opt.refine_intersect_pos_type == "rel",
        pnet_pos_rel=opt.refine_pnet_pos_type == "rel", rayfeat=dd.get("rayfeat"),
        precision=precision, pnet_select=sel,
        grid=dd if (cell_lookup and all(k in dd for k in ("voxel_coord", "grid_dims", "xmin", "part_size"))) else None)
    dd["pred_pos_refine"], dd["end_voxel_id"] = pos, end_voxel
    depth = dd["xyz_corrupt_flat"][:, :, 2].reshape(-1).clone()
    depth[dd["miss_bid"] * (h * w) + dd["miss_flat_img_id"]] = pos[:, 2]
    dd["pred_depth_refine"] = depth.reshape(bs, h, w)
    return dd


def eval_metrics(dd, key="pred_depth", frame=0):
    """The nine ClearGrasp statistics of LIDF.compute_loss's bs == 1 branch
    (models/pipeline.py:577-627) for one frame, on the device."""
    h, w = dd["h"], dd["w"]
    gt = dd["xyz_flat"][frame, :, 2].reshape(h, w).contiguous()
    return Q.depth_metrics(dd[key][frame].contiguous(), gt, dd["corrupt_mask"][frame])


# ------------------------------------------------------------------------------------------------
# The same evaluation path as ONE library call per batch, without a host round trip
# ------------------------------------------------------------------------------------------------
_FC = {"R": 0, "P": 1, "V": 2, "NV": 3, "NV0": 4, "NVS": 5, "NPN": 6, "OVERFLOW": 7}


class FrameRunner:
    """LIDF.forward (+ RefineNet.forward) for evaluation through lidf_frame_f32: every compacted list
    (valid points, occupied voxels, rays, ray/voxel pairs) lives in a buffer sized for the worst case
    and its length stays on the device, so a frame is enqueued without a single .item() / nonzero()
    size read (the reference — and lidf_forward above — stall the stream four times per frame), and
    the launch sequence depends on (batch, h, w, max_pairs) only: `capture()` records it into a HIP
    graph that `run()` replays.

        runner = FrameRunner(bs, h, w, device, pnet, prob_dec, offset_dec, opt,
                             pnet_refine=..., offset_dec_refine=...)
        runner.run(batch, full_rgb_feat)      # enqueue only (optionally: runner.capture(...) once)
        dd = runner.result()                  # ONE sync: reads the counts, returns the data_dict

    result() gives the reference's data_dict keys as views of the capacity buffers (int32 index
    tensors; `reference_dtypes=True` adds the int64 forms). mask_type 'all' / 'pred', every valid
    pixel or every opt.valid_stride-th; intersect_pos_type / refine_intersect_pos_type / refine_pnet_pos_type 'abs' or
    'rel'; precision "f32"
    or "f16x3" (as lidf_query / lidf_refine).
    max_pairs bounds the pair list (default 32 per pixel: a ray crosses at most 25 cells of the 9^3
    grid); a frame with more pairs raises in result().
    offsets="selected" (opt-in; see query.lidf_query): the offset decoder runs on the arg-max pair of every ray
    only — pred_pos, the depth maps, stage 2 and the statistics are bit-identical, pred_offset / pair_pred_pos
    are meaningful at the selected pairs only (other rows: NaN, or a previous frame's selected values).

    Weight streams: the runner owns ONE blob with the packed streams of all its modules and the
    device-side fingerprints they were built from (nothing of it lives in the per-module caches, so a
    captured graph never references memory another call can free). Every frame re-validates them with one
    fingerprint launch over all parameter buffers (+ two early-exit pack launches, ~20 us) — in-place
    updates, `p.data` writes and load_state_dict are picked up by the next frame. guard_every = N > 1
    validates every N-th frame only and trusts the blob in between (evaluation loops whose parameters do
    not change: the reference's eval runs under torch.no_grad() with the modules in eval());
    `invalidate()` forces the check on the next frame; modules frozen with _lib.freeze_packed are
    validated once."""

    def __init__(self, bs, h, w, device, pnet_model, prob_dec, offset_dec, opt=None, pnet_model_refine=None,
                 offset_dec_refine=None, max_pairs=None, lds_voxels=None,
                 precision="f32", guard_every=1, offsets="all", side_stream=None):
        import ctypes as C
        import math
        from .decoders import _check_supported
        from .pointnet import check_pointnet
        self.C = C
        self.opt = opt = opt or LidfOptions()
        if opt.intersect_pos_type not in ("abs", "rel"):
            raise NotImplementedError("intersect_pos_type %s" % opt.intersect_pos_type)
        if opt.mask_type not in ("all", "pred"):
            raise NotImplementedError("mask_type %s" % opt.mask_type)
        if precision not in Q.PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(Q.PRECISIONS))
        self.precision = precision
        if offsets not in ("all", "selected") or (offsets == "selected" and precision != "f32"):
            raise ValueError("offsets must be 'all' or 'selected' (f32)")
        self.offsets = offsets      # 'selected': offset_dec on the arg-max pair of every ray only (lidf_query)
        self.bs, self.h, self.w, self.dev = bs, h, w, torch.device(device)
        self.mods = (pnet_model, prob_dec, offset_dec, pnet_model_refine, offset_dec_refine)
        what = "FrameRunner (use lidf_forward / refine_forward, which run other widths layer by layer)"
        _check_supported(prob_dec, what), _check_supported(offset_dec, what), check_pointnet(pnet_model, what)
        if opt.roi_out_bbox != 2:
            raise RuntimeError("lidf_hip: %s is built for roi_out_bbox = 2" % what)
        self.refine = pnet_model_refine is not None
        if self.refine:
            _check_supported(offset_dec_refine, what), check_pointnet(pnet_model_refine, what)
        # the widened grid of LIDF.get_occ_vox_bound (models/pipeline.py:167-173), in the reference's f32
        t32 = lambda v: torch.tensor(v, dtype=torch.float32)  # noqa: E731
        lo, hi = t32(opt.xmin), t32(opt.xmax)
        self.part_size = float(torch.min(hi - lo).item()) / opt.grid_res
        lo, hi = lo - 0.5 * self.part_size, hi + 0.5 * self.part_size
        self.xmin = [float(v) for v in lo.tolist()]
        self.res = [int(v) for v in torch.ceil((hi - lo) / self.part_size).tolist()]
        N, Cc = bs * h * w, bs * self.res[0] * self.res[1] * self.res[2]
        self.N, self.Cc = N, Cc
        self.max_pairs = int(max_pairs) if max_pairs else 32 * N
        # bound of the PointNet's LDS pooling tables. One frame: 288 (the largest table that fits; a frame
        # with more occupied voxels falls to per-point global atomics, +0.3 ms) at 1.5 % over the 128 that
        # lets two workgroups share a CU; batches: 128 (their larger tables take the voxel-sorted walk)
        self.lds_voxels = int(lds_voxels) if lds_voxels else (288 if bs == 1 else 128)
        Ed = 3 + 6 * opt.multires_views
        dev = self.dev
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        P = self.max_pairs
        self.buf = {
            "counts": torch.zeros((8,), **i32),
            "valid_bid": torch.empty((N,), **i32), "valid_flat": torch.empty((N,), **i32),
            "valid_xyz": torch.empty((N, 3), **f32), "valid_rgb": torch.empty((N, 3), **f32),
            "occ_bid_coord": torch.empty((Cc, 4), **i32), "voxel_bound": torch.empty((Cc, 6), **f32),
            "valid_v_pid": torch.empty((2 * N,), **i32), "revidx": torch.empty((2 * N,), **i32),
            "valid_v_rel_coord": torch.empty((N, 3), **f32), "pnet_inp": torch.empty((2 * N, 6), **f32),
            "occ_voxel_feat": torch.empty((Cc, 128), **f32),
            "ray_bid": torch.empty((N,), **i32), "ray_flat": torch.empty((N,), **i32),
            "ray_pix": torch.empty((N, 2), **i32), "ray_dir": torch.empty((N, 3), **f32),
            "pair_off": torch.zeros((N + 1,), **i32), "pair_ray": torch.empty((P,), **i32),
            "pair_vox": torch.empty((P,), **i32), "pair_t": torch.empty((P, 2), **f32),
            "pred_offset": torch.empty((P,), **f32), "pred_prob": torch.empty((P,), **f32),
            "pred_prob_softmax": torch.empty((P,), **f32), "pair_pred_pos": torch.empty((P, 3), **f32),
            "max_pair_id": torch.empty((N,), dtype=torch.int64, device=dev),
            "pred_pos": torch.empty((N, 3), **f32), "rayfeat": torch.empty((N, 128 + Ed), **f32),
            "pred_depth": torch.empty((bs, h, w), **f32),
        }
        if self.offsets == "selected":   # rows of the per-pair outputs that no frame writes must not look like results
            self.buf["pred_offset"].fill_(float("nan"))
            self.buf["pair_pred_pos"].fill_(float("nan"))
        if self.refine:
            self.buf.update({"pred_pos_refine": torch.empty((N, 3), **f32),
                             "end_voxel_id": torch.empty((N,), **i32),
                             "pred_depth_refine": torch.empty((bs, h, w), **f32)})
        L = _lib.lib()
        self.times = opt.refine_forward_times if self.refine else 0
        res = (C.c_int32 * 3)(*self.res)
        wsb = L.lidf_frame_workspace_bytes(bs, h, w, res, self.max_pairs, self.lds_voxels, self.times)
        self.ws = torch.empty((max(wsb, 1),), dtype=torch.uint8, device=dev)
        # static inputs (a captured graph replays fixed addresses): load() copies a batch into them
        self.inp = {
            "rgb": torch.empty((bs, 3, h, w), **f32), "xyz_corrupt": torch.empty((bs, 3, h, w), **f32),
            "valid_mask": torch.empty((bs, h, w), **f32), "intr": torch.empty((bs, 4), **f32),
            "feat_grid": torch.empty((bs, 32, h, w), **f32),
            "miss_mask": torch.empty((bs, h, w), **f32) if opt.mask_type == "pred" else None,
        }
        self.graph = None
        self.graph_trusted = None                  # the same launch sequence without the guard (guard_every > 1)
        self._keep = None
        self.guard_every = max(1, int(guard_every))
        self._frames_since_guard = None            # None: the next frame validates
        # side_stream (lidf_hip.h: LidfFrameArgs.aux_stream): the weight-stream guard, the box sums and the
        # per-ray features of a frame run on a second stream beside its head, voxel list, pairs and PointNet;
        # bit-identical results. None (default) = for eager calls only: a replayed graph pays more for the
        # cross-stream edges than they save, and frames pipelined over several streams fill each other's gaps
        # already (FramePipeline passes False). True / False force it.
        # In a process that also runs a process group (RCCL creates streams of its own) HIP's default of 4 hardware
        # queues can put this side stream and the caller's stream on ONE queue: the fork then runs in sequence and a
        # frame is 6 % slower than without the group (A/B in one session, profiles/r06_ab_pg.txt: 1.660 ms plain,
        # 1.766 under a 1-rank nccl group, 1.675 with GPU_MAX_HW_QUEUES=8, 1.700 without the side stream). Export
        # GPU_MAX_HW_QUEUES=8 before the first HIP call of such a process (bench.py --workload e2e --gpus N does).
        self.side_mode = side_stream
        self.side = None                           # (stream, ev_fork, ev_join): created by the first frame that forks
        self._fail_after = 0                       # LidfFrameArgs.fail_after (test hook: a mid-frame failure)
        self.profile_events = None                 # benchmarks: six hipEvent_t recorded around the matrix launches
        self.pack_blob = torch.empty((L.lidf_frame_pack_bytes(),), dtype=torch.uint8, device=dev)
        self.pack_guard = torch.zeros((L.lidf_frame_pack_guard_bytes(),), dtype=torch.uint8, device=dev)
        self.vidx, self.n_valid_idx = None, 0      # explicit valid points (load(valid_idx=))
        self.src = dict(self.inp)
        math.isfinite(self.part_size)

    # -- inputs -----------------------------------------------------------------------------------
    def load(self, batch, full_rgb_feat, pred_mask=None, copy=None, valid_idx=None):
        """Hand a batch (the reference's dataset item keys) to the runner. mask_type 'all': valid <=>
        the measured depth is non-zero (prepare_data, pipeline.py:119-121), so depth_corrupt itself is
        the valid mask; 'pred': valid_mask = 1 - pred_mask, rays where pred_mask is non-zero.
        valid_idx [M,2] (image, flat pixel; any integer dtype, M <= bs*h*w): the valid points as the
        code upstream sampled them (LIDF.get_valid_points with grid.valid_sample_num != -1 keeps the
        output of utils/point_utils.py sample_valid_points) instead of every opt.valid_stride-th valid
        pixel; copied into the runner's static index buffers without a sync. A captured graph replays
        the M it was captured with.
        copy=True (the default once a graph is captured: a graph replays fixed addresses) copies the
        tensors into the runner's static input buffers — device-to-device, no sync; copy=False passes
        the caller's own contiguous float32 tensors to the library as they lie (kept alive by the
        runner until the next load)."""
        copy = (self.graph is not None) if copy is None else copy
        bs, h, w = self.bs, self.h, self.w
        i = self.inp
        intr = [batch[k].reshape(bs) for k in ("fx", "fy", "cx", "cy")]
        if all(t.dtype == torch.float32 for t in intr):
            torch.stack(intr, 1, out=i["intr"])
        else:
            i["intr"].copy_(torch.stack(intr, 1), non_blocking=True)
        if self.opt.mask_type == "all":
            valid, miss = batch["depth_corrupt"].reshape(bs, h, w), None
        else:
            miss = pred_mask.reshape(bs, h, w)
            valid = None
        srcs = {"rgb": batch["rgb"], "xyz_corrupt": batch["xyz_corrupt"], "feat_grid": full_rgb_feat,
                "valid_mask": valid, "miss_mask": miss}
        direct = not copy and all(t is None or (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
                                                and t.device == self.dev) for t in srcs.values())
        if direct:
            self.src = dict(srcs, intr=i["intr"])
        else:
            if self.graph is None and not copy:
                copy = True   # (not plain contiguous float32 device tensors: go through the static buffers)
            for k in ("rgb", "xyz_corrupt", "feat_grid"):
                i[k].copy_(srcs[k], non_blocking=True)
            if valid is not None:
                i["valid_mask"].copy_(valid, non_blocking=True)
            if miss is not None:
                i["miss_mask"].copy_(miss, non_blocking=True)
            self.src = dict(i)
        if miss is not None:   # valid_mask = 1 - pred_mask (prepare_data, pipeline.py:116-117)
            torch.sub(1.0, self.src["miss_mask"], out=i["valid_mask"])
            self.src["valid_mask"] = i["valid_mask"]
        m = 0
        if valid_idx is not None:
            if valid_idx.dim() != 2 or valid_idx.shape[1] != 2:
                raise RuntimeError("valid_idx must be [M,2] (image, flat pixel)")
            m = int(valid_idx.shape[0])
            if m > self.N:
                raise RuntimeError("valid_idx holds %d points, the frame buffers %d" % (m, self.N))
            if self.vidx is None:
                self.vidx = torch.empty((2, self.N), dtype=torch.int32, device=self.dev)
            if m:
                self.vidx[:, :m].copy_(valid_idx.t(), non_blocking=True)
        if self.graph is not None and m != self.n_valid_idx:
            raise RuntimeError("the captured graph replays %d listed valid points, this batch has %d"
                               % (self.n_valid_idx, m))
        self.n_valid_idx = m

    # -- the launch sequence ------------------------------------------------------------------------
    def invalidate(self):
        """The next frame re-validates the packed weight streams (guard_every > 1, frozen modules)."""
        self._frames_since_guard = None

    def _guard_due(self):
        """Whether the coming frame fingerprints the parameters (and re-packs what changed)."""
        n = self._frames_since_guard
        if n is None:
            return True
        if all(m is None or m in _lib.FROZEN for m in self.mods):
            return False
        return n >= self.guard_every

    def enqueue(self, guard=True):
        """lidf_frame_f32 on the static inputs: one fingerprint launch over every module's parameters +
        the early-exit packs (guard=True) + the frame; nothing here reads a size or waits for the device."""
        from .decoders import _decoder_struct
        from .pointnet import pointnet_struct
        C = self.C
        pnet, prob, off, pnet_r, off_r = self.mods
        opt, b, i = self.opt, self.buf, self.src
        keep = []
        dp, do = _decoder_struct(prob, keep), _decoder_struct(off, keep)
        pn = pointnet_struct(pnet, keep)
        a = _lib.LidfFrameArgs()
        a.batch, a.height, a.width = self.bs, self.h, self.w
        for k in ("rgb", "xyz_corrupt", "valid_mask", "intr", "feat_grid"):
            setattr(a, k, i[k].data_ptr())
        a.miss_mask = i["miss_mask"].data_ptr() if i["miss_mask"] is not None else None
        a.xmin = (C.c_float * 3)(*self.xmin)
        a.res = (C.c_int32 * 3)(*self.res)
        a.part_size = self.part_size
        a.valid_stride = int(opt.valid_stride) if opt.valid_stride and opt.valid_stride > 1 else 1
        a.pnet, a.prob, a.off = C.pointer(pn), C.pointer(dp), C.pointer(do)
        a.packed_query = None
        a.multires, a.multires_views, a.roi_inp_bbox = opt.multires, opt.multires_views, opt.roi_inp_bbox
        a.pos_rel = int(opt.intersect_pos_type == "rel")
        a.offset_range0, a.offset_range1 = float(opt.offset_range[0]), float(opt.offset_range[1])
        a.refine_times = self.times
        a.precision = Q.PRECISIONS[self.precision]
        if self.refine:
            dr = _decoder_struct(off_r, keep)
            pr = pointnet_struct(pnet_r, keep)
            a.pnet_refine, a.off_refine, a.packed_refine = C.pointer(pr), C.pointer(dr), None
            a.refine_pos_rel = int(opt.refine_intersect_pos_type == "rel")
            a.refine_pnet_pos_rel = int(opt.refine_pnet_pos_type == "rel")
            a.refine_use_all_pix = int(bool(opt.refine_use_all_pix) or opt.mask_type != "all")
            a.refine_offset_range0 = float(opt.refine_offset_range[0])
            a.refine_offset_range1 = float(opt.refine_offset_range[1])
        a.max_pairs, a.lds_voxels = self.max_pairs, self.lds_voxels
        for k, t in b.items():
            setattr(a, k, t.data_ptr())
        a.workspace, a.workspace_bytes = self.ws.data_ptr(), self.ws.numel()
        if self.n_valid_idx > 0:
            a.valid_idx_bid, a.valid_idx_flat = self.vidx[0].data_ptr(), self.vidx[1].data_ptr()
            a.n_valid_idx = self.n_valid_idx
        # the runner's own packed streams + fingerprints (lidf_hip.h: LIDF_FRAME_PACK_GUARDED / _TRUSTED)
        a.pack_blob, a.pack_blob_bytes = self.pack_blob.data_ptr(), self.pack_blob.numel()
        a.pack_guard = self.pack_guard.data_ptr()
        a.pack_mode = 1 if guard else 2
        a.offsets_selected = int(self.offsets == "selected")
        if self.side_mode or (self.side_mode is None and not torch.cuda.is_current_stream_capturing()):
            if self.side is None:   # the side stream and its two events exist only for runners that fork
                with torch.cuda.device(self.dev):
                    self.side = (torch.cuda.Stream(self.dev), _lib.hip_event(), _lib.hip_event())
            a.aux_stream, a.ev_fork, a.ev_join = self.side[0].cuda_stream, self.side[1], self.side[2]
        a.fail_after = int(self._fail_after)
        if self.profile_events is not None:   # (benchmarks: six hipEvent_t, LidfFrameArgs.profile_events)
            self._pev = (C.c_void_p * 6)(*[e if isinstance(e, C.c_void_p) else C.c_void_p(e)
                                           for e in self.profile_events])
            a.profile_events = C.cast(self._pev, C.POINTER(C.c_void_p))
        try:
            with torch.cuda.device(self.dev):
                _lib.check(_lib.lib().lidf_frame_f32(C.byref(a), _lib.current_stream(self.dev)))
        except RuntimeError:
            self._frames_since_guard = None   # (the library reset the fingerprints; validate again)
            raise
        self._keep = keep

    def __del__(self):
        # the two raw events of the side stream are the runner's (created through the library)
        side, self.side = getattr(self, "side", None), None
        if side is not None:
            try:
                self.graph = self.graph_trusted = None   # (a captured graph may reference them)
                torch.cuda.synchronize(self.dev)
                _lib.hip_event_destroy(side[1]), _lib.hip_event_destroy(side[2])
            except Exception:   # interpreter shutdown: the process releases them
                pass

    def capture(self):
        """Record enqueue() into a HIP graph (torch.cuda.CUDAGraph) after one eager warm-up call; run()
        then replays it. The graph holds the parameters' CURRENT storage pointers: in-place updates are
        picked up by the fingerprint check inside the graph, a replaced `.data` needs a new capture.
        With guard_every > 1 (or frozen modules) a second graph without the check is recorded as well."""
        for k, t in self.src.items():        # a batch handed over in place moves into the static buffers:
            if t is not None and self.inp.get(k) is not None and t is not self.inp[k]:
                self.inp[k].copy_(t)         # the graph reads those
        self.src = dict(self.inp)
        # warm-up on the stream the capture will use: kernel attributes are set, the packed weight
        # streams exist (and are valid), nothing is allocated during the capture
        st = torch.cuda.Stream(self.dev)
        st.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(st):
            self.enqueue(True)
        torch.cuda.synchronize(self.dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            self.enqueue(True)
        self.graph = g
        self.graph_trusted = None
        if self.guard_every > 1 or all(m is None or m in _lib.FROZEN for m in self.mods):
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, stream=st):
                self.enqueue(False)
            self.graph_trusted = g2
        self._frames_since_guard = None
        return self

    def run(self, batch=None, full_rgb_feat=None, pred_mask=None, valid_idx=None):
        """load() (when a batch is given) + the frame (graph replay if captured). No sync."""
        Q._refuse_autograd("pipeline.FrameRunner.run", "query.lidf_query_train (stage-1 training step)",
                           (("full_rgb_feat", full_rgb_feat),),
                           tuple(zip(("pnet_model", "prob_dec", "offset_dec", "pnet_model_refine",
                                      "offset_dec_refine"), self.mods)))
        if batch is not None:
            self.load(batch, full_rgb_feat, pred_mask, valid_idx=valid_idx)
        guard = self._guard_due()
        if self.graph is not None:
            (self.graph if guard or self.graph_trusted is None else self.graph_trusted).replay()
        else:
            self.enqueue(guard)
        self._frames_since_guard = 1 if guard else self._frames_since_guard + 1
        return self

    # -- outputs ------------------------------------------------------------------------------------
    def counts(self):
        """The list lengths (ONE device -> host copy; the only sync of a frame)."""
        c = self.buf["counts"].tolist()
        return {k: c[i] for k, i in _FC.items()}

    def result(self, reference_dtypes=False):
        """(success, data_dict) as lidf_forward / refine_forward return them: views of the capacity
        buffers cut to the frame's counts. success False = one of the reference's early exits (no
        occupied voxel / no ray / no pair: the launches ran on empty lists)."""
        c = self.counts()
        if c["OVERFLOW"]:
            raise RuntimeError("FrameRunner: the frame has more than max_pairs = %d ray/voxel pairs; "
                               "construct the runner with a larger max_pairs" % self.max_pairs)
        b, R, P, V, NV, NVS = self.buf, c["R"], c["P"], c["V"], c["NV"], c["NVS"]
        dd = {
            "bs": self.bs, "h": self.h, "w": self.w, "part_size": self.part_size,
            "xmin": torch.tensor(self.xmin, device=self.dev), "grid_dims": tuple(self.res),
            "valid_bid": b["valid_bid"][:NVS], "valid_flat_img_id": b["valid_flat"][:NVS],
            "valid_xyz": b["valid_xyz"][:NVS], "valid_rgb": b["valid_rgb"][:NVS],
            "occ_vox_bid": b["occ_bid_coord"][:V, 0], "occ_vox_global_coord": b["occ_bid_coord"][:V, 1:],
            "voxel_bound": b["voxel_bound"][:V], "valid_v_pid": b["valid_v_pid"][:NV],
            "revidx": b["revidx"][:NV], "valid_v_rel_coord": b["valid_v_rel_coord"][:NV],
            "pnet_inp": b["pnet_inp"][:NV], "occ_voxel_feat": b["occ_voxel_feat"][:V],
            "ray_bid": b["ray_bid"][:R], "ray_flat": b["ray_flat"][:R], "ray_pix": b["ray_pix"][:R],
            "miss_ray_dir": b["ray_dir"][:R], "total_miss_sample_num": R,
            "pair_off": b["pair_off"][:R + 1], "pair_ray": b["pair_ray"][:P], "pair_vox": b["pair_vox"][:P],
            "pair_t": b["pair_t"][:P], "pred_offset": b["pred_offset"][:P].unsqueeze(1),
            "pred_prob_end": b["pred_prob"][:P].unsqueeze(1), "pair_pred_pos": b["pair_pred_pos"][:P],
            "pred_prob_end_softmax": b["pred_prob_softmax"][:P], "max_pair_id": b["max_pair_id"][:R],
            "pred_pos": b["pred_pos"][:R], "rayfeat": b["rayfeat"][:R], "pred_depth": b["pred_depth"],
            "full_rgb_feat": self.src["feat_grid"], "rgb_img": self.src["rgb"], "counts": c,
        }
        if reference_dtypes:   # the reference's int64 index tensors (torch.nonzero / torch.unique)
            dd["miss_bid"], dd["miss_flat_img_id"] = dd["ray_bid"].long(), dd["ray_flat"].long()
            dd["miss_img_ind"] = dd["ray_pix"].long()
            for k in ("valid_bid", "valid_flat_img_id", "occ_vox_bid", "occ_vox_global_coord", "valid_v_pid",
                      "revidx"):
                dd[k] = dd[k].long()
        ok = V > 0 and R > 0 and P > 0
        if self.refine and ok:
            dd["pred_pos_refine"] = b["pred_pos_refine"][:R]
            dd["end_voxel_id"] = b["end_voxel_id"][:R]
            dd["pred_depth_refine"] = b["pred_depth_refine"]
        return ok, dd

    def metrics(self, batch, key=None, frame=0):
        """The nine ClearGrasp statistics (models/pipeline.py:577-627) of frame `frame` on the device:
        xyz[frame, 2] of the NCHW batch is the ground-truth depth map as it lies (no copy)."""
        key = key or ("pred_depth_refine" if self.refine else "pred_depth")
        seg = batch["corrupt_mask"].reshape(self.bs, self.h, self.w)[frame]
        return Q.depth_metrics(self.buf[key][frame], batch["xyz"][frame, 2], seg)


class FramePipeline:
    """Independent frames pipelined over S streams (2.09 -> 1.61 ms per 240x320 frame at
    S = 3 on one MI355X): S FrameRunners, each with its own buffers, packed-weight entries and HIP
    stream, take the batches in turn, so that one frame's low-occupancy stretches (PointNet chains, the
    per-voxel layers, scans, the partial last round of the matrix kernels) are filled by its neighbours'
    kernels. Frames come back in submission order, each with its one size read.

        pipe = FramePipeline(3, bs, h, w, device, pnet, prob_dec, offset_dec, opt, pnet_refine, offset_refine)
        for batch in loader:
            if pipe.full:
                ok, data_dict, metrics = pipe.collect()            # the oldest frame; frees its runner
                ...                                                # use (or clone) data_dict here
            pipe.submit(batch, lidf.resnet_model(batch['rgb']))   # enqueue only
        while pipe.pending:
            ok, data_dict, metrics = pipe.collect()

    data_dict holds views of a runner's buffers, valid until the submit() that reuses the runner (the one
    after S - 1 further submits). A runner's stream waits for the caller's current stream before it starts
    a frame: inputs produced there and reads of the previous result enqueued there are ordered before it."""

    def __init__(self, streams, bs, h, w, device, *models, with_metrics=True, **kw):
        if streams < 1:
            raise ValueError("streams must be >= 1")
        self.dev = torch.device(device)
        if streams > 1:
            kw.setdefault("side_stream", False)   # (the lanes fill each other's gaps: see FrameRunner)
        self.runners = [FrameRunner(bs, h, w, device, *models, **kw) for _ in range(streams)]
        self.lanes = [torch.cuda.Stream(self.dev) for _ in range(streams)]
        self.with_metrics = with_metrics
        self.pending = []          # [(slot, metrics)] in submission order
        self.n = 0

    @property
    def full(self):
        return len(self.pending) == len(self.runners)

    def submit(self, batch, full_rgb_feat, pred_mask=None, valid_idx=None):
        if self.full:
            raise RuntimeError("FramePipeline: %d frames in flight — collect() the oldest first" % len(self.pending))
        slot = self.n % len(self.runners)
        self.n += 1
        lane = self.lanes[slot]
        lane.wait_stream(torch.cuda.current_stream(self.dev))
        # the lane reads the caller's tensors (copies into the runner's static buffers, the library call when
        # they are handed over in place, the metrics kernel): tell the caching allocator, or a tensor the
        # caller drops on its next loop iteration could be handed out again on the caller's stream while the
        # lane's reads are still queued
        for t in list(batch.values()) + [full_rgb_feat, pred_mask, valid_idx]:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(lane)
        with torch.cuda.stream(lane), torch.no_grad():
            self.runners[slot].run(batch, full_rgb_feat, pred_mask, valid_idx=valid_idx)
            m = self.runners[slot].metrics(batch) if self.with_metrics else None
        self.pending.append((slot, m))

    def collect(self):
        """(success, data_dict, metrics) of the oldest frame in flight (waits for that frame only)."""
        if not self.pending:
            raise RuntimeError("FramePipeline: nothing in flight")
        slot, m = self.pending.pop(0)
        with torch.cuda.stream(self.lanes[slot]):
            ok, dd = self.runners[slot].result()
        return ok, dd, m
