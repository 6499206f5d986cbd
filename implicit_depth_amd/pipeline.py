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
                 valid_idx=None, precision="f32", workspace=None, marks=None):
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
                       roi_inp_bbox=opt.roi_inp_bbox, offset_range=opt.offset_range,
                       part_size=occ["part_size"], vox_center=vox_center,
                       pos_rel=opt.intersect_pos_type == "rel", ray_flat=dd["ray_flat"], depth=depth,
                       want_rayfeat=True, precision=precision, workspace=workspace)
    dd.update(out)
    dd["pred_depth"] = depth
    _mark(marks, "query")
    return True, dd


def refine_forward(dd, pnet_model_refine, offset_dec_refine, opt=None, precision="f32"):
    """RefineNet.forward for evaluation (models/pipeline.py:1032-1041) on lidf_forward's data_dict:
    opt.refine_forward_times x get_pred_refine; adds pred_pos_refine and pred_depth_refine."""
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
        multires_views=opt.multires_views, roi_inp_bbox=opt.roi_inp_bbox,
        offset_range=opt.refine_offset_range, pos_rel=opt.refine_intersect_pos_type == "rel",
        pnet_pos_rel=opt.refine_pnet_pos_type == "rel", rayfeat=dd.get("rayfeat"),
        precision=precision, pnet_select=sel)
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
