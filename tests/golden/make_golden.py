"""make_golden.py — generate the golden vectors that pin oracle/lidf_oracle.py to the REFERENCE.

Run in the authoring container only (it imports /root/reference/src, which does not exist on the
GPU box):   python tests/golden/make_golden.py
Writes small .npz fixtures next to this file. Nothing of the reference is copied: the reference's
Python modules are imported and executed, and only their inputs/outputs are stored.

  g1_embed.npz      get_embedder(8) / get_embedder(4) outputs          (models/implicit_net.py:42-57)
  g2_decoders.npz   IMNet / IEF forward outputs on closed-form weights  (models/implicit_net.py:60-152)
  g3_pipeline.npz   LIDF.get_miss_ray -> compute_ray_aabb -> get_embedding -> get_pred trace on a
                    2 x 16 x 24 synthetic batch                           (models/pipeline.py:203-466)
  g5_decoder_grads.npz  autograd gradients of the reference IMNet / IEF (input rows and every
                    parameter) for a closed-form upstream gradient   (models/implicit_net.py:60-152)
  g6_miss_ray.npz   LIDF.get_miss_ray on float masks (holes, an empty image, -0.0, tiny values)
                                                                       (models/pipeline.py:203-269)
  g7_refine_select.npz  the same two refine iterations with refine.use_all_pix = False
                                                                       (models/pipeline.py:987-996)
  g8_metrics.npz    the nine evaluation statistics of LIDF.compute_loss's bs == 1 branch (a1 .. sq_rel)
                    from the reference's own LIDF.forward(batch, 'test', 0) on a one-frame batch with
                    NaN / inf ground-truth pixels (cv2.resize stubbed: nearest-neighbour restatement)
                                                                       (models/pipeline.py:577-618)
  g4_refine.npz     RefineNet.get_pred_refine x 2 on the same batch (stage 2), incl. the refine
                    PointNet2Stage outputs                     (models/pipeline.py:922-1041, pointnet.py)

The reference's pipeline imports cv2, torchvision, torch_scatter and two JIT CUDA extensions, none
of which exist here. They are replaced by stub modules whose bodies are oracle/lidf_oracle.py's
restatements (roi_align, scatter_*, ray_aabb, pcl_aabb) — so g3 pins the reference's own Python
(ray maths, gather/concat order, scaling constants, dummy-row handling), not those third-party ops.
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/src"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, REF)

from oracle import lidf_oracle as orc  # noqa: E402
from util import closed_form, closed_form_params, closed_form_pointnet  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    def scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum"):
        assert dim == 0
        if out is not None:  # in-place max into `out` (pipeline.py:944)
            assert reduce == "max"
            out.scatter_reduce_(0, index, src, reduce="amax", include_self=True)
            return out
        n = dim_size if dim_size is not None else (int(index.max()) + 1 if index.numel() else 0)
        shape = (n,) + tuple(src.shape[1:])
        idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
        if reduce == "sum":
            return torch.zeros(shape, dtype=src.dtype).scatter_add_(0, idx, src)
        if reduce == "max":  # torch_scatter fills untouched rows with 0
            o = torch.full(shape, -float("inf"), dtype=src.dtype).scatter_reduce_(
                0, idx, src, reduce="amax", include_self=True)
            return torch.where(torch.isinf(o), torch.zeros_like(o), o)
        raise NotImplementedError(reduce)

    def scatter_softmax(src, index, dim=0, dim_size=None):
        return orc.scatter_softmax(src, index, dim_size)

    def scatter_max(src, index, dim=0, dim_size=None):
        return orc.scatter_max(src, index, dim_size)

    def scatter_log_softmax(src, index, dim=0, dim_size=None):
        return torch.log(orc.scatter_softmax(src, index, dim_size))

    def roi_align(inp, boxes, output_size, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
        return orc.roi_align(inp, boxes, output_size, spatial_scale, aligned)

    class _RayAabb:
        @staticmethod
        def forward(ray_dir, voxel_bound, ray_bid, voxel_bid):
            m, d = orc.ray_aabb(ray_dir.numpy(), voxel_bound.numpy(), ray_bid.numpy(), voxel_bid.numpy())
            return torch.from_numpy(m), torch.from_numpy(d)

    class _PclAabb:
        @staticmethod
        def forward(pos, voxel_bound, pcl_bid, voxel_bid):
            return torch.from_numpy(orc.pcl_aabb(pos.numpy(), voxel_bound.numpy(), pcl_bid.numpy(),
                                                 voxel_bid.numpy()))

    def cv2_resize(img, dsize, interpolation=None):
        # cv2.resize(img, (W, H), interpolation=cv2.INTER_NEAREST) — cv2 is not installed: the
        # oracle's restatement of OpenCV's nearest-neighbour index rule (UNPINNED against cv2)
        assert interpolation == "nearest"
        sy = orc.resize_nearest_index(img.shape[0], dsize[1])
        sx = orc.resize_nearest_index(img.shape[1], dsize[0])
        return np.ascontiguousarray(img[sy][:, sx])

    _stub("cv2", resize=cv2_resize, INTER_NEAREST="nearest")
    tv = _stub("torchvision")
    tv.ops = _stub("torchvision.ops", roi_align=roi_align)
    tv.transforms = _stub("torchvision.transforms")
    _stub("torch_scatter", scatter=scatter, scatter_softmax=scatter_softmax, scatter_max=scatter_max,
          scatter_log_softmax=scatter_log_softmax)
    _stub("extensions")
    _stub("extensions.ray_aabb")
    _stub("extensions.ray_aabb.jit", ray_aabb=_RayAabb)
    _stub("extensions.pcl_aabb")
    _stub("extensions.pcl_aabb.jit", pcl_aabb=_PclAabb)


def g1_embed():
    import models.implicit_net as ref
    x = torch.cat((torch.tensor([[0.0, 0.0, 0.0], [1e-7, -1e-7, 2.3], [-2.3, 2.3, -1.0],
                                 [0.5, 0.25, 0.125]]),
                   closed_form((60, 3), 0.7548776662, 0.3, 2.3)), 0)
    out = {"x": x.numpy()}
    for L in (8, 4):
        fn, dim = ref.get_embedder(L)
        y = fn(x)
        assert y.shape[1] == dim
        out["embed_L%d" % L] = y.numpy()
    np.savez_compressed(os.path.join(HERE, "g1_embed.npz"), **out)


def g5_decoder_grads():
    """The reference modules' own autograd: what the training path of the product must return."""
    import models.implicit_net as ref
    out = {}
    def min_abs_preact(p, x, kind, n_iter):
        """Smallest |pre-activation| of the three leaky-ReLU layers over all rows and passes (f64):
        a fixture with one of them within rounding of 0 would pin a coin flip of the f32 summation
        order (the derivative jumps there), not the algorithm."""
        pd = {k: v.double() for k, v in p.items()}
        xd = x.detach().double()
        off = torch.full((xd.shape[0], 1), 0.001, dtype=torch.float64)
        lo = 1e9
        for _ in range(n_iter):
            h = torch.cat([xd, off @ pd["offset_enc.weight"].t() + pd["offset_enc.bias"]], 1) if kind == "IEF" else xd
            for i in (1, 2, 3):
                z = h @ pd["linear_%d.weight" % i].t() + pd["linear_%d.bias" % i]
                lo = min(lo, z.abs().min().item())
                h = torch.nn.functional.leaky_relu(z, 0.02)
            off = off + h @ pd["linear_4.weight"].t() + pd["linear_4.bias"]
        return lo

    for kind, d, n_iter, sig in (("IMNET", 385, 1, False), ("IEF", 385, 2, False), ("IEF", 334, 3, True)):
        seed = 51 + 10 * len(out)
        while True:
            p = closed_form_params(kind, d, seed=seed)
            x = closed_form((96, d), 0.5698402910, 0.1 * d, 1.0)
            if min_abs_preact(p, x, kind, n_iter) > 2e-5:
                break
            seed += 1
        x.requires_grad_(True)
        wgt = closed_form((96, 1), 0.7390851332, 0.37, 1.0)
        if kind == "IEF":
            m = ref.IEF(torch.device("cpu"), d, 1, 64, n_iter=n_iter, use_sigmoid=sig)
        else:
            m = ref.IMNet(d, 1, 64, use_sigmoid=sig)
        m.load_state_dict(p)
        y = m(x)
        (y * wgt).sum().backward()
        key = "%s_%d_%d_%d" % (kind, d, n_iter, int(sig))
        out[key + "_seed"] = np.int64(seed)
        out[key + "_y"] = y.detach().numpy()
        out[key + "_g_input"] = x.grad.numpy()
        for k, v in m.named_parameters():
            out[key + "_g_" + k] = v.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "g5_decoder_grads.npz"), **out)


def g2_decoders():
    import models.implicit_net as ref
    out = {}
    cases = [("IMNET", 385, False), ("IEF", 385, False), ("IEF", 334, False), ("IMNET", 385, True),
             ("IEF", 385, True), ("IMNET", 265, False)]
    for kind, d, sig in cases:
        p = closed_form_params(kind, d, seed=len(out) + 1)
        x = closed_form((256, d), 0.5698402910, 0.1 * d, 1.0)
        if kind == "IEF":
            m = ref.IEF(torch.device("cpu"), d, 1, 64, n_iter=2, use_sigmoid=sig)
        else:
            m = ref.IMNet(d, 1, 64, use_sigmoid=sig)
        m.load_state_dict(p)
        with torch.no_grad():
            y = m(x)
        key = "%s_%d_%d" % (kind, d, int(sig))
        out[key] = y.numpy()
        out[key + "_seed"] = np.array(len(out))  # bookkeeping: seed used for the params
    np.savez_compressed(os.path.join(HERE, "g2_decoders.npz"), **out)


def g3_pipeline():
    install_stubs()
    import models.pipeline as pl
    from opt import Params
    cfg = os.path.join(REF, "experiments", "implicit_depth")
    opt = Params(os.path.join(cfg, "default_config.yaml"))
    opt.update(os.path.join(cfg, "test_lidf.yaml"))
    opt.grid.valid_sample_num = -1  # all valid points: no random sub-sampling in the trace
    torch.manual_seed(1234)
    dev = torch.device("cpu")
    lidf = pl.LIDF(opt, dev).eval()
    D = lidf.prob_dec.inp_dim
    lidf.prob_dec.load_state_dict(closed_form_params("IMNET", D, seed=21))
    lidf.offset_dec.load_state_dict(closed_form_params("IEF", D, seed=22))
    B, h, w = 2, 16, 24
    fx = torch.tensor([21.6, 20.0], dtype=torch.float64)
    fy = torch.tensor([21.6, 22.0], dtype=torch.float64)
    cx = torch.tensor([11.5, 12.25], dtype=torch.float64)
    cy = torch.tensor([7.5, 7.0], dtype=torch.float64)
    d, _ = orc.ray_dirs(fx.float(), fy.float(), cx.float(), cy.float(), h, w)  # [B,h,w,3]
    ys, xs = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
    depth = torch.stack((0.9 + 0.01 * xs + 0.005 * ys, 1.3 - 0.012 * xs + 0.02 * ys), 0)  # [B,h,w]
    xyz = (d / d[..., 2:3] * depth.unsqueeze(-1)).permute(0, 3, 1, 2).contiguous()  # [B,3,h,w]
    hole = torch.zeros(B, 1, h, w)
    hole[0, :, 5:11, 8:17] = 1
    hole[1, :, 2:9, 3:12] = 1
    depth_corrupt = depth.unsqueeze(1) * (1 - hole)
    xyz_corrupt = xyz * (1 - hole)
    batch = {
        "rgb": closed_form((B, 3, h, w), 0.3819660113, 0.2, 1.5),
        "xyz": xyz, "xyz_corrupt": xyz_corrupt, "depth_corrupt": depth_corrupt,
        "corrupt_mask": hole.clone(), "valid_mask": 1 - hole,
        "fx": fx, "fy": fy, "cx": cx, "cy": cy, "item_path": ["a", "b"],
    }
    cap = {}
    hk = lidf.pnet_model.register_forward_hook(lambda m, i, o: cap.__setitem__("occ_voxel_feat", o.detach().clone()))
    with torch.no_grad():
        dd = lidf.prepare_data(batch, "test", None)
        lidf.get_valid_points(dd)
        assert lidf.get_occ_vox_bound(dd)
        lidf.get_miss_ray(dd, "test")
        assert lidf.compute_ray_aabb(dd)
        lidf.get_embedding(dd)
        lidf.get_pred(dd, "test", 0)
    hk.remove()
    keep = ["miss_bid", "miss_flat_img_id", "miss_ray_dir", "miss_img_ind", "voxel_bound", "occ_vox_bid",
            "occ_vox_intersect_idx", "miss_ray_intersect_idx", "intersect_enter_dist",
            "intersect_leave_dist", "intersect_enter_pos", "full_rgb_feat", "intersect_rgb_feat",
            "intersect_voxel_feat", "pair_pred_pos", "max_pair_id", "pred_prob_end",
            "pred_prob_end_softmax", "pred_pos"]
    out = {k: dd[k].detach().numpy() for k in keep}
    out["occ_voxel_feat"] = cap["occ_voxel_feat"].numpy()
    for k in ("valid_xyz", "valid_bid", "revidx", "valid_v_pid", "valid_v_rel_coord",
              "occ_vox_global_coord", "xmin"):
        out[k] = dd[k].detach().numpy()
    out["part_size"] = np.float32(dd["part_size"])
    out["intr"] = torch.stack((dd["fx"], dd["fy"], dd["cx"], dd["cy"]), 1).numpy()
    out["hw"] = np.array([h, w])
    out["offset_range"] = np.array(opt.grid.offset_range, dtype=np.float32)
    # the decoders' direct outputs on the reference's own inp_embed (pred_offset is not in data_dict)
    with torch.no_grad():
        inp = torch.cat((dd["intersect_voxel_feat"], dd["intersect_rgb_feat"],
                         dd["intersect_enter_pos_embed"], dd["intersect_leave_pos_embed"],
                         dd["intersect_dir_embed"]), -1)
        out["pred_offset"] = lidf.offset_dec(inp).numpy()
    print("g3: R=%d V=%d P=%d" % (out["miss_ray_dir"].shape[0], out["voxel_bound"].shape[0],
                                   out["pair_pred_pos"].shape[0]))
    np.savez_compressed(os.path.join(HERE, "g3_pipeline.npz"), **out)

    # ---- g4: stage 2 on the same data_dict
    opt2 = Params(os.path.join(cfg, "default_config.yaml"))
    opt2.update(os.path.join(cfg, "test_refine.yaml"))
    refine = pl.RefineNet(opt2, dev).eval()
    Dr = refine.offset_dec.inp_dim
    refine.offset_dec.load_state_dict(closed_form_params("IEF", Dr, seed=31))
    refine.pnet_model.load_state_dict(closed_form_pointnet(41))
    feats = []
    hk = refine.pnet_model.register_forward_hook(lambda m, i, o: feats.append(o.detach().clone()))
    with torch.no_grad():
        p1 = refine.get_pred_refine(dd, dd["pred_pos"], "test", 0)
        p2 = refine.get_pred_refine(dd, p1, "test", 1)
    hk.remove()
    valid_v_rgb = dd["valid_rgb"][dd["valid_v_pid"]]
    g4 = {
        "rgb_img": dd["rgb_img"].numpy(),
        "valid_inp": torch.cat((dd["valid_v_rel_coord"], valid_v_rgb), -1).numpy(),
        "valid_vox": dd["revidx"].numpy(),
        "pred_pos_refine_1": p1.numpy(), "pred_pos_refine_2": p2.numpy(),
        "occ_voxel_feat_1": feats[0].numpy(), "occ_voxel_feat_2": feats[1].numpy(),
        "offset_range": np.array(opt2.refine.offset_range, dtype=np.float32),
        "D": np.array(Dr),
    }
    print("g4: Nv=%d Dr=%d" % (g4["valid_inp"].shape[0], Dr))
    np.savez_compressed(os.path.join(HERE, "g4_refine.npz"), **g4)

    # ---- g7: the same two iterations with refine.use_all_pix = False (pipeline.py:987-996): only
    # the predicted points of zero-depth input pixels are fed back into the PointNet
    opt2.refine.use_all_pix = False
    feats = []
    hk = refine.pnet_model.register_forward_hook(lambda m, i, o: feats.append(o.detach().clone()))
    with torch.no_grad():
        q1 = refine.get_pred_refine(dd, dd["pred_pos"], "test", 0)
        q2 = refine.get_pred_refine(dd, q1, "test", 1)
    hk.remove()
    g7 = {"inp_zero_mask": (1 - dd["valid_mask"]).numpy(), "pred_pos_refine_1": q1.numpy(),
          "pred_pos_refine_2": q2.numpy(), "occ_voxel_feat_1": feats[0].numpy(),
          "occ_voxel_feat_2": feats[1].numpy()}
    print("g7: selected pixels = %d of %d" % (int(g7["inp_zero_mask"].sum()), g7["inp_zero_mask"].size))
    np.savez_compressed(os.path.join(HERE, "g7_refine_select.npz"), **g7)


def g8_metrics():
    """The reference's LIDF.forward(batch, 'test', 0) on ONE frame (bs == 1 selects the resized
    ClearGrasp statistics, models/pipeline.py:577-618): the depth maps that enter the statistics and
    the nine numbers compute_loss returns. Ground truth carries a NaN and an inf inside the hole."""
    install_stubs()
    import models.pipeline as pl
    from opt import Params
    cfg = os.path.join(REF, "experiments", "implicit_depth")
    opt = Params(os.path.join(cfg, "default_config.yaml"))
    opt.update(os.path.join(cfg, "test_lidf.yaml"))
    opt.grid.valid_sample_num = -1
    torch.manual_seed(4321)
    dev = torch.device("cpu")
    lidf = pl.LIDF(opt, dev).eval()
    D = lidf.prob_dec.inp_dim
    lidf.prob_dec.load_state_dict(closed_form_params("IMNET", D, seed=81))
    lidf.offset_dec.load_state_dict(closed_form_params("IEF", D, seed=82))
    B, h, w = 1, 45, 60
    fx, fy = torch.tensor([54.0], dtype=torch.float64), torch.tensor([54.0], dtype=torch.float64)
    cx, cy = torch.tensor([29.5], dtype=torch.float64), torch.tensor([22.0], dtype=torch.float64)
    d, _ = orc.ray_dirs(fx.float(), fy.float(), cx.float(), cy.float(), h, w)
    ys, xs = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
    depth = (0.8 + 0.004 * xs + 0.006 * ys).unsqueeze(0)
    xyz = (d / d[..., 2:3] * depth.unsqueeze(-1)).permute(0, 3, 1, 2).contiguous()
    hole = torch.zeros(B, 1, h, w)
    hole[0, :, 10:34, 14:47] = 1
    hole[0, :, 2:6, 50:58] = 1
    xyz_gt = xyz.clone()
    xyz_gt[0, 2, 12, 20] = float("nan")      # non-finite ground truth counts as 0 (pipeline.py:582-583)
    xyz_gt[0, 2, 30, 40] = float("inf")
    xyz_gt[0, 2, 3, 52] = 0.0                # and a true zero
    batch = {
        "rgb": closed_form((B, 3, h, w), 0.3819660113, 0.2, 1.5),
        "xyz": xyz_gt, "xyz_corrupt": xyz * (1 - hole), "depth_corrupt": depth.unsqueeze(1) * (1 - hole),
        "corrupt_mask": hole.clone(), "valid_mask": 1 - hole,
        "fx": fx, "fy": fy, "cx": cx, "cy": cy, "item_path": ["a"],
    }
    with torch.no_grad():
        ok, dd, loss = lidf(batch, "test", 0)
    assert ok
    pred_xyz = dd["xyz_corrupt_flat"].clone()
    pred_xyz[dd["miss_bid"], dd["miss_flat_img_id"]] = dd["pred_pos"]
    out = {
        "pred_depth": pred_xyz.reshape(B, h, w, 3)[0, :, :, 2].numpy(),
        "gt_depth": dd["xyz_flat"].reshape(B, h, w, 3)[0, :, :, 2].numpy(),
        "seg_mask": dd["corrupt_mask"][0].numpy().astype(np.uint8),
        "out_size": np.array([144, 256]),
    }
    for k in ("a1", "a2", "a3", "rmse", "rmse_log", "log10", "abs_rel", "mae", "sq_rel"):
        out[k] = np.float32(loss[k].item())
    print("g8:", {k: float(out[k]) for k in ("a1", "a2", "a3", "rmse", "mae")},
          "rays", dd["total_miss_sample_num"])
    np.savez_compressed(os.path.join(HERE, "g8_metrics.npz"), **out)


def g6_miss_ray():
    """LIDF.get_miss_ray of the reference on float masks with holes, an empty image and odd values
    (models/pipeline.py:203-269, eval flavour)."""
    install_stubs()
    import models.pipeline as pl
    from opt import Params
    cfg = os.path.join(REF, "experiments", "implicit_depth")
    opt = Params(os.path.join(cfg, "default_config.yaml"))
    opt.update(os.path.join(cfg, "test_lidf.yaml"))
    lidf = pl.LIDF(opt, torch.device("cpu")).eval()
    out = {}
    cases = {"a": (3, 10, 13), "b": (2, 33, 47), "c": (1, 5, 1030)}
    for key, (B, h, w) in cases.items():
        g = torch.Generator().manual_seed(600 + B * h)
        mask = (torch.rand(B, h, w, generator=g) < 0.37).float()
        mask = mask * (torch.rand(B, h, w, generator=g) * 4 - 2)      # non-zero values of both signs
        if B > 1:
            mask[1] = 0                                                # an image without miss rays
        mask[0, 0, 0] = -0.0                                           # negative zero is zero
        mask[0, h - 1, w - 1] = 1e-30
        fx = torch.rand(B, generator=g) * 10 + 0.9 * w
        fy = fx * (1 + 0.05 * torch.rand(B, generator=g))
        cx = torch.rand(B, generator=g) + w / 2 - 0.5
        cy = torch.rand(B, generator=g) + h / 2 - 0.5
        dd = {"bs": B, "h": h, "w": w, "fx": fx, "fy": fy, "cx": cx, "cy": cy, "pred_mask": mask}
        with torch.no_grad():
            lidf.get_miss_ray(dd, "test")
        out[key + "_mask"] = mask.numpy()
        out[key + "_intr"] = torch.stack((fx, fy, cx, cy), 1).numpy()
        for k in ("miss_bid", "miss_flat_img_id", "miss_ray_dir", "miss_img_ind"):
            out[key + "_" + k] = dd[k].numpy()
        print("g6", key, "R =", dd["total_miss_sample_num"])
    # train flavour: corrupt_mask + the random contiguous window (pipeline.py:229-254) with the
    # reference's own np.random.choice calls, seeded
    B, h, w = 3, 12, 16
    g = torch.Generator().manual_seed(611)
    mask = (torch.rand(B, h, w, generator=g) < 0.5).float()
    mask[2, 2:, :] = 0                                                # an image with fewer rays than the window
    fx = torch.full((B,), 14.0)
    cx, cy = torch.full((B,), 7.5), torch.full((B,), 5.5)
    lidf.opt.grid.miss_sample_num = 20
    dd = {"bs": B, "h": h, "w": w, "fx": fx, "fy": fx, "cx": cx, "cy": cy, "corrupt_mask": mask}
    np.random.seed(12345)
    with torch.no_grad():
        lidf.get_miss_ray(dd, "train")
    out["t_mask"] = mask.numpy()
    out["t_intr"] = torch.stack((fx, fx, cx, cy), 1).numpy()
    out["t_seed"] = np.int64(12345)
    out["t_miss_sample_num"] = np.int64(20)
    for k in ("miss_bid", "miss_flat_img_id", "miss_ray_dir", "miss_img_ind"):
        out["t_" + k] = dd[k].numpy()
    print("g6 train window: R =", dd["total_miss_sample_num"])
    np.savez_compressed(os.path.join(HERE, "g6_miss_ray.npz"), **out)


if __name__ == "__main__":
    if "--only-g6" in sys.argv:
        g6_miss_ray()
        sys.exit(0)
    if "--only-g8" in sys.argv:
        g8_metrics()
        sys.exit(0)
    g1_embed()
    g2_decoders()
    g5_decoder_grads()
    g3_pipeline()
    g6_miss_ray()
    g8_metrics()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
