"""CPU: the oracle against the golden vectors generated FROM THE REFERENCE (tests/golden/
make_golden.py). This is what pins oracle/lidf_oracle.py; the GPU tests then compare the HIP path
with the oracle and with the same fixtures."""
import os

import numpy as np
import torch

from util import closed_form, closed_form_params, orc

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return {k: v for k, v in np.load(os.path.join(G, name)).items()}


def test_g1_embed():
    g = load("g1_embed.npz")
    x = torch.from_numpy(g["x"])
    for L in (8, 4):
        got = orc.embed(x, L).numpy()
        assert got.shape == g["embed_L%d" % L].shape
        assert np.abs(got - g["embed_L%d" % L]).max() == 0.0  # same torch ops, same order: bit-equal


def g2_cases(g):
    for key in sorted(k for k in g if not k.endswith("_seed")):
        kind, d, sig = key.split("_")
        yield key, kind, int(d), bool(int(sig)), int(g[key + "_seed"])


def test_g2_decoders():
    g = load("g2_decoders.npz")
    n = 0
    for key, kind, d, sig, seed in g2_cases(g):
        p = closed_form_params(kind, d, seed=seed)
        x = closed_form((256, d), 0.5698402910, 0.1 * d, 1.0)
        got = orc.decoder_forward(p, x, kind, 2, sig).numpy()
        assert np.abs(got - g[key]).max() <= 1e-6, key
        n += 1
    assert n == 6


def g3_inputs(g):
    """Ray-major oracle/product inputs rebuilt from the reference trace."""
    h, w = [int(v) for v in g["hw"]]
    ray_dir = torch.from_numpy(g["miss_ray_dir"])
    ray_pix = torch.from_numpy(g["miss_img_ind"]).int()
    ray_bid = torch.from_numpy(g["miss_bid"]).int()
    ray_flat = torch.from_numpy(g["miss_flat_img_id"]).int()
    vb = torch.from_numpy(g["voxel_bound"])
    vbid = torch.from_numpy(g["occ_vox_bid"]).int()
    return h, w, ray_dir, ray_pix, ray_bid, ray_flat, vb, vbid


def test_g3_rays_and_pairs():
    g = load("g3_pipeline.npz")
    h, w, ray_dir, ray_pix, ray_bid, ray_flat, vb, vbid = g3_inputs(g)
    intr = torch.from_numpy(g["intr"])
    d, pix = orc.ray_dirs(intr[:, 0], intr[:, 1], intr[:, 2], intr[:, 3], h, w)
    assert np.abs(d.reshape(-1, 3).numpy() - g["miss_ray_dir"]).max() == 0.0  # mask_type 'all'
    assert (pix.reshape(-1, 2).numpy() == g["miss_img_ind"]).all()
    mask, dist = orc.ray_aabb(ray_dir.numpy(), vb.numpy(), ray_bid.numpy(), vbid.numpy())
    idx = np.argwhere(mask)  # voxel-major, the reference's nonzero order
    assert (idx[:, 0] == g["occ_vox_intersect_idx"]).all()
    assert (idx[:, 1] == g["miss_ray_intersect_idx"]).all()
    assert np.abs(dist[idx[:, 0], idx[:, 1], 0] - g["intersect_enter_dist"]).max() == 0.0
    assert np.abs(dist[idx[:, 0], idx[:, 1], 1] - g["intersect_leave_dist"]).max() == 0.0


def g3_oracle(g):
    h, w, ray_dir, ray_pix, ray_bid, ray_flat, vb, vbid = g3_inputs(g)
    mask, dist = orc.ray_aabb(ray_dir.numpy(), vb.numpy(), ray_bid.numpy(), vbid.numpy())
    pair_ray, pair_vox, pair_t, pair_off = orc.pairs_from_dense(mask, dist)
    D = 385
    res = orc.query(ray_dir, ray_pix, ray_bid, pair_ray, pair_vox, pair_t, pair_off,
                    torch.from_numpy(g["full_rgb_feat"]), torch.from_numpy(g["occ_voxel_feat"]),
                    closed_form_params("IMNET", D, seed=21), closed_form_params("IEF", D, seed=22),
                    offset_range=tuple(float(v) for v in g["offset_range"]),
                    part_size=float(g["part_size"]))
    # permutation ray-major -> reference (voxel-major) order
    R = ray_dir.shape[0]
    perm = torch.argsort(pair_vox * R + pair_ray, stable=True)
    return res, perm, (pair_ray, pair_vox, pair_t, pair_off)


def test_g3_query_matches_reference_trace():
    g = load("g3_pipeline.npz")
    res, perm, _ = g3_oracle(g)
    P = perm.shape[0]
    assert P == g["pair_pred_pos"].shape[0]
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(P)
    for k, ref in (("pred_offset", "pred_offset"), ("pred_prob_end", "pred_prob_end"),
                   ("pair_pred_pos", "pair_pred_pos"), ("pred_prob_end_softmax", "pred_prob_end_softmax")):
        got = res[k][perm].numpy()
        assert np.abs(got - g[ref]).max() <= 2e-6, k
    assert np.abs(res["pred_pos"].numpy() - g["pred_pos"]).max() <= 2e-6
    ref_id = torch.from_numpy(g["max_pair_id"])
    got_id = res["max_pair_id"]
    got_ref_order = torch.where(got_id < P, inv[got_id.clamp(max=P - 1)], torch.full_like(got_id, P))
    assert (got_ref_order == ref_id).all()
    # ROI feature and voxel feature gathers as the reference saw them (per pair)
    pr = torch.from_numpy(g["miss_ray_intersect_idx"])
    assert np.abs(res["ray_rgb"][pr].numpy() - g["intersect_rgb_feat"]).max() == 0.0


def g4_oracle(g3, g4, pnet_select=None):
    """Oracle refine iterations on the reference's stage-1 trace; returns per-iteration outputs."""
    from util import closed_form_pointnet
    h, w, ray_dir, ray_pix, ray_bid, ray_flat, vb, vbid = g3_inputs(g3)
    res, perm, (pair_ray, pair_vox, pair_t, pair_off) = g3_oracle(g3)
    Dr = int(g4["D"])
    pnet_p = closed_form_pointnet(41)
    off_p = closed_form_params("IEF", Dr, seed=31)
    pos = torch.from_numpy(g3["pred_pos"])
    outs = []
    for _ in range(2):
        pos, ev, feat = orc.refine_step(
            pos, ray_dir, ray_pix, ray_bid, ray_flat, res["max_pair_id"], pair_vox, vb, vbid,
            torch.from_numpy(g4["rgb_img"]), torch.from_numpy(g3["full_rgb_feat"]),
            torch.from_numpy(g4["valid_inp"]), torch.from_numpy(g4["valid_vox"]), pnet_p, off_p,
            offset_range=tuple(float(v) for v in g4["offset_range"]), ray_rgb=res["ray_rgb"],
            pnet_select=pnet_select)
        outs.append((pos, ev, feat))
    return outs


def test_g4_refine_matches_reference_trace():
    g3, g4 = load("g3_pipeline.npz"), load("g4_refine.npz")
    outs = g4_oracle(g3, g4)
    for i, (pos, ev, feat) in enumerate(outs, 1):
        assert np.abs(feat.numpy() - g4["occ_voxel_feat_%d" % i]).max() <= 2e-6
        assert np.abs(pos.numpy() - g4["pred_pos_refine_%d" % i]).max() <= 2e-6


def test_g7_refine_select_matches_reference_trace():
    """refine.use_all_pix = False (pipeline.py:987-996): only zero-depth pixels feed the PointNet."""
    g3, g4, g7 = load("g3_pipeline.npz"), load("g4_refine.npz"), load("g7_refine_select.npz")
    sel = torch.from_numpy(g7["inp_zero_mask"]).reshape(-1) != 0
    outs = g4_oracle(g3, g4, pnet_select=sel)
    for i, (pos, ev, feat) in enumerate(outs, 1):
        assert np.abs(feat.numpy() - g7["occ_voxel_feat_%d" % i]).max() <= 2e-6
        assert np.abs(pos.numpy() - g7["pred_pos_refine_%d" % i]).max() <= 2e-6
    assert np.abs(g7["pred_pos_refine_2"] - g4["pred_pos_refine_2"]).max() > 1e-5  # the branch matters


def test_g3_occupied_voxels():
    g = load("g3_pipeline.npz")
    res = orc.occupied_voxels(torch.from_numpy(g["valid_xyz"]), torch.from_numpy(g["valid_bid"]))
    assert abs(res["part_size"] - float(g["part_size"])) == 0.0
    for k in ("revidx", "valid_v_pid", "occ_vox_bid", "occ_vox_global_coord"):
        assert (res[k].numpy() == g[k]).all(), k
    for k in ("valid_v_rel_coord", "voxel_bound", "xmin"):
        assert np.abs(res[k].numpy() - g[k]).max() == 0.0, k


def g5_cases(g):
    for key in sorted(k[:-5] for k in g if k.endswith("_seed")):
        kind, d, n_iter, sig = key.split("_")
        yield key, kind, int(d), int(n_iter), bool(int(sig)), int(g[key + "_seed"])


def g5_inputs(d):
    x = closed_form((96, d), 0.5698402910, 0.1 * d, 1.0)
    wgt = closed_form((96, 1), 0.7390851332, 0.37, 1.0)
    return x, wgt


def test_g5_decoder_grads():
    """Autograd of the oracle's restatement against the reference modules' own gradients."""
    g = load("g5_decoder_grads.npz")
    n = 0
    for key, kind, d, n_iter, sig, seed in g5_cases(g):
        p = {k: v.clone().requires_grad_(True) for k, v in closed_form_params(kind, d, seed=seed).items()}
        x, wgt = g5_inputs(d)
        x.requires_grad_(True)
        y = orc.decoder_forward(p, x, kind, n_iter, sig)
        (y * wgt).sum().backward()
        assert np.abs(y.detach().numpy() - g[key + "_y"]).max() <= 1e-6, key
        ref = g[key + "_g_input"]
        assert np.abs(x.grad.numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), key
        for k, v in p.items():
            ref = g[key + "_g_" + k]
            assert np.abs(v.grad.numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), (key, k)
        n += 1
    assert n == 3


def test_g6_miss_ray():
    """Oracle get_miss_ray vs the reference's LIDF.get_miss_ray on float masks."""
    g = load("g6_miss_ray.npz")
    for key in ("a", "b", "c"):
        mask = torch.from_numpy(g[key + "_mask"])
        intr = torch.from_numpy(g[key + "_intr"])
        res = orc.get_miss_ray(mask, intr[:, 0], intr[:, 1], intr[:, 2], intr[:, 3])
        for k in ("miss_bid", "miss_flat_img_id", "miss_img_ind"):
            assert (res[k].numpy() == g[key + "_" + k]).all(), (key, k)
        assert np.abs(res["miss_ray_dir"].numpy() - g[key + "_miss_ray_dir"]).max() == 0.0


def test_g6_miss_ray_train_window():
    """Train flavour (pipeline.py:229-254): the random contiguous window, drawn with the
    reference's own np.random.choice calls under the same seed."""
    g = load("g6_miss_ray.npz")
    mask = torch.from_numpy(g["t_mask"])
    intr = torch.from_numpy(g["t_intr"])
    res = orc.get_miss_ray(mask, intr[:, 0], intr[:, 1], intr[:, 2], intr[:, 3])
    idx = torch.stack((res["miss_bid"], res["miss_flat_img_id"]), 1)
    np.random.seed(int(g["t_seed"]))
    sel = orc.sample_miss_window(idx, mask.shape[0], int(g["t_miss_sample_num"]))
    for k in ("miss_bid", "miss_flat_img_id", "miss_img_ind"):
        assert (res[k][sel].numpy() == g["t_" + k]).all(), k
    assert np.abs(res["miss_ray_dir"][sel].numpy() - g["t_miss_ray_dir"]).max() == 0.0


G8_KEYS = ("a1", "a2", "a3", "rmse", "rmse_log", "log10", "abs_rel", "mae", "sq_rel")


def test_g8_depth_metrics_match_reference_compute_loss():
    """The nine evaluation statistics against the reference's own LIDF.compute_loss, bs == 1 branch
    (models/pipeline.py:577-618), run on a one-frame batch whose ground truth holds NaN / inf / 0
    pixels. (cv2.resize itself stays unpinned: the generator's cv2 stub is the oracle's index rule.)"""
    g8 = load("g8_metrics.npz")
    m = orc.depth_metrics(torch.from_numpy(g8["pred_depth"]), torch.from_numpy(g8["gt_depth"]),
                          torch.from_numpy(g8["seg_mask"]), tuple(int(v) for v in g8["out_size"]))
    assert 0.0 < float(g8["a1"]) < float(g8["a2"]) < float(g8["a3"]) < 1.0   # a non-trivial frame
    assert not np.isfinite(g8["gt_depth"]).all()
    for k in G8_KEYS:
        assert abs(float(m[k]) - float(g8[k])) <= 1e-6 * max(1.0, abs(float(g8[k]))), k
