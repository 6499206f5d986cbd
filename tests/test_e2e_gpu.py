"""The whole evaluation path chained once: prepare_data -> valid points ->
occupied voxels -> PointNet2Stage -> get_miss_ray -> compute_ray_aabb -> fused query -> stage 2
(2 x get_pred_refine) -> eval metrics, on a ragged geometry-derived frame (table + boxes + sphere
with holes), HIP (implicit_depth_amd.pipeline, every compute step through the C ABI) against the
oracle chain (oracle.lidf_forward + refine_step + depth_metrics)."""
import pytest
import torch

from util import TOL, make_module, make_pointnet, orc

pytestmark = pytest.mark.gpu


def _dev(batch, dev):
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}


@pytest.mark.parametrize("shape,use_all_pix,precision", [((1, 240, 320), True, "f32"),
                                                         ((1, 240, 320), True, "f16x3"),
                                                         ((2, 48, 64), False, "f32")])
def test_eval_chain_vs_oracle(cuda, shape, use_all_pix, precision):
    from implicit_depth_amd import pipeline as pl
    from implicit_depth_amd.synthetic import init_decoder_params, synthetic_batch
    B, h, w = shape
    batch, feat = synthetic_batch(B, h, w, seed=77)
    pnet_p, pnet_r = orc.init_pointnet(3, 1.5), orc.init_pointnet(4, 1.5)
    prob_p, off_p = init_decoder_params("IMNET", 385, 7, 5.0), init_decoder_params("IEF", 385, 8, 5.0)
    offr_p = init_decoder_params("IEF", 334, 9, 5.0)
    ok_ref, ref = orc.lidf_forward(batch, feat, pnet_p, prob_p, off_p)
    assert ok_ref

    opt = pl.LidfOptions(refine_use_all_pix=use_all_pix)
    pnet = make_pointnet(pnet_p, cuda)
    prob, off = make_module("IMNET", prob_p, 385, cuda), make_module("IEF", off_p, 385, cuda)
    with torch.no_grad():
        ok, dd = pl.lidf_forward(_dev(batch, cuda), feat.to(cuda), pnet, prob, off, opt, precision=precision)
    assert ok
    # --- geometry: bit-exact integers and box / slab arithmetic
    assert (dd["voxel_bound"].cpu() == ref["voxel_bound"]).all()
    assert (dd["occ_vox_bid"].cpu() == ref["occ_vox_bid"]).all()
    assert (dd["revidx"].cpu() == ref["revidx"]).all() and (dd["valid_v_pid"].cpu() == ref["valid_v_pid"]).all()
    assert (dd["valid_v_rel_coord"].cpu() == ref["valid_v_rel_coord"]).all()
    assert (dd["miss_flat_img_id"].cpu() == ref["miss_flat_img_id"]).all()
    assert (dd["miss_ray_dir"].cpu() - ref["miss_ray_dir"]).abs().max().item() <= 2e-7
    assert (dd["pair_off"].cpu().long() == ref["pair_off"]).all()
    assert (dd["pair_ray"].cpu().long() == ref["pair_ray"]).all()
    assert (dd["pair_vox"].cpu().long() == ref["pair_vox"]).all()
    # the HIP rays differ from the oracle's by <= 1 ulp (normalisation), so t_enter/t_leave are
    # compared with a tolerance here; bit-exactness on identical rays is test_boxes_gpu / g3
    assert (dd["pair_t"].cpu() - ref["pair_t"]).abs().max().item() <= 2e-6
    P, R = ref["pair_ray"].shape[0], ref["miss_ray_dir"].shape[0]
    cnt = ref["pair_off"][1:] - ref["pair_off"][:-1]
    assert cnt.min().item() >= 0 and cnt.max().item() >= 6 and P > 3 * R    # ragged, several pairs per ray
    # --- features and predictions
    assert (dd["occ_voxel_feat"].cpu() - ref["occ_voxel_feat"]).abs().max().item() <= 2e-5
    for k in ("pred_offset", "pred_prob_end", "pair_pred_pos"):
        assert (dd[k].cpu() - ref[k]).abs().max().item() <= TOL, k
    gid, rid = dd["max_pair_id"].cpu(), ref["max_pair_id"]
    same = gid == rid
    sm = ref["pred_prob_end_softmax"]
    for r in (~same).nonzero().flatten().tolist():   # only float-noise ties may pick another pair
        assert abs(sm[gid[r]] - sm[rid[r]]) <= 1e-6
    assert (~same).sum().item() <= 2
    assert (dd["pred_pos"].cpu()[same] - ref["pred_pos"][same]).abs().max().item() <= TOL
    assert R == B * h * w                                                     # mask_type 'all'
    l1 = (dd["pred_depth"].cpu() - ref["pred_depth"]).abs().reshape(-1)[same].mean().item()
    assert l1 <= TOL                                                          # depth L1 vs ref
    # --- stage 2 on the HIP stage-1 outputs, oracle fed the same stage-1 state
    pnr = make_pointnet(pnet_r, cuda)
    offr = make_module("IEF", offr_p, 334, cuda)
    with torch.no_grad():
        pl.refine_forward(dd, pnr, offr, opt, precision=precision)
    pos = dd["pred_pos"].cpu()
    sel = None
    if not use_all_pix:
        sel = (1 - ref["valid_mask"]).reshape(-1)[ref["miss_bid"] * (h * w) + ref["miss_flat_img_id"]] != 0
        assert 0 < int(sel.sum()) < R
    for _ in range(2):
        pos, ev, _ = orc.refine_step(pos, ref["miss_ray_dir"], ref["miss_img_ind"], ref["miss_bid"],
                                     ref["miss_flat_img_id"], gid, ref["pair_vox"], ref["voxel_bound"],
                                     ref["occ_vox_bid"], batch["rgb"], feat, ref["pnet_inp"], ref["revidx"],
                                     pnet_r, offr_p, ray_rgb=ref["ray_rgb"], pnet_select=sel)
    assert (dd["end_voxel_id"].cpu().long() == ev).all()
    assert (dd["pred_pos_refine"].cpu() - pos).abs().max().item() <= TOL
    # --- eval metrics of both stages vs the oracle's on its own depth maps
    gt = batch["xyz"][0, 2]
    for key, ref_depth in (("pred_depth", None), ("pred_depth_refine", None)):
        got = pl.eval_metrics(dd, key)
        want = orc.depth_metrics(dd[key][0].cpu(), gt, batch["corrupt_mask"][0, 0])
        for k in want:
            assert abs(float(got[k]) - float(want[k])) <= 2e-5 * max(1.0, abs(float(want[k]))), (key, k)


def test_early_exits(cuda):
    """The reference's three early exits (pipeline.py:671-672, :686-687, :700-701)."""
    from implicit_depth_amd import pipeline as pl
    from implicit_depth_amd.synthetic import init_decoder_params, synthetic_batch
    batch, feat = synthetic_batch(1, 24, 32, seed=5)
    pnet = make_pointnet(orc.init_pointnet(3, 1.5), cuda)
    prob = make_module("IMNET", init_decoder_params("IMNET", 385, 7, 5.0), 385, cuda)
    off = make_module("IEF", init_decoder_params("IEF", 385, 8, 5.0), 385, cuda)
    b = _dev(batch, cuda)
    # no occupied voxel: every point outside the grid
    far = dict(b)
    far["xyz_corrupt"] = b["xyz_corrupt"] + 50.0
    with pytest.raises(RuntimeError, match="inference path"):   # autograd recording: refused loudly
        pl.lidf_forward(far, feat.to(cuda), pnet, prob, off)
    torch.set_grad_enabled(False)
    ok, dd = pl.lidf_forward(far, feat.to(cuda), pnet, prob, off)
    assert not ok and dd["voxel_bound"].shape[0] == 0
    # no miss ray: mask_type 'pred' with an empty predicted mask
    opt = pl.LidfOptions(mask_type="pred")
    ok, dd = pl.lidf_forward(b, feat.to(cuda), pnet, prob, off, opt,
                             pred_mask=torch.zeros(1, 24, 32, device=cuda))
    assert not ok and dd["total_miss_sample_num"] == 0
    # no intersecting pair: one ray that passes no occupied voxel (a lone far-corner valid point)
    lone = dict(b)
    xyzc = torch.full_like(b["xyz_corrupt"], 50.0)     # everything else far outside the grid
    xyzc[0, :, 0, 0] = torch.tensor([-1.0, -1.0, 0.1], device=cuda)
    lone["xyz_corrupt"] = xyzc
    dc = torch.zeros_like(b["depth_corrupt"])
    dc[0, 0, 0, 0] = 0.1
    lone["depth_corrupt"] = dc
    pm = torch.zeros(1, 24, 32, device=cuda)
    pm[0, 12, 16] = 1
    ok, dd = pl.lidf_forward(lone, feat.to(cuda), pnet, prob, off, opt, pred_mask=pm)
    torch.set_grad_enabled(True)
    assert not ok and dd["voxel_bound"].shape[0] == 1 and dd["pair_ray"].shape[0] == 0
