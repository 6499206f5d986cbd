"""GPU: stage-2 refinement (PointNet2Stage + get_pred_refine) against the oracle and against the
golden trace of the reference's RefineNet (tests/golden/g4_refine.npz)."""
import numpy as np
import pytest
import torch

from test_oracle_golden import g3_inputs, g3_oracle, g4_oracle, load
from util import TOL, closed_form_params, closed_form_pointnet, make_module, make_pointnet, orc, to_dev

pytestmark = pytest.mark.gpu


# (70000, 50): more 128-point tiles than workgroups — a wavefront walks several tiles and the weight
# ring wraps; (40, 300) and (90000, 400): a voxel table beyond the LDS pooling path (copies of the
# table + atomic maxima); the others pool through the per-workgroup LDS tables
@pytest.mark.parametrize("n,v", [(5000, 60), (131, 7), (1, 1), (40, 300), (70000, 50), (90000, 400)])
def test_pointnet_vs_oracle(cuda, n, v):
    g = torch.Generator().manual_seed(n + v)
    p = orc.init_pointnet(7, 1.5)
    inp = torch.randn(n, 6, generator=g)
    vox = torch.randint(0, v, (n,), generator=g)
    ref = orc.pointnet2stage(p, inp, vox, v)
    m = make_pointnet(p, cuda)
    with torch.no_grad():
        got = m(inp.to(cuda), vox.to(cuda), n_vox=v).cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() <= 2e-5, (got - ref).abs().max().item()
    # voxels without points: the reference's scatter-max leaves 0 -> relu(vox_lin(0)) rows
    empty = torch.bincount(vox, minlength=v) == 0
    if empty.any():
        assert (got[empty] - ref[empty]).abs().max().item() <= 2e-5


def test_pointnet_without_points(cuda):
    """No point at all: every voxel keeps the scatter-max's 0 -> out = relu(vox_lin2(relu(...)(0)))
    rows, the same for every voxel (nothing is launched over the points)."""
    p = orc.init_pointnet(7, 1.5)
    ref = orc.pointnet2stage(p, torch.zeros(0, 6), torch.zeros(0, dtype=torch.long), 5)
    m = make_pointnet(p, cuda)
    with torch.no_grad():
        got = m(torch.zeros(0, 6, device=cuda), torch.zeros(0, dtype=torch.long, device=cuda), n_vox=5).cpu()
    assert got.shape == (5, 128) and (got - ref).abs().max().item() <= 2e-5


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_refine_golden_hip(cuda, precision):
    from implicit_depth_amd.query import compute_ray_aabb, lidf_query, lidf_refine
    g3, g4 = load("g3_pipeline.npz"), load("g4_refine.npz")
    h, w, ray_dir, ray_pix, ray_bid, ray_flat, vb, vbid = g3_inputs(g3)
    dev = cuda
    rd = ray_dir.to(dev)
    off, pr, pv, pt = compute_ray_aabb(rd, vb.to(dev), ray_bid.to(dev), vbid.to(dev))
    D = 385
    prob = make_module("IMNET", closed_form_params("IMNET", D, seed=21), D, dev)
    offd = make_module("IEF", closed_form_params("IEF", D, seed=22), D, dev)
    feat_grid = torch.from_numpy(g3["full_rgb_feat"]).to(dev)
    with torch.no_grad():
        s1 = lidf_query(rd, ray_pix.to(dev), ray_bid.to(dev), off, pr, pv, pt, feat_grid,
                        torch.from_numpy(g3["occ_voxel_feat"]).to(dev), prob, offd)
    Dr = int(g4["D"])
    pnet = make_pointnet(closed_form_pointnet(41), dev)
    offr = make_module("IEF", closed_form_params("IEF", Dr, seed=31), Dr, dev)
    args = (rd, ray_pix.to(dev), ray_bid.to(dev), ray_flat.to(dev), s1["pred_pos"], s1["max_pair_id"],
            pv, vb.to(dev), vbid.to(dev), torch.from_numpy(g4["rgb_img"]).to(dev), feat_grid,
            torch.from_numpy(g4["valid_inp"]).to(dev),
            torch.from_numpy(g4["valid_vox"]).int().to(dev), pnet, offr)
    kw = dict(offset_range=tuple(float(v) for v in g4["offset_range"]), precision=precision)
    with torch.no_grad():
        p1, ev1 = lidf_refine(*args, forward_times=1, **kw)
        p2, ev2 = lidf_refine(*args, forward_times=2, **kw)
    assert np.abs(p1.cpu().numpy() - g4["pred_pos_refine_1"]).max() <= TOL
    assert np.abs(p2.cpu().numpy() - g4["pred_pos_refine_2"]).max() <= TOL
    outs = g4_oracle(g3, g4)
    assert (ev1.cpu().long() == outs[0][1]).all()  # end voxel ids: exact


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_refine_select_golden_hip(cuda, precision):
    """refine.use_all_pix = False (pipeline.py:987-996) against the reference's own trace g7."""
    from implicit_depth_amd.query import compute_ray_aabb, lidf_query, lidf_refine
    g3, g4, g7 = load("g3_pipeline.npz"), load("g4_refine.npz"), load("g7_refine_select.npz")
    h, w, ray_dir, ray_pix, ray_bid, ray_flat, vb, vbid = g3_inputs(g3)
    dev = cuda
    rd = ray_dir.to(dev)
    off, pr, pv, pt = compute_ray_aabb(rd, vb.to(dev), ray_bid.to(dev), vbid.to(dev))
    prob = make_module("IMNET", closed_form_params("IMNET", 385, seed=21), 385, dev)
    offd = make_module("IEF", closed_form_params("IEF", 385, seed=22), 385, dev)
    feat_grid = torch.from_numpy(g3["full_rgb_feat"]).to(dev)
    with torch.no_grad():
        s1 = lidf_query(rd, ray_pix.to(dev), ray_bid.to(dev), off, pr, pv, pt, feat_grid,
                        torch.from_numpy(g3["occ_voxel_feat"]).to(dev), prob, offd)
    Dr = int(g4["D"])
    pnet = make_pointnet(closed_form_pointnet(41), dev)
    offr = make_module("IEF", closed_form_params("IEF", Dr, seed=31), Dr, dev)
    sel = torch.from_numpy(g7["inp_zero_mask"]).to(dev)     # [B,h,w] float, rays = all pixels in order
    with torch.no_grad():
        p2, _ = lidf_refine(rd, ray_pix.to(dev), ray_bid.to(dev), ray_flat.to(dev), s1["pred_pos"],
                            s1["max_pair_id"], pv, vb.to(dev), vbid.to(dev),
                            torch.from_numpy(g4["rgb_img"]).to(dev), feat_grid,
                            torch.from_numpy(g4["valid_inp"]).to(dev),
                            torch.from_numpy(g4["valid_vox"]).int().to(dev), pnet, offr, forward_times=2,
                            offset_range=tuple(float(v) for v in g4["offset_range"]),
                            precision=precision, pnet_select=sel)
    assert np.abs(p2.cpu().numpy() - g7["pred_pos_refine_2"]).max() <= TOL
    assert np.abs(p2.cpu().numpy() - g4["pred_pos_refine_2"]).max() > 1e-5   # not the all-pixel result


def test_pointnet_negative_index_rows_are_left_out(cuda):
    g = torch.Generator().manual_seed(5)
    p = orc.init_pointnet(7, 1.5)
    inp = torch.randn(700, 6, generator=g)
    vox = torch.randint(0, 9, (700,), generator=g)
    drop = torch.rand(700, generator=g) < 0.4
    ref = orc.pointnet2stage(p, inp[~drop], vox[~drop], 9)
    m = make_pointnet(p, cuda)
    v2 = vox.clone()
    v2[drop] = -1
    with torch.no_grad():
        got = m(inp.to(cuda), v2.to(cuda), n_vox=9).cpu()
    assert (got - ref).abs().max().item() <= 2e-5


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_refine_synthetic_vs_oracle(cuda, precision):
    """Stage 1 + 2 on a synthetic frame: rays without pairs (dummy voxel 0), relative positions."""
    from implicit_depth_amd.query import lidf_query, lidf_refine
    scene = orc.synthetic_scene(2, 12, 16, 6, seed=21, ragged=True)
    s = to_dev(scene, cuda)
    D = scene["D"]
    prob, off = make_module("IMNET", scene["prob_p"], D, cuda), make_module("IEF", scene["off_p"], D, cuda)
    with torch.no_grad():
        s1 = lidf_query(s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"],
                        s["pair_vox"], s["pair_t"], s["feat_grid"], s["vox_feat"], prob, off)
    g = torch.Generator().manual_seed(3)
    V = scene["V"]
    half = 0.125
    vb = torch.cat((scene["vox_center"] - half, scene["vox_center"] + half), 1)
    vbid = torch.arange(2).repeat_interleave(729).int()
    rgb = torch.randn(2, 3, 12, 16, generator=g)
    Nv = 500
    valid_inp = torch.randn(Nv, 6, generator=g) * 0.2
    valid_vox = torch.randint(0, V, (Nv,), generator=g).int()
    pnet_p = orc.init_pointnet(5, 1.5)
    Dr = 334
    offr_p = orc.randomize_biases(orc.init_decoder("IEF", Dr, 77, 5.0), 78)
    ref_pos = s1["pred_pos"].cpu()
    mid = s1["max_pair_id"].cpu()
    for pos_rel in (False, True):
        pos = ref_pos
        for _ in range(2):
            pos, ev, _ = orc.refine_step(pos, scene["ray_dir"], scene["ray_pix"], scene["ray_bid"],
                                         scene["ray_flat"], mid, scene["pair_vox"], vb, vbid, rgb,
                                         scene["feat_grid"], valid_inp, valid_vox, pnet_p, offr_p,
                                         pos_rel=pos_rel)
        with torch.no_grad():
            got, gev = lidf_refine(s["ray_dir"], s["ray_pix"], s["ray_bid"], s["ray_flat"],
                                   s1["pred_pos"], s1["max_pair_id"], s["pair_vox"], vb.to(cuda),
                                   vbid.to(cuda), rgb.to(cuda), s["feat_grid"], valid_inp.to(cuda),
                                   valid_vox.to(cuda), make_pointnet(pnet_p, cuda),
                                   make_module("IEF", offr_p, Dr, cuda), pos_rel=pos_rel,
                                   precision=precision)
        assert (got.cpu() - pos).abs().max().item() <= TOL, pos_rel
        assert (gev.cpu().long() == ev).all()


def test_refine_packed_weights_follow_parameter_updates(cuda):
    """lidf_refine keeps the IEF's and the PointNet's packed weight streams per module and
    re-validates them on the device (lidf_refine_pack_guarded_f32 / lidf_pointnet_pack_guarded_f32):
    a second call reuses them, in-place updates and writes through `.data` are picked up."""
    from implicit_depth_amd import _lib
    from implicit_depth_amd.query import lidf_query, lidf_refine
    scene = orc.synthetic_scene(1, 12, 16, 6, seed=23, ragged=True)
    s = to_dev(scene, cuda)
    D = scene["D"]
    prob, off = make_module("IMNET", scene["prob_p"], D, cuda), make_module("IEF", scene["off_p"], D, cuda)
    with torch.no_grad():
        s1 = lidf_query(s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"],
                        s["pair_vox"], s["pair_t"], s["feat_grid"], s["vox_feat"], prob, off)
    g = torch.Generator().manual_seed(4)
    vb = torch.cat((scene["vox_center"] - 0.125, scene["vox_center"] + 0.125), 1)
    vbid = torch.zeros(729, dtype=torch.int32)
    rgb = torch.randn(1, 3, 12, 16, generator=g)
    valid_inp = torch.randn(300, 6, generator=g) * 0.2
    valid_vox = torch.randint(0, scene["V"], (300,), generator=g).int()
    pnet_p = orc.init_pointnet(5, 1.5)
    offr_p = orc.randomize_biases(orc.init_decoder("IEF", 334, 77, 5.0), 78)
    pnet, offr = make_pointnet(pnet_p, cuda), make_module("IEF", offr_p, 334, cuda)

    def run(mod):
        with torch.no_grad():   # `pnet` is looked up at call time (the test swaps it below)
            return lidf_refine(s["ray_dir"], s["ray_pix"], s["ray_bid"], s["ray_flat"], s1["pred_pos"],
                               s1["max_pair_id"], s["pair_vox"], vb.to(cuda), vbid.to(cuda), rgb.to(cuda),
                               s["feat_grid"], valid_inp.to(cuda), valid_vox.to(cuda), pnet, mod)[0]
    from test_parity_gpu import _guard_state
    a = run(offr)
    (entry,), (pentry,) = _lib.packed_entries(_lib.PACK_CACHE_REFINE, offr), _lib.packed_entries(_lib.PACK_CACHE, pnet)
    assert _guard_state(entry)[1:] == (1, 1) and _guard_state(pentry)[2] == 1
    b = run(offr)
    assert _lib.packed_entries(_lib.PACK_CACHE_REFINE, offr) == [entry] and torch.equal(a, b)
    assert _guard_state(entry)[1] == 0 and _guard_state(pentry)[1] == 0     # nothing re-packed
    with torch.no_grad():
        offr.linear_1.weight.mul_(1.5)          # in place, as an optimizer step
    offr.linear_3.bias.data.add_(0.05)          # through .data: no version bump
    pnet.point_lin3.weight.data.mul_(0.9)
    pnet.vox_lin1.bias.data.copy_(pnet.vox_lin1.bias.data + 0.01)
    c = run(offr)
    assert _guard_state(entry)[1] == 1 and _guard_state(pentry)[1] == 1
    p2 = {k: v.clone() for k, v in offr_p.items()}
    p2["linear_1.weight"] = p2["linear_1.weight"] * 1.5
    p2["linear_3.bias"] = p2["linear_3.bias"] + 0.05
    pn2 = {k: v.clone() for k, v in pnet_p.items()}
    pn2["point_lin3.weight"] = pn2["point_lin3.weight"] * 0.9
    pn2["vox_lin1.bias"] = pn2["vox_lin1.bias"] + 0.01
    pnet_old, pnet = pnet, make_pointnet(pn2, cuda)
    d = run(make_module("IEF", p2, 334, cuda))   # fresh modules with the updated parameters
    assert torch.equal(c, d) and (c - a).abs().max().item() > 1e-5
    pos = s1["pred_pos"].cpu()
    for _ in range(2):
        pos, _, _ = orc.refine_step(pos, scene["ray_dir"], scene["ray_pix"], scene["ray_bid"], scene["ray_flat"],
                                    s1["max_pair_id"].cpu(), scene["pair_vox"], vb, vbid, rgb, scene["feat_grid"],
                                    valid_inp, valid_vox, pn2, p2)
    assert (c.cpu() - pos).abs().max().item() <= TOL
    # the stand-alone PointNet2Stage.forward goes through the same guarded cache
    x, idx = valid_inp.to(cuda), valid_vox.to(cuda)
    with torch.no_grad():
        f0 = pnet_old(x, idx, n_vox=scene["V"])
        pnet_old.point_lin1.weight.data.mul_(1.2)
        f1 = pnet_old(x, idx, n_vox=scene["V"])
    pn3 = {k: v.clone() for k, v in pn2.items()}
    pn3["point_lin1.weight"] = pn3["point_lin1.weight"] * 1.2
    ref = orc.pointnet2stage(pn3, valid_inp, valid_vox.long(), scene["V"])
    assert (f1.cpu() - ref).abs().max().item() <= 2e-5 and (f1 - f0).abs().max().item() > 1e-5


@pytest.mark.parametrize("frames,occupancy", [(1, 0.1), (3, 0.4), (2, 1.0)])
def test_refine_end_voxel_through_the_cell_table(cuda, frames, occupancy):
    """LidfRefineArgs.voxel_coord / lidf_refine(grid=): the end voxel of a ray looked up in the cell table of
    get_occ_vox_bound's grid instead of testing every ray against every voxel (the reference's pcl_aabb +
    scatter max, models/pipeline.py:939-944). Same ids and — the rest of the iteration being the same kernels —
    bit-identical positions, on points inside cells, exactly on cell faces / edges / corners (inclusive bounds:
    the larger voxel wins), outside the grid, far away, +-inf and NaN (a NaN coordinate fails no comparison of
    the reference's predicate: "inside" every voxel of its image), rays without a pair, random arg-max pairs."""
    from implicit_depth_amd.query import get_occ_vox_bound, lidf_refine
    g = torch.Generator().manual_seed(100 + frames)
    h, w = 24, 32
    # occupied cells: random valid points (occupancy 1.0: a point in every cell)
    res, part = 9, 0.25
    lo = torch.tensor([-1.125, -1.125, -0.125])
    ncell = frames * res ** 3
    cells = torch.nonzero(torch.rand(ncell, generator=g) < occupancy)[:, 0]
    if cells.numel() == 0:
        cells = torch.tensor([5])
    cb, cr = cells // res ** 3, cells % res ** 3
    cxyz = torch.stack((cr // 81, (cr // 9) % 9, cr % 9), 1).float()
    pts = lo + (cxyz + 0.2 + 0.6 * torch.rand(cells.numel(), 3, generator=g)) * part
    occ = get_occ_vox_bound(pts.to(cuda), cb.int().to(cuda), frames, res=8)
    vb, vbid = occ["voxel_bound"], occ["occ_vox_bid"].int().contiguous()
    V = vb.shape[0]
    assert V == cells.numel() and tuple(occ["grid_dims"]) == (9, 9, 9)
    # query points
    R = 6000
    bid = torch.randint(0, frames, (R,), generator=g)
    pos = lo + torch.rand(R, 3, generator=g) * (res * part)
    k = R // 6
    face = torch.randint(0, res + 1, (k, 3), generator=g).float()
    pos[:k] = lo + face * part                                            # cell corners
    pos[k:2 * k, 0] = (lo + torch.randint(0, res + 1, (k, 3), generator=g).float() * part)[:, 0]   # on x faces
    pos[2 * k:3 * k] = vb.cpu()[torch.randint(0, V, (k,), generator=g)][:, [0, 4, 2]]   # stored bounds themselves
    pos[3 * k:3 * k + 50] = lo - 0.3                                       # outside, near
    pos[3 * k + 50:3 * k + 100] = 1e6
    pos[3 * k + 100:3 * k + 120, 1] = float("nan")
    pos[3 * k + 120:3 * k + 130, 2] = float("inf")
    pos[3 * k + 130:3 * k + 140, 0] = -float("inf")
    P = 4000
    pair_vox = torch.randint(0, V, (P,), generator=g).int()
    mid = torch.randint(0, P + 1, (R,), generator=g)                       # P = a ray without pairs
    ray_dir = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=1)
    flat = torch.randint(0, h * w, (R,), generator=g)
    ray_pix = torch.stack((flat % w, flat // w), 1).int()
    rgb = torch.randn(frames, 3, h, w, generator=g)
    feat = torch.randn(frames, 32, h, w, generator=g)
    valid_inp = torch.randn(300, 6, generator=g) * 0.2
    valid_vox = torch.randint(0, V, (300,), generator=g).int()
    pnet = make_pointnet(orc.init_pointnet(5, 1.5), cuda)
    offr = make_module("IEF", orc.init_decoder("IEF", 334, 77, 5.0), 334, cuda)
    args = [t.to(cuda) for t in (ray_dir, ray_pix, bid.int(), flat.int(), pos, mid, pair_vox)]
    rest = [rgb.to(cuda), feat.to(cuda), valid_inp.to(cuda), valid_vox.to(cuda), pnet, offr]
    with torch.no_grad():
        ref_pos, ref_ev = lidf_refine(*args, vb, vbid, *rest, forward_times=1)
        got_pos, got_ev = lidf_refine(*args, vb, vbid, *rest, forward_times=1, grid=occ)
        got2_pos, got2_ev = lidf_refine(*args, vb, vbid, *rest, forward_times=2, grid=occ)
        ref2_pos, ref2_ev = lidf_refine(*args, vb, vbid, *rest, forward_times=2)
    assert torch.equal(ref_ev, got_ev)
    assert torch.equal(ref2_ev, got2_ev)
    assert torch.equal(ref_pos.nan_to_num(1.0, 2.0, 3.0), got_pos.nan_to_num(1.0, 2.0, 3.0))
    assert torch.equal(ref2_pos.nan_to_num(1.0, 2.0, 3.0), got2_pos.nan_to_num(1.0, 2.0, 3.0))
    # and against the definition itself on the CPU: largest voxel of the image containing the point, or the
    # arg-max pair's voxel (0 for the dummy row) when that is larger
    vbc, vbidc = vb.cpu(), vbid.cpu()
    inside = ~((pos[:, None, :] < vbc[None, :, :3]) | (pos[:, None, :] > vbc[None, :, 3:])).any(2)
    inside &= vbidc[None, :] == bid[:, None]
    last = torch.where(inside, torch.arange(V)[None, :], torch.zeros(1, dtype=torch.long)).max(1).values
    argv = torch.cat((pair_vox.long(), torch.zeros(1, dtype=torch.long)))[mid]
    assert torch.equal(got_ev.cpu().long(), torch.maximum(last, argv))
