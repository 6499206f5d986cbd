"""Training path of the query: gradients to feat_grid (through RoIAlign), vox_feat and the decoders'
parameters, against autograd of the CPU oracle's torch restatement of get_embedding + get_pred."""
import pytest
import torch

from util import TOL, make_module, orc, to_dev

pytestmark = pytest.mark.gpu


def _loss(out, w):
    return ((out["pred_prob_end"][:, 0] * w["prob"]).sum() + (out["pred_offset"][:, 0] * w["off"]).sum()
            + (out["pred_pos"] * w["pos"]).sum())


@pytest.mark.parametrize("factorised", [True, False])
@pytest.mark.parametrize("ragged,pos_rel", [(False, False), (True, True)])
def test_query_gradients(cuda, ragged, pos_rel, factorised):
    from implicit_depth_amd.query import lidf_query, lidf_query_train
    scene = orc.synthetic_scene(2, 12, 16, 8, seed=71, ragged=ragged)
    R, P, D = scene["R"], scene["P"], scene["D"]
    gen = torch.Generator().manual_seed(72)
    w = {"prob": torch.randn(P, generator=gen), "off": torch.randn(P, generator=gen),
         "pos": torch.randn(R, 3, generator=gen)}
    kw = dict(offset_range=(-0.2, 0.2), part_size=0.25)
    # ---- oracle autograd (torch CPU)
    pp = {k: v.clone().requires_grad_(True) for k, v in scene["prob_p"].items()}
    po = {k: v.clone().requires_grad_(True) for k, v in scene["off_p"].items()}
    fg = scene["feat_grid"].clone().requires_grad_(True)
    vf = scene["vox_feat"].clone().requires_grad_(True)
    ref = orc.query(scene["ray_dir"], scene["ray_pix"], scene["ray_bid"], scene["pair_ray"].long(),
                    scene["pair_vox"].long(), scene["pair_t"], scene["pair_off"], fg, vf, pp, po,
                    fast_roi=True, vox_center=scene["vox_center"], pos_rel=pos_rel, **kw)
    _loss(ref, w).backward()
    # ---- product
    s = to_dev(scene, cuda)
    prob = make_module("IMNET", scene["prob_p"], D, cuda).train()
    off = make_module("IEF", scene["off_p"], D, cuda).train()
    fgd = s["feat_grid"].clone().requires_grad_(True)
    vfd = s["vox_feat"].clone().requires_grad_(True)
    args = (s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"], s["pair_vox"], s["pair_t"])
    out = lidf_query_train(*args, fgd, vfd, prob, off, vox_center=s["vox_center"], pos_rel=pos_rel,
                           factorised=factorised, **kw)
    _loss(out, {k: v.to(cuda) for k, v in w.items()}).backward()
    # forward values: the oracle, and the inference kernel
    for k in ("pred_offset", "pred_prob_end", "pair_pred_pos", "pred_pos"):
        assert (out[k].detach().cpu() - ref[k].detach()).abs().max().item() <= TOL, k
    with torch.no_grad():
        inf = lidf_query(*args, s["feat_grid"], s["vox_feat"], prob, off, vox_center=s["vox_center"],
                         pos_rel=pos_rel, **kw)
    assert (inf["pred_prob_end"] - out["pred_prob_end"]).abs().max().item() <= 1e-5
    assert (inf["max_pair_id"] == out["max_pair_id"]).all()
    # gradients
    def close(a, b, what):
        scale = max(1e-2, b.abs().max().item())
        assert (a - b).abs().max().item() <= 5e-4 * scale, (what, (a - b).abs().max().item(), scale)
    close(fgd.grad.cpu(), fg.grad, "feat_grid")
    close(vfd.grad.cpu(), vf.grad, "vox_feat")
    for k, v in pp.items():
        close(dict(prob.named_parameters())[k].grad.cpu(), v.grad, "prob." + k)
    for k, v in po.items():
        close(dict(off.named_parameters())[k].grad.cpu(), v.grad, "off." + k)


def test_ray_features_backward_border(cuda):
    """RoIAlign backward on its own, boxes clamped at the image border (fractional samples)."""
    from implicit_depth_amd.query import _RayFeaturesFn
    scene = orc.synthetic_scene(2, 12, 16, 1, seed=73)
    s = to_dev(scene, cuda)
    gen = torch.Generator().manual_seed(74)
    for bbox in (8, 7, 4):
        wgt = torch.randn(scene["R"], 128, generator=gen)
        fg = scene["feat_grid"].clone().requires_grad_(True)
        boxes = orc.roi_boxes(scene["ray_pix"].long(), scene["ray_bid"].long(), 12, 16, bbox)
        (orc.roi_align_fast(fg, boxes).reshape(scene["R"], -1) * wgt).sum().backward()
        fgd = s["feat_grid"].clone().requires_grad_(True)
        got = _RayFeaturesFn.apply(fgd, s["ray_dir"], s["ray_pix"], s["ray_bid"], bbox, 4)
        (got[:, :128] * wgt.to(cuda)).sum().backward()
        assert (fgd.grad.cpu() - fg.grad).abs().max().item() <= 1e-5 * max(1.0, fg.grad.abs().max().item()), bbox


@pytest.mark.parametrize("factorised", [True, False])
def test_query_train_no_pairs(cuda, factorised):
    """No ray meets a voxel: empty outputs, zero gradients everywhere, nothing launched on P = 0."""
    from implicit_depth_amd.query import lidf_query_train
    scene = orc.synthetic_scene(1, 8, 8, 4, seed=75)
    s = to_dev(scene, cuda)
    D = scene["D"]
    prob = make_module("IMNET", scene["prob_p"], D, cuda).train()
    off = make_module("IEF", scene["off_p"], D, cuda).train()
    R = scene["R"]
    zero_off = torch.zeros(R + 1, dtype=torch.int32, device=cuda)
    e_i = torch.zeros(0, dtype=torch.int32, device=cuda)
    fg = s["feat_grid"].clone().requires_grad_(True)
    vf = s["vox_feat"].clone().requires_grad_(True)
    out = lidf_query_train(s["ray_dir"], s["ray_pix"], s["ray_bid"], zero_off, e_i, e_i,
                           torch.zeros(0, 2, device=cuda), fg, vf, prob, off, factorised=factorised)
    assert out["pred_prob_end"].shape == (0, 1) and out["pred_pos"].shape == (R, 3)
    assert (out["max_pair_id"] == 0).all() and (out["pred_pos"] == 0).all()
    (out["pred_pos"].sum() + out["pred_offset"].sum() + out["pred_prob_end"].sum()).backward()
    assert (fg.grad == 0).all() and (vf.grad == 0).all()
    assert all(p.grad is not None and (p.grad == 0).all() for p in list(prob.parameters()) + list(off.parameters()))


def test_query_train_label_selection_and_detached_softmax(cuda):
    """max_pair_id override (the reference's ground-truth selection, pipeline.py:444-446), and no
    gradient through the per-ray softmax (the reference detaches the logits, :442)."""
    from implicit_depth_amd.query import lidf_query_train
    scene = orc.synthetic_scene(1, 8, 12, 6, seed=81)
    s = to_dev(scene, cuda)
    D, R, P = scene["D"], scene["R"], scene["P"]
    prob = make_module("IMNET", scene["prob_p"], D, cuda).train()
    off = make_module("IEF", scene["off_p"], D, cuda).train()
    args = (s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"], s["pair_vox"], s["pair_t"])
    g = torch.Generator().manual_seed(82)
    sel = (torch.arange(R) * 6 + torch.randint(0, 6, (R,), generator=g))
    sel[::5] = P                                            # some rays select the dummy row
    out = lidf_query_train(*args, s["feat_grid"], s["vox_feat"], prob, off, max_pair_id=sel.to(cuda))
    assert (out["max_pair_id"].cpu() == sel).all()
    dummy = torch.cat((out["pair_pred_pos"].detach().cpu(), torch.zeros(1, 3)), 0)
    assert (out["pred_pos"].detach().cpu() == dummy[sel]).all()
    assert not out["pred_prob_end_softmax"].requires_grad
    out["pred_pos"].sum().backward()
    assert all(p.grad is None or (p.grad == 0).all() for p in prob.parameters())   # nothing via logits
    assert off.linear_4.weight.grad.abs().sum().item() > 0


def test_ray_features_backward_duplicate_rays(cuda):
    """Two rays naming the same pixel: their gradients add (the parked store overwrote)."""
    from implicit_depth_amd.query import _RayFeaturesFn
    scene = orc.synthetic_scene(1, 16, 20, 1, seed=83)
    s = to_dev(scene, cuda)
    idx = torch.tensor([105, 105, 106, 190, 105, 0, 0])     # interior duplicates and a border duplicate
    pix, bid, d = s["ray_pix"][idx].contiguous(), s["ray_bid"][idx].contiguous(), s["ray_dir"][idx].contiguous()
    wgt = torch.randn(len(idx), 128, generator=torch.Generator().manual_seed(84))
    fg = scene["feat_grid"].clone().requires_grad_(True)
    boxes = orc.roi_boxes(scene["ray_pix"][idx].long(), scene["ray_bid"][idx].long(), 16, 20, 8)
    (orc.roi_align_fast(fg, boxes).reshape(len(idx), -1) * wgt).sum().backward()
    fgd = s["feat_grid"].clone().requires_grad_(True)
    got = _RayFeaturesFn.apply(fgd, d, pix, bid, 8, 4)
    (got[:, :128] * wgt.to(cuda)).sum().backward()
    assert (fgd.grad.cpu() - fg.grad).abs().max().item() <= 1e-5 * max(1.0, fg.grad.abs().max().item())


def test_frozen_parameters_and_inplace_update_check(cuda):
    """Frozen parameters get no gradient; an in-place parameter update between forward and backward
    is caught by torch's version counters (the backward uses what the forward saved)."""
    d = 385
    p = orc.randomize_biases(orc.init_decoder("IEF", d, 5, 5.0), 6)
    m = make_module("IEF", p, d, cuda).train()
    m.linear_2.weight.requires_grad_(False)
    x = torch.randn(40, d, generator=torch.Generator().manual_seed(7)).to(cuda)
    y = m(x)
    y.sum().backward()
    assert m.linear_2.weight.grad is None and m.linear_1.weight.grad is not None
    y = m(x)
    with torch.no_grad():
        m.linear_3.weight.mul_(1.5)
    with pytest.raises(RuntimeError):
        y.sum().backward()


def test_query_gradients_larger_scene_and_run_to_run_identical(cuda):
    """49k pairs over several sort blocks of the per-voxel reduction (stable counting sort + chunk
    sums in a fixed order) and several row slices of the weight-gradient kernels: gradients against
    oracle autograd, and bit-identical between two runs for vox_feat and every parameter."""
    from implicit_depth_amd.query import lidf_query_train
    scene = orc.synthetic_scene(1, 48, 64, 16, seed=91, ragged=True)
    R, P, D = scene["R"], scene["P"], scene["D"]
    gen = torch.Generator().manual_seed(92)
    w = {"prob": torch.randn(P, generator=gen), "off": torch.randn(P, generator=gen),
         "pos": torch.randn(R, 3, generator=gen)}
    kw = dict(offset_range=(0.0, 1.0), part_size=0.25)
    pp = {k: v.clone().requires_grad_(True) for k, v in scene["prob_p"].items()}
    po = {k: v.clone().requires_grad_(True) for k, v in scene["off_p"].items()}
    vf = scene["vox_feat"].clone().requires_grad_(True)
    ref = orc.query(scene["ray_dir"], scene["ray_pix"], scene["ray_bid"], scene["pair_ray"].long(),
                    scene["pair_vox"].long(), scene["pair_t"], scene["pair_off"], scene["feat_grid"], vf,
                    pp, po, fast_roi=True, **kw)
    _loss(ref, w).backward()
    s = to_dev(scene, cuda)
    wd = {k: v.to(cuda) for k, v in w.items()}
    args = (s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"], s["pair_vox"], s["pair_t"])
    runs = []
    for _ in range(2):
        prob = make_module("IMNET", scene["prob_p"], D, cuda).train()
        off = make_module("IEF", scene["off_p"], D, cuda).train()
        vfd = s["vox_feat"].clone().requires_grad_(True)
        out = lidf_query_train(*args, s["feat_grid"], vfd, prob, off, **kw)
        _loss(out, wd).backward()
        g = {"vox_feat": vfd.grad.clone()}
        g.update({"prob." + k: v.grad.clone() for k, v in prob.named_parameters()})
        g.update({"off." + k: v.grad.clone() for k, v in off.named_parameters()})
        runs.append(g)
    differ = [(k, (runs[0][k] - runs[1][k]).abs().max().item()) for k in runs[0]
              if not torch.equal(runs[0][k], runs[1][k])]
    assert not differ, differ

    def close(a, b, what):
        scale = max(1e-2, b.abs().max().item())
        assert (a - b).abs().max().item() <= 5e-4 * scale, (what, (a - b).abs().max().item(), scale)
    close(runs[0]["vox_feat"].cpu(), vf.grad, "vox_feat")
    for k, v in pp.items():
        close(runs[0]["prob." + k].cpu(), v.grad, "prob." + k)
    for k, v in po.items():
        close(runs[0]["off." + k].cpu(), v.grad, "off." + k)


@pytest.mark.parametrize("h,w,bbox", [(17, 23, 8), (18, 35, 8), (40, 50, 4), (33, 48, 16), (9, 9, 8), (8, 8, 8)])
def test_ray_features_backward_image_shapes(cuda, h, w, bbox):
    """Every pixel a ray, image sizes that are no multiple of the ring kernel's 16-pixel tiles, a
    last tile narrower than the ring of clamped boxes ((18, 35): falls back to the atomic path),
    boxes as wide as the image ((8, 8): every box clamped on both sides)."""
    from implicit_depth_amd.query import _RayFeaturesFn
    scene = orc.synthetic_scene(2, h, w, 1, seed=h * 100 + w)
    s = to_dev(scene, cuda)
    R = scene["R"]
    wgt = torch.randn(R, 128, generator=torch.Generator().manual_seed(bbox))
    fg = scene["feat_grid"].clone().requires_grad_(True)
    boxes = orc.roi_boxes(scene["ray_pix"].long(), scene["ray_bid"].long(), h, w, bbox)
    (orc.roi_align_fast(fg, boxes).reshape(R, -1) * wgt).sum().backward()
    fgd = s["feat_grid"].clone().requires_grad_(True)
    got = _RayFeaturesFn.apply(fgd, s["ray_dir"], s["ray_pix"], s["ray_bid"], bbox, 4)
    (got[:, :128] * wgt.to(cuda)).sum().backward()
    assert (fgd.grad.cpu() - fg.grad).abs().max().item() <= 2e-5 * max(1.0, fg.grad.abs().max().item())


@pytest.mark.parametrize("L,Lv,n_vox", [(4, 2, 0), (0, 0, 0), (8, 4, 17000)])
def test_query_gradients_other_encoding_widths(cuda, L, Lv, n_vox):
    """multires 4 / multires_views 2 (D = 325), no positional encoding (pos_encode: False, D = 265,
    models/pipeline.py:44-47) and an IMNET offset decoder with sigmoid outputs (offdec_type: IMNET,
    use_sigmoid, :69-71): other layer-1 widths through the chained training forward, the
    position-embedding rows and the per-ray parts. n_vox = 17,000: a voxel table beyond the sorted
    per-voxel reduction's 16,384 rows (the run-walking atomic form takes over)."""
    from implicit_depth_amd.query import lidf_query_train
    D = 256 + 2 * (3 + 6 * L) + 3 + 6 * Lv
    scene = orc.synthetic_scene(2, 10, 14, 6, seed=101, ragged=True)
    R, P = scene["R"], scene["P"]
    if n_vox:
        g0 = torch.Generator().manual_seed(107)
        scene["vox_feat"] = torch.relu(torch.randn(n_vox, 128, generator=g0))
        scene["pair_vox"] = torch.randint(0, n_vox, (P,), generator=g0, dtype=torch.int32)
    prob_p = orc.randomize_biases(orc.init_decoder("IMNET", D, 102, 5.0), 103)
    off_p = orc.randomize_biases(orc.init_decoder("IMNET", D, 104, 5.0), 105)
    gen = torch.Generator().manual_seed(106)
    w = {"prob": torch.randn(P, generator=gen), "off": torch.randn(P, generator=gen),
         "pos": torch.randn(R, 3, generator=gen)}
    kw = dict(multires=L, multires_views=Lv, offset_range=(0.0, 1.0), part_size=0.25)
    pp = {k: v.clone().requires_grad_(True) for k, v in prob_p.items()}
    po = {k: v.clone().requires_grad_(True) for k, v in off_p.items()}
    fg = scene["feat_grid"].clone().requires_grad_(True)
    vf = scene["vox_feat"].clone().requires_grad_(True)
    ref = orc.query(scene["ray_dir"], scene["ray_pix"], scene["ray_bid"], scene["pair_ray"].long(),
                    scene["pair_vox"].long(), scene["pair_t"], scene["pair_off"], fg, vf, pp, po,
                    off_kind="IMNET", use_sigmoid=True, fast_roi=True, **kw)
    _loss(ref, w).backward()
    s = to_dev(scene, cuda)
    prob = make_module("IMNET", prob_p, D, cuda, use_sigmoid=True).train()
    off = make_module("IMNET", off_p, D, cuda, use_sigmoid=True).train()
    fgd = s["feat_grid"].clone().requires_grad_(True)
    vfd = s["vox_feat"].clone().requires_grad_(True)
    out = lidf_query_train(s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"],
                           s["pair_vox"], s["pair_t"], fgd, vfd, prob, off, **kw)
    _loss(out, {k: v.to(cuda) for k, v in w.items()}).backward()
    for k in ("pred_offset", "pred_prob_end", "pair_pred_pos", "pred_pos"):
        assert (out[k].detach().cpu() - ref[k].detach()).abs().max().item() <= TOL, k

    def close(a, b, what):
        scale = max(1e-2, b.abs().max().item())
        assert (a - b).abs().max().item() <= 5e-4 * scale, (what, (a - b).abs().max().item(), scale)
    close(fgd.grad.cpu(), fg.grad, "feat_grid")
    close(vfd.grad.cpu(), vf.grad, "vox_feat")
    for k, v in pp.items():
        close(dict(prob.named_parameters())[k].grad.cpu(), v.grad, "prob." + k)
    for k, v in po.items():
        close(dict(off.named_parameters())[k].grad.cpu(), v.grad, "off." + k)


def test_query_gradients_three_passes(cuda):
    """IEF n_iter = 3 in the factorised query: the last pass writes dZ1 into the running sum, the
    middle one adds its own in the offset-encoding sweep, the first one is added by the chained
    input-gradient launch and gets its offset-encoding share from column sums."""
    from implicit_depth_amd.query import lidf_query_train
    scene = orc.synthetic_scene(2, 10, 14, 6, seed=111, ragged=True)
    R, P, D = scene["R"], scene["P"], scene["D"]
    off_p = orc.randomize_biases(orc.init_decoder("IEF", D, 112, 5.0), 113)
    gen = torch.Generator().manual_seed(114)
    w = {"prob": torch.randn(P, generator=gen), "off": torch.randn(P, generator=gen),
         "pos": torch.randn(R, 3, generator=gen)}
    pp = {k: v.clone().requires_grad_(True) for k, v in scene["prob_p"].items()}
    po = {k: v.clone().requires_grad_(True) for k, v in off_p.items()}
    vf = scene["vox_feat"].clone().requires_grad_(True)
    ref = orc.query(scene["ray_dir"], scene["ray_pix"], scene["ray_bid"], scene["pair_ray"].long(),
                    scene["pair_vox"].long(), scene["pair_t"], scene["pair_off"], scene["feat_grid"], vf,
                    pp, po, n_iter=3, fast_roi=True)
    _loss(ref, w).backward()
    s = to_dev(scene, cuda)
    prob = make_module("IMNET", scene["prob_p"], D, cuda).train()
    off = make_module("IEF", off_p, D, cuda, n_iter=3).train()
    vfd = s["vox_feat"].clone().requires_grad_(True)
    out = lidf_query_train(s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"],
                           s["pair_vox"], s["pair_t"], s["feat_grid"], vfd, prob, off)
    _loss(out, {k: v.to(cuda) for k, v in w.items()}).backward()
    for k in ("pred_offset", "pred_prob_end", "pred_pos"):
        assert (out[k].detach().cpu() - ref[k].detach()).abs().max().item() <= TOL, k

    def close(a, b, what):
        scale = max(1e-2, b.abs().max().item())
        assert (a - b).abs().max().item() <= 5e-4 * scale, (what, (a - b).abs().max().item(), scale)
    close(vfd.grad.cpu(), vf.grad, "vox_feat")
    for k, v in po.items():
        close(dict(off.named_parameters())[k].grad.cpu(), v.grad, "off." + k)


def _ref_loss(out, w):
    """The shape of the reference's training loss (pipeline.py:468-490): every term on offset_dec goes through
    pred_pos (the selected pair of every ray), the logits receive a gradient at every pair."""
    return (out["pred_prob_end"][:, 0] * w["prob"]).sum() + (out["pred_pos"] * w["pos"]).sum()


@pytest.mark.parametrize("labels", [False, True])
@pytest.mark.parametrize("ragged,pos_rel,shape", [(False, False, (2, 12, 16, 8)), (True, True, (1, 48, 64, 16))])
def test_query_gradients_offset_decoder_from_selected_rows(cuda, ragged, pos_rel, shape, labels):
    """A loss that reaches offset_dec through pred_pos alone: its backward runs over the selected pair of every ray
    (lidf_query_decoder_backward_rows_f32; rays without a pair, label selections of the dummy row) — gradients
    against oracle autograd, against the dense backward of the same loss (forced by a zero-weight term on
    pred_offset), and bit-identical run to run."""
    from implicit_depth_amd.query import lidf_query_train
    scene = orc.synthetic_scene(*shape, seed=131, ragged=ragged)
    R, P, D = scene["R"], scene["P"], scene["D"]
    gen = torch.Generator().manual_seed(132)
    w = {"prob": torch.randn(P, generator=gen), "pos": torch.randn(R, 3, generator=gen)}
    kw = dict(offset_range=(-0.2, 0.2), part_size=0.25)
    sel = None
    if labels:   # (pipeline.py:444-446: ground-truth selection; every 7th ray selects the dummy row)
        cnt = (scene["pair_off"][1:] - scene["pair_off"][:-1]).long()
        pick = (torch.rand(R, generator=gen) * cnt.clamp(min=1)).long().clamp(max=(cnt - 1).clamp(min=0))
        sel = torch.where(cnt > 0, scene["pair_off"][:-1].long() + pick, torch.full((R,), P))
        sel[::7] = P
    pp = {k: v.clone().requires_grad_(True) for k, v in scene["prob_p"].items()}
    po = {k: v.clone().requires_grad_(True) for k, v in scene["off_p"].items()}
    fg = scene["feat_grid"].clone().requires_grad_(True)
    vf = scene["vox_feat"].clone().requires_grad_(True)
    ref = orc.query(scene["ray_dir"], scene["ray_pix"], scene["ray_bid"], scene["pair_ray"].long(),
                    scene["pair_vox"].long(), scene["pair_t"], scene["pair_off"], fg, vf, pp, po,
                    fast_roi=True, vox_center=scene["vox_center"], pos_rel=pos_rel, max_pair_id=sel, **kw)
    _ref_loss(ref, w).backward()
    s = to_dev(scene, cuda)
    wd = {k: v.to(cuda) for k, v in w.items()}
    args = (s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"], s["pair_vox"], s["pair_t"])
    runs = {}
    for name in ("rows", "rows again", "dense"):
        prob = make_module("IMNET", scene["prob_p"], D, cuda).train()
        off = make_module("IEF", scene["off_p"], D, cuda).train()
        fgd = s["feat_grid"].clone().requires_grad_(True)
        vfd = s["vox_feat"].clone().requires_grad_(True)
        out = lidf_query_train(*args, fgd, vfd, prob, off, vox_center=s["vox_center"], pos_rel=pos_rel,
                               max_pair_id=None if sel is None else sel.to(cuda), **kw)
        loss = _ref_loss(out, wd)
        if name == "dense":
            loss = loss + 0.0 * out["pred_offset"].sum()
        loss.backward()
        g = {"feat_grid": fgd.grad.clone(), "vox_feat": vfd.grad.clone()}
        g.update({"prob." + k: v.grad.clone() for k, v in prob.named_parameters()})
        g.update({"off." + k: v.grad.clone() for k, v in off.named_parameters()})
        runs[name] = g
    assert (out["max_pair_id"].cpu() == ref["max_pair_id"]).all()
    differ = [k for k in runs["rows"] if k != "feat_grid" and not torch.equal(runs["rows"][k], runs["rows again"][k])]
    assert not differ, differ

    def close(a, b, what, tol):
        scale = max(1e-2, b.abs().max().item())
        assert (a - b).abs().max().item() <= tol * scale, (what, (a - b).abs().max().item(), scale)
    for k in runs["rows"]:
        close(runs["rows"][k], runs["dense"][k], "rows vs dense " + k, 2e-5)
    # (the larger scene has pre-activations within rounding of the leaky-relu kink — 11 M of them, min |z| 1e-7 —
    # and a flipped slope moves the input gradient of that one ray by a few 1e-3 of the largest entry, in the dense
    # backward of prob_dec as much as here: 1e-2 there for everything prob_dec's gradient enters, 5e-4 for offset_dec's)
    tol_in = 5e-4 if P < 4096 else 1e-2
    close(runs["rows"]["feat_grid"].cpu(), fg.grad, "feat_grid", tol_in)
    close(runs["rows"]["vox_feat"].cpu(), vf.grad, "vox_feat", tol_in)
    for k, v in pp.items():
        close(runs["rows"]["prob." + k].cpu(), v.grad, "prob." + k, tol_in)
    for k, v in po.items():
        close(runs["rows"]["off." + k].cpu(), v.grad, "off." + k, 5e-4)
    assert runs["rows"]["off.linear_4.weight"].abs().sum().item() > 0


def test_query_train_unused_outputs(cuda):
    """Losses that use one output only: no gradient reaches the other decoder (zeros, not None)."""
    from implicit_depth_amd.query import lidf_query_train
    scene = orc.synthetic_scene(1, 8, 12, 6, seed=141)
    s = to_dev(scene, cuda)
    D = scene["D"]
    args = (s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"], s["pair_vox"], s["pair_t"])
    for key, live, dead in (("pred_prob_end", "prob", "off"), ("pred_pos", "off", "prob"),
                            ("pair_pred_pos", "off", "prob"), ("pred_offset", "off", "prob")):
        mods = {"prob": make_module("IMNET", scene["prob_p"], D, cuda).train(),
                "off": make_module("IEF", scene["off_p"], D, cuda).train()}
        out = lidf_query_train(*args, s["feat_grid"], s["vox_feat"], mods["prob"], mods["off"])
        out[key].sum().backward()
        assert all(p.grad is not None and (p.grad == 0).all() for p in mods[dead].parameters()), key
        assert mods[live].linear_2.weight.grad.abs().sum().item() > 0, key


@pytest.mark.parametrize("labels", [False, True])
@pytest.mark.parametrize("ragged,pos_rel,shape", [(False, False, (2, 12, 16, 8)), (True, True, (1, 48, 64, 16))])
def test_query_train_offsets_selected(cuda, ragged, pos_rel, shape, labels):
    """offsets="selected" in training (lidf_query_forward_train_selected_f32): offset_dec runs on the selected pair of
    every ray only, forward and backward. The logits, softmax, selection, pred_pos and the selected pairs' slots of
    pred_offset / pair_pred_pos equal offsets="all" (the other slots are zero), and so do the gradients of a loss
    on pred_pos and the logits; a loss on the selected slots of pair_pred_pos / pred_offset is differentiated too."""
    from implicit_depth_amd.query import lidf_query_train
    scene = orc.synthetic_scene(*shape, seed=151, ragged=ragged)
    R, P, D = scene["R"], scene["P"], scene["D"]
    gen = torch.Generator().manual_seed(152)
    w = {"prob": torch.randn(P, generator=gen).to(cuda), "pos": torch.randn(R, 3, generator=gen).to(cuda)}
    wpp, wpo = torch.randn(P, 3, generator=gen).to(cuda), torch.randn(P, generator=gen).to(cuda)
    kw = dict(offset_range=(-0.2, 0.2), part_size=0.25)
    sel = None
    if labels:
        cnt = (scene["pair_off"][1:] - scene["pair_off"][:-1]).long()
        pick = (torch.rand(R, generator=gen) * cnt.clamp(min=1)).long().clamp(max=(cnt - 1).clamp(min=0))
        sel = torch.where(cnt > 0, scene["pair_off"][:-1].long() + pick, torch.full((R,), P))
        sel[::7] = P
        sel = sel.to(cuda)
    s = to_dev(scene, cuda)
    args = (s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"], s["pair_vox"], s["pair_t"])
    for extra in (False, True):   # extra: the loss also reads pair_pred_pos / pred_offset at the selected pairs
        runs = {}
        for mode in ("all", "selected"):
            prob = make_module("IMNET", scene["prob_p"], D, cuda).train()
            off = make_module("IEF", scene["off_p"], D, cuda).train()
            fgd = s["feat_grid"].clone().requires_grad_(True)
            vfd = s["vox_feat"].clone().requires_grad_(True)
            out = lidf_query_train(*args, fgd, vfd, prob, off, vox_center=s["vox_center"], pos_rel=pos_rel,
                                   max_pair_id=sel, offsets=mode, **kw)
            loss = _ref_loss(out, w)
            if extra:
                used = out["max_pair_id"] if sel is None else sel
                m = torch.zeros(P + 1, device=cuda)
                m[used] = 1.0
                m = m[:P]   # 1 at the selected pairs (the dummy row dropped)
                loss = loss + (out["pair_pred_pos"] * wpp * m[:, None]).sum() + (out["pred_offset"][:, 0] * wpo * m).sum()
            loss.backward()
            g = {"feat_grid": fgd.grad.clone(), "vox_feat": vfd.grad.clone()}
            g.update({"prob." + k: v.grad.clone() for k, v in prob.named_parameters()})
            g.update({"off." + k: v.grad.clone() for k, v in off.named_parameters()})
            runs[mode] = (out, g, m if extra else None)
        oa, os_ = runs["all"][0], runs["selected"][0]
        for k in ("pred_prob_end", "pred_prob_end_softmax", "max_pair_id"):
            assert torch.equal(oa[k], os_[k]), k
        assert (oa["pred_pos"] - os_["pred_pos"]).abs().max().item() <= 1e-6
        used = oa["max_pair_id"] if sel is None else sel
        live = used[used < P]
        assert (oa["pred_offset"][live] - os_["pred_offset"][live]).abs().max().item() <= 1e-6
        assert (oa["pair_pred_pos"][live] - os_["pair_pred_pos"][live]).abs().max().item() <= 1e-6
        rest = torch.ones(P, dtype=torch.bool, device=cuda)
        rest[live] = False
        assert (os_["pred_offset"][rest] == 0).all() and (os_["pair_pred_pos"][rest] == 0).all()
        for k in runs["all"][1]:
            a, b = runs["selected"][1][k], runs["all"][1][k]
            scale = max(1e-2, b.abs().max().item())
            assert (a - b).abs().max().item() <= 5e-5 * scale, (extra, k, (a - b).abs().max().item(), scale)
        assert runs["selected"][1]["off.linear_4.weight"].abs().sum().item() > 0


def test_query_train_offsets_selected_no_pairs(cuda):
    from implicit_depth_amd.query import lidf_query_train
    scene = orc.synthetic_scene(1, 8, 12, 6, seed=161)
    s = to_dev(scene, cuda)
    D, R = scene["D"], scene["R"]
    prob = make_module("IMNET", scene["prob_p"], D, cuda).train()
    off = make_module("IEF", scene["off_p"], D, cuda).train()
    z = torch.zeros
    out = lidf_query_train(s["ray_dir"], s["ray_pix"], s["ray_bid"], z(R + 1, dtype=torch.int32, device=cuda),
                           z(0, dtype=torch.int32, device=cuda), z(0, dtype=torch.int32, device=cuda),
                           z((0, 2), device=cuda), s["feat_grid"], s["vox_feat"], prob, off, offsets="selected")
    assert out["pred_pos"].shape == (R, 3) and (out["pred_pos"] == 0).all()
    assert (out["max_pair_id"] == 0).all()
    out["pred_pos"].sum().backward()
    assert all(p.grad is not None and (p.grad == 0).all() for p in list(prob.parameters()) + list(off.parameters()))


@pytest.mark.parametrize("loss_kind", ["pred_pos_sum", "dense_sums"])
def test_query_train_expanded_output_gradients(cuda, loss_kind):
    """Losses whose output gradients reach the node as expanded stride-0 tensors (pred_pos.sum() hands a
    [R,3] view of ONE element to the backward): the library reads raw pointers, so the node must
    densify them — gradients against the oracle's autograd for the rows-only route (pred_pos alone) and the
    dense route (pair_pred_pos / pred_offset / logits touched)."""
    from implicit_depth_amd.query import lidf_query_train
    scene = orc.synthetic_scene(2, 10, 14, 6, seed=171, ragged=True)
    D = scene["D"]
    kw = dict(offset_range=(0.0, 1.0), part_size=0.25)

    def loss(o):
        if loss_kind == "pred_pos_sum":
            return o["pred_pos"].sum()
        return o["pred_pos"].sum() + o["pair_pred_pos"].sum() + o["pred_offset"].sum() + o["pred_prob_end"].sum()
    pp = {k: v.clone().requires_grad_(True) for k, v in scene["prob_p"].items()}
    po = {k: v.clone().requires_grad_(True) for k, v in scene["off_p"].items()}
    fg = scene["feat_grid"].clone().requires_grad_(True)
    vf = scene["vox_feat"].clone().requires_grad_(True)
    ref = orc.query(scene["ray_dir"], scene["ray_pix"], scene["ray_bid"], scene["pair_ray"].long(),
                    scene["pair_vox"].long(), scene["pair_t"], scene["pair_off"], fg, vf, pp, po, fast_roi=True, **kw)
    loss(ref).backward()
    s = to_dev(scene, cuda)
    prob = make_module("IMNET", scene["prob_p"], D, cuda).train()
    off = make_module("IEF", scene["off_p"], D, cuda).train()
    fgd = s["feat_grid"].clone().requires_grad_(True)
    vfd = s["vox_feat"].clone().requires_grad_(True)
    out = lidf_query_train(s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"], s["pair_vox"],
                           s["pair_t"], fgd, vfd, prob, off, **kw)
    loss(out).backward()

    def close(a, b, what):
        scale = max(1e-2, b.abs().max().item())
        assert (a - b).abs().max().item() <= 5e-4 * scale, (what, (a - b).abs().max().item(), scale)
    close(fgd.grad.cpu(), fg.grad, "feat_grid")
    close(vfd.grad.cpu(), vf.grad, "vox_feat")
    for k, v in po.items():
        close(dict(off.named_parameters())[k].grad.cpu(), v.grad, "off." + k)
    for k, v in pp.items():
        g = dict(prob.named_parameters())[k].grad
        if v.grad is None:
            assert g is None or (g == 0).all(), k
        else:
            close(g.cpu(), v.grad, "prob." + k)


def test_query_train_selection_output_is_a_copy(cuda):
    """A caller-supplied max_pair_id comes back as a tensor of the node's own (never the caller's tensor aliased
    as an output), equal in value."""
    from implicit_depth_amd.query import lidf_query_train
    scene = orc.synthetic_scene(1, 8, 12, 6, seed=181)
    s = to_dev(scene, cuda)
    D, R = scene["D"], scene["R"]
    prob = make_module("IMNET", scene["prob_p"], D, cuda).train()
    off = make_module("IEF", scene["off_p"], D, cuda).train()
    sel = (torch.arange(R) * 6).to(cuda)
    for offsets in ("all", "selected"):
        out = lidf_query_train(s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"], s["pair_vox"],
                               s["pair_t"], s["feat_grid"], s["vox_feat"], prob, off, max_pair_id=sel, offsets=offsets)
        assert out["max_pair_id"].data_ptr() != sel.data_ptr() and torch.equal(out["max_pair_id"], sel)
