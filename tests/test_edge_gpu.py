"""GPU: edge cases of the fused query the reference handles by early exits (models/pipeline.py:
181-183, 287-289, 686-687) and the alternative configurations of SURVEY.md §8a (iv)-(vi), plus
size-independent properties at the full BASELINE size."""
import pytest
import torch

from util import TOL, make_module, oracle_query, orc, run_query, to_dev

pytestmark = pytest.mark.gpu


def test_no_pairs_and_no_rays(cuda):
    from implicit_depth_amd.query import lidf_query
    scene = orc.synthetic_scene(1, 8, 8, 4, seed=5)
    s = to_dev(scene, cuda)
    D = scene["D"]
    prob, off = make_module("IMNET", scene["prob_p"], D, cuda), make_module("IEF", scene["off_p"], D, cuda)
    R = scene["R"]
    zero_off = torch.zeros(R + 1, dtype=torch.int32, device=cuda)
    e_i = torch.zeros(0, dtype=torch.int32, device=cuda)
    depth = torch.full((1, 8, 8), 7.0, device=cuda)
    with torch.no_grad():
        out = lidf_query(s["ray_dir"], s["ray_pix"], s["ray_bid"], zero_off, e_i, e_i,
                         torch.zeros(0, 2, device=cuda), s["feat_grid"], s["vox_feat"], prob, off,
                         ray_flat=s["ray_flat"], depth=depth)
    assert out["pair_pred_pos"].shape == (0, 3)
    assert (out["max_pair_id"] == 0).all()            # P == 0 -> the dummy row index
    assert (out["pred_pos"] == 0).all() and (depth == 0).all()
    with torch.no_grad():
        out = lidf_query(s["ray_dir"][:0], s["ray_pix"][:0], s["ray_bid"][:0], zero_off[:1], e_i, e_i,
                         torch.zeros(0, 2, device=cuda), s["feat_grid"], s["vox_feat"], prob, off)
    assert out["pred_pos"].shape == (0, 3)


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_relative_positions_and_offset_range(cuda, precision):
    scene = orc.synthetic_scene(1, 12, 16, 8, seed=6, ragged=True)
    kw = dict(offset_range=(-0.2, 0.2), part_size=0.25)
    ref = oracle_query(scene, vox_center=scene["vox_center"], pos_rel=True, **kw)
    got = run_query(scene, cuda, vox_center=scene["vox_center"].to(cuda), pos_rel=True,
                    precision=precision, **kw)
    for k in ("pred_offset", "pred_prob_end", "pair_pred_pos", "pred_pos"):
        assert (got[k].cpu() - ref[k]).abs().max().item() <= TOL, k


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_identity_embedder_and_imnet_offset(cuda, precision):
    """pos_encode False (D = 265) and offdec_type IMNET."""
    from implicit_depth_amd.query import lidf_query
    scene = orc.synthetic_scene(1, 12, 16, 8, seed=7, multires=0, multires_views=0)
    D = 256 + 9
    pp = orc.randomize_biases(orc.init_decoder("IMNET", D, 31, 5.0), 1)
    po = orc.randomize_biases(orc.init_decoder("IMNET", D, 32, 5.0), 2)
    ref = orc.query(scene["ray_dir"], scene["ray_pix"], scene["ray_bid"], scene["pair_ray"].long(),
                    scene["pair_vox"].long(), scene["pair_t"], scene["pair_off"], scene["feat_grid"],
                    scene["vox_feat"], pp, po, off_kind="IMNET", multires=0, multires_views=0)
    s = to_dev(scene, cuda)
    with torch.no_grad():
        got = lidf_query(s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"],
                         s["pair_vox"], s["pair_t"], s["feat_grid"], s["vox_feat"],
                         make_module("IMNET", pp, D, cuda), make_module("IMNET", po, D, cuda),
                         multires=0, multires_views=0, precision=precision)
    for k in ("pred_offset", "pred_prob_end", "pair_pred_pos", "pred_pos"):
        assert (got[k].cpu() - ref[k]).abs().max().item() <= TOL, k


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_sigmoid_outputs_and_three_iterations(cuda, precision):
    from implicit_depth_amd.query import lidf_query
    scene = orc.synthetic_scene(1, 10, 12, 6, seed=8)
    D = scene["D"]
    ref = orc.query(scene["ray_dir"], scene["ray_pix"], scene["ray_bid"], scene["pair_ray"].long(),
                    scene["pair_vox"].long(), scene["pair_t"], scene["pair_off"], scene["feat_grid"],
                    scene["vox_feat"], scene["prob_p"], scene["off_p"], n_iter=3, use_sigmoid=True)
    s = to_dev(scene, cuda)
    with torch.no_grad():
        got = lidf_query(s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"],
                         s["pair_vox"], s["pair_t"], s["feat_grid"], s["vox_feat"],
                         make_module("IMNET", scene["prob_p"], D, cuda, use_sigmoid=True),
                         make_module("IEF", scene["off_p"], D, cuda, n_iter=3, use_sigmoid=True),
                         precision=precision)
    for k in ("pred_offset", "pred_prob_end", "pair_pred_pos", "pred_pos"):
        assert (got[k].cpu() - ref[k]).abs().max().item() <= TOL, k


def test_ray_features_border_pixels(cuda):
    """ROIAlign near the image border: clamped boxes of width 4..7 give fractional sample points."""
    from implicit_depth_amd.query import ray_features
    scene = orc.synthetic_scene(2, 12, 16, 1, seed=9)
    s = to_dev(scene, cuda)
    for bbox in (8, 7, 4):
        got = ray_features(s["feat_grid"], s["ray_dir"], s["ray_pix"], s["ray_bid"], bbox, 4).cpu()
        boxes = orc.roi_boxes(scene["ray_pix"].long(), scene["ray_bid"].long(), 12, 16, bbox)
        ref = orc.roi_align(scene["feat_grid"], boxes).reshape(scene["R"], -1)
        assert (got[:, :128] - ref).abs().max().item() <= 2e-6
        assert (got[:, 128:] - orc.embed(scene["ray_dir"], 4)).abs().max().item() <= 1e-6


@pytest.mark.parametrize("bbox,h,w", [(8, 37, 83), (8, 240, 320), (6, 20, 70), (16, 33, 65), (2, 9, 9)])
def test_ray_features_interior_bins_bit_exact(cuda, bbox, h, w):
    """Unclamped boxes: a bin is a half x half block mean, summed row-wise from 0 and then over the
    rows from 0 — the order of the box-sum kernels (plain and LDS-tiled). Bit-exact against those
    f32 additions replayed on the CPU, for image shapes that are not multiples of the 64 x 16 tile."""
    from implicit_depth_amd.query import ray_features
    g = torch.Generator().manual_seed(bbox + h)
    B = 2
    feat = torch.randn(B, 32, h, w, generator=g)
    feat[0, 0, :2] = 0.0
    feat[0, 1, :, :3] = -0.0          # signed zeros survive the same additions
    half = bbox // 2
    ys, xs = torch.meshgrid(torch.arange(half, h - half), torch.arange(half, w - half), indexing="ij")
    pix = torch.stack((xs.reshape(-1), ys.reshape(-1)), 1).int()
    R1 = pix.shape[0]
    pix = pix.repeat(B, 1).contiguous()
    bid = torch.arange(B).repeat_interleave(R1).int()
    d = torch.nn.functional.normalize(torch.randn(pix.shape[0], 3, generator=g), dim=1)
    got = ray_features(feat.to(cuda), d.to(cuda), pix.to(cuda), bid.to(cuda), bbox, 4).cpu()[:, :128]
    # box sums in the kernels' order
    rows = torch.zeros(B, 32, h, w - half + 1)
    for dx in range(half):
        rows = rows + feat[:, :, :, dx:dx + w - half + 1]
    box = torch.zeros(B, 32, h - half + 1, w - half + 1)
    for dy in range(half):
        box = box + rows[:, :, dy:dy + h - half + 1]
    ref = torch.empty(B, R1, 32, 2, 2)
    x1, y1 = (xs - half).reshape(-1), (ys - half).reshape(-1)
    for ph in range(2):
        for pw in range(2):
            ref[:, :, :, ph, pw] = box[:, :, y1 + ph * half, x1 + pw * half].permute(0, 2, 1) / float(half * half)
    assert torch.equal(got.view(B, R1, 128).view(torch.int32), ref.reshape(B, R1, 128).view(torch.int32))


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_full_size_properties(cuda, precision):
    """BASELINE configs[1] size (240x320x64): properties that need no full-size oracle run."""
    scene = orc.synthetic_scene(1, 240, 320, 64, seed=1235)
    got = run_query(scene, cuda, precision=precision)
    P, R = scene["P"], scene["R"]
    sm = got["pred_prob_end_softmax"]
    assert torch.isfinite(got["pair_pred_pos"]).all() and torch.isfinite(sm).all()
    ray = to_dev(scene, cuda)["pair_ray"].long()
    sums = torch.zeros(R, device=cuda).index_add_(0, ray, sm)
    assert (sums - 1).abs().max().item() <= 1e-5                      # softmax sums to 1 per ray
    mid = got["max_pair_id"]
    assert ((mid >= 0) & (mid < P)).all() and (ray[mid] == torch.arange(R, device=cuda)).all()
    logit = got["pred_prob_end"][:, 0].reshape(R, 64)
    assert (logit.gather(1, (mid - torch.arange(R, device=cuda) * 64).unsqueeze(1))[:, 0]
            >= logit.max(1).values - 1e-6).all()                       # argmax(softmax) is a max logit
    assert (got["pred_pos"] == got["pair_pred_pos"][mid]).all()          # select is a pure gather
    assert (got["depth"].reshape(-1) == got["pred_pos"][:, 2]).all()
    # a random subset of rays against the oracle (whole rays, so the per-ray reduction is covered)
    g = torch.Generator().manual_seed(0)
    rows = torch.randperm(R, generator=g)[:96].sort().values
    pidx = (rows.unsqueeze(1) * 64 + torch.arange(64)).reshape(-1)
    sub_ray = torch.arange(96).repeat_interleave(64)
    ref = orc.query(scene["ray_dir"][rows], scene["ray_pix"][rows], scene["ray_bid"][rows], sub_ray,
                    scene["pair_vox"][pidx].long(), scene["pair_t"][pidx], None, scene["feat_grid"],
                    scene["vox_feat"], scene["prob_p"], scene["off_p"], fast_roi=True)
    for k in ("pred_offset", "pred_prob_end", "pair_pred_pos"):
        assert (got[k][pidx.to(cuda)].cpu() - ref[k]).abs().max().item() <= TOL, k
    assert (got["pred_pos"][rows.to(cuda)].cpu() - ref["pred_pos"]).abs().max().item() <= TOL


def test_ray_reduce_nonfinite_logits(cuda):
    """NaN / all -inf / +inf logits of a ray: torch_scatter's scatter_max leaves its out-of-range
    index (no value compares greater), i.e. the dummy row: id = P, position (0,0,0) — and no
    out-of-bounds read."""
    import ctypes as C
    from implicit_depth_amd import _lib
    off = torch.tensor([0, 3, 6, 9, 9, 12], dtype=torch.int32, device=cuda)
    inf, nan = float("inf"), float("nan")
    prob = torch.tensor([0.1, 0.5, 0.2, nan, nan, nan, -inf, -inf, -inf, inf, 1.0, 2.0], device=cuda)
    pos = torch.arange(36, dtype=torch.float32, device=cuda).reshape(12, 3) + 1
    R, P = 5, 12
    sm = torch.empty(P, device=cuda)
    mid = torch.empty(R, dtype=torch.int64, device=cuda)
    pp = torch.empty(R, 3, device=cuda)
    _lib.check(_lib.lib().lidf_ray_reduce_f32(_lib.ptr(prob), _lib.ptr(pos), _lib.ptr(off), R, P, None,
                                              None, 0, _lib.ptr(sm), _lib.ptr(mid), _lib.ptr(pp), None,
                                              _lib.current_stream(cuda)))
    torch.cuda.synchronize()
    assert mid.tolist() == [1, P, P, P, P]
    assert (pp[0] == pos[1]).all() and (pp[1:] == 0).all()
    ref_sm, ref_id = orc.scatter_softmax(prob.cpu(), torch.tensor([0, 0, 0, 1, 1, 1, 2, 2, 2, 4, 4, 4]), 5), None
    assert torch.allclose(sm[:3].cpu(), ref_sm[:3], atol=1e-6)


def test_embed_large_and_nonfinite(cuda):
    """|x| up to 1e3 (arguments up to 1.3e5 rad at octave 7) against float64 sin/cos of the exact
    f32 products, and NaN / inf inputs -> NaN like torch.sin / torch.cos."""
    from implicit_depth_amd import get_embedder
    g = torch.Generator().manual_seed(11)
    x = (torch.rand(4096, 3, generator=g) - 0.5) * 2000.0
    x[:8] = torch.tensor([[1e3, -1e3, 999.999]]).expand(8, 3)
    fn, dim = get_embedder(8)
    got = fn(x.to(cuda)).cpu().double()
    xd = x.double()
    ref = [xd]
    for o in range(8):
        ref += [torch.sin(xd * 2.0 ** o), torch.cos(xd * 2.0 ** o)]
    ref = torch.cat(ref, -1)
    assert (got - ref).abs().max().item() <= 1e-6
    # and against torch's own f32 sin/cos (the reference's arithmetic): both within a few ulp of exact
    ref32 = orc.embed(x, 8).double()
    assert (got - ref32).abs().max().item() <= 2e-6
    bad = torch.tensor([[float("nan"), float("inf"), -float("inf")], [0.0, 1.0, float("nan")]])
    gb = fn(bad.to(cuda)).cpu()
    rb = orc.embed(bad, 8)
    assert (torch.isnan(gb) == torch.isnan(rb)).all()
    inf = torch.isinf(rb)                      # the passed-through input columns
    assert (gb[inf] == rb[inf]).all()
    ok = torch.isfinite(rb)
    assert (gb[ok] - rb[ok]).abs().max().item() <= 1e-6
