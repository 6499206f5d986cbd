"""The split-f16 (f16x3, f32 accumulate) per-point kernel: cases specific to it. The generic
query tests (small / ragged / golden g3 / edge options / full size) run for both precisions."""
import pytest
import torch

from util import TOL, make_module, oracle_query, orc, run_query, to_dev

pytestmark = pytest.mark.gpu


def test_one_pair_per_ray(cuda):
    """N = 1: every point of a 32-point tile belongs to a different ray, so the ray part of layer 1
    takes 16 rank-1 rounds per tile instead of one."""
    scene = orc.synthetic_scene(1, 16, 24, 1, seed=41)
    ref = oracle_query(scene)
    got = run_query(scene, cuda, precision="f16x3")
    for k in ("pred_offset", "pred_prob_end", "pair_pred_pos", "pred_pos"):
        assert (got[k].cpu() - ref[k]).abs().max().item() <= TOL, k


def test_close_to_f32_kernel(cuda):
    """Same inputs through both kernels: the split products must stay at f32 rounding level, far
    inside the 1e-4 contract (a plain f16 product would be off by ~1e-3 here)."""
    scene = orc.synthetic_scene(2, 24, 32, 16, seed=42, ragged=True)
    a = run_query(scene, cuda, precision="f32")
    b = run_query(scene, cuda, precision="f16x3")
    for k in ("pred_offset", "pred_prob_end", "pair_pred_pos"):
        assert (a[k] - b[k]).abs().max().item() <= 2e-5, k


def test_large_activations(cuda):
    """Weights scaled up 4x on top of the test scale: activations of a few hundred, still far
    from the f16 range, relative accuracy unchanged."""
    scene = orc.synthetic_scene(1, 12, 16, 8, seed=43, weight_scale=20.0)
    ref = oracle_query(scene)
    got = run_query(scene, cuda, precision="f16x3")
    for k in ("pred_offset", "pred_prob_end"):
        scale = max(1.0, ref[k].abs().max().item())
        assert (got[k].cpu() - ref[k]).abs().max().item() <= TOL * scale, k


def test_unsupported_and_bad_precision(cuda):
    from implicit_depth_amd.query import lidf_query
    scene = orc.synthetic_scene(1, 8, 8, 4, seed=44, multires=10)
    s = to_dev(scene, cuda)
    D = scene["D"]
    prob = make_module("IMNET", orc.init_decoder("IMNET", D, 1, 1.0), D, cuda)
    off = make_module("IEF", orc.init_decoder("IEF", D, 2, 1.0), D, cuda)
    args = (s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"], s["pair_vox"],
            s["pair_t"], s["feat_grid"], s["vox_feat"], prob, off)
    with torch.no_grad():
        lidf_query(*args, multires=10)                      # the f32 kernel takes up to 16 octaves
        with pytest.raises(RuntimeError):
            lidf_query(*args, multires=10, precision="f16x3")
        with pytest.raises(ValueError):
            lidf_query(*args, multires=10, precision="bf16")


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_fewer_octaves(cuda, precision):
    """multires = 4, multires_views = 2 (D = 256 + 2*27 + 15): the split-f16 kernel is built for 8
    octaves and must ignore the unused ones."""
    from implicit_depth_amd.query import lidf_query
    scene = orc.synthetic_scene(1, 12, 16, 8, seed=45, multires=4, multires_views=2)
    D = scene["D"]
    assert D == 256 + 54 + 15
    ref = orc.query(scene["ray_dir"], scene["ray_pix"], scene["ray_bid"], scene["pair_ray"].long(),
                    scene["pair_vox"].long(), scene["pair_t"], scene["pair_off"], scene["feat_grid"],
                    scene["vox_feat"], scene["prob_p"], scene["off_p"], multires=4, multires_views=2)
    s = to_dev(scene, cuda)
    with torch.no_grad():
        got = lidf_query(s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"],
                         s["pair_vox"], s["pair_t"], s["feat_grid"], s["vox_feat"],
                         make_module("IMNET", scene["prob_p"], D, cuda),
                         make_module("IEF", scene["off_p"], D, cuda),
                         multires=4, multires_views=2, precision=precision)
    for k in ("pred_offset", "pred_prob_end", "pair_pred_pos", "pred_pos"):
        assert (got[k].cpu() - ref[k]).abs().max().item() <= TOL, k


@pytest.mark.parametrize("kind,d,n,sig", [("IMNET", 385, 1000, False), ("IEF", 385, 1000, False),
                                          ("IEF", 334, 333, True), ("IMNET", 265, 5, False),
                                          ("IEF", 17, 129, False)])
def test_decoders_boundary_split(cuda, kind, d, n, sig):
    """lidf_decoders_split_f32: the decoder boundary on materialised rows with split-f16 products,
    against the oracle (same tolerance as the f32 kernel) and close to the f32 kernel."""
    from implicit_depth_amd.decoders import decoders_forward
    p = orc.randomize_biases(orc.init_decoder(kind, d, 81, 5.0), 82)
    m = make_module(kind, p, d, cuda, use_sigmoid=sig)
    x = torch.randn(n, d, generator=torch.Generator().manual_seed(n + d))
    ref = orc.decoder_forward(p, x, kind, 2, sig)
    kw = {"prob_dec": m} if kind == "IMNET" else {"offset_dec": m}
    got = [o for o in decoders_forward(x.to(cuda), precision="f16x3", **kw) if o is not None][0]
    f32 = [o for o in decoders_forward(x.to(cuda), precision="f32", **kw) if o is not None][0]
    assert (got.cpu() - ref).abs().max().item() <= TOL
    assert (got - f32).abs().max().item() <= 2e-5


def test_decoders_boundary_split_pair_strided(cuda):
    """Both decoders in one call on a strided view of a wider tensor."""
    from implicit_depth_amd.decoders import decoders_forward
    d = 385
    pp = orc.randomize_biases(orc.init_decoder("IMNET", d, 83, 5.0), 84)
    po = orc.randomize_biases(orc.init_decoder("IEF", d, 85, 5.0), 86)
    mp, mo = make_module("IMNET", pp, d, cuda), make_module("IEF", po, d, cuda)
    wide = torch.randn(700, d + 19, generator=torch.Generator().manual_seed(9)).to(cuda)
    x = wide[:, 7:7 + d]
    gp, go = decoders_forward(x, mp, mo, precision="f16x3")
    assert (gp.cpu() - orc.imnet_forward(pp, x.cpu())).abs().max().item() <= TOL
    assert (go.cpu() - orc.ief_forward(po, x.cpu(), 2)).abs().max().item() <= TOL


@pytest.mark.parametrize("scale", [0.05, 1.0, 5.0, 20.0])
def test_split_f16_weight_magnitudes(cuda, scale):
    """(absolute 1e-4 while |output| <= 1, relative beyond.) The split-f16 products at other weight magnitudes than the benchmark's x5: the reference's
    own initialisation N(0, 0.02) (scale 1.0: the low weight pieces are f16 subnormals), much
    smaller and much larger weights. Errors are measured against the f32 oracle and against the
    exact-f32 kernel; the 1e-4 contract must hold wherever the activations stay inside f16 range."""
    scene = orc.synthetic_scene(1, 16, 24, 16, seed=77, weight_scale=scale)
    ref = oracle_query(scene, fast_roi=True)
    got32 = run_query(scene, cuda, precision="f32")
    got16 = run_query(scene, cuda, precision="f16x3")
    # (pred_pos is a selection by arg-max: with tiny weights all logits of a ray nearly tie, so it
    # is compared through the per-pair outputs it selects from)
    for k in ("pred_offset", "pred_prob_end", "pair_pred_pos"):
        e32 = (got32[k].cpu() - ref[k]).abs().max().item()
        e16 = (got16[k].cpu() - ref[k]).abs().max().item()
        mag = ref[k].abs().max().item()
        assert torch.isfinite(got16[k]).all(), (k, scale)
        tol = TOL * max(1.0, mag)      # large weights give outputs >> 1: the bound scales with them
        assert e32 <= tol, (k, scale, e32, mag)
        assert e16 <= tol, (k, scale, e16, mag)
        # and not materially worse than the exact-f32 kernel relative to the output magnitude
        assert e16 <= max(4 * e32, 2e-6 * max(1.0, mag)), (k, scale, e16, e32, mag)
