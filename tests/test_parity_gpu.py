"""GPU parity tests: HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs."""
import pytest
import torch

from util import TOL, make_module, oracle_query, orc, run_query

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("multires", [0, 4, 8])
def test_embed(cuda, multires):
    from implicit_depth_amd import get_embedder
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(1000, 3, generator=g) - 0.5) * 4.6
    x[0] = 0.0
    x[1] = torch.tensor([2.3, -2.3, 1e-7])
    fn, dim = get_embedder(multires)
    got = fn(x.to(cuda)).cpu()
    ref = orc.embed(x, multires)
    assert dim == ref.shape[1] == got.shape[1]
    # sin/cos of arguments up to 2.3*128 rad: 1e-6 absolute leaves room for a few ulp at |v|<=1
    assert (got - ref).abs().max().item() <= 1e-6


@pytest.mark.parametrize("kind,d,n,sig", [("IMNET", 385, 1000, False), ("IEF", 385, 1000, False),
                                          ("IEF", 334, 777, False), ("IMNET", 265, 130, True),
                                          ("IEF", 385, 31, True), ("IEF", 7, 64, False)])
def test_decoder_modules(cuda, kind, d, n, sig):
    p = orc.randomize_biases(orc.init_decoder(kind, d, 11, 5.0), 12)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, d, generator=g)
    ref = orc.decoder_forward(p, x, kind, 2, sig)
    m = make_module(kind, p, d, cuda, 2, sig)
    with torch.no_grad():
        got = m(x.to(cuda)).cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() <= TOL


def test_decoders_pair_and_empty(cuda):
    from implicit_depth_amd import decoders_forward
    d = 385
    pp = orc.randomize_biases(orc.init_decoder("IMNET", d, 1, 5.0), 2)
    po = orc.randomize_biases(orc.init_decoder("IEF", d, 3, 5.0), 4)
    x = torch.randn(513, d, generator=torch.Generator().manual_seed(5))
    mp, mo = make_module("IMNET", pp, d, cuda), make_module("IEF", po, d, cuda)
    with torch.no_grad():
        gp, go = decoders_forward(x.to(cuda), mp, mo)
        ep, eo = decoders_forward(x[:0].to(cuda), mp, mo)
    assert (gp.cpu() - orc.imnet_forward(pp, x)).abs().max().item() <= TOL
    assert (go.cpu() - orc.ief_forward(po, x, 2)).abs().max().item() <= TOL
    assert ep.shape == (0, 1) and eo.shape == (0, 1)


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("ragged", [False, True])
def test_query_small(cuda, ragged, precision):
    scene = orc.synthetic_scene(2, 16, 24, 16, seed=1234, ragged=ragged)
    ref = oracle_query(scene)
    got = run_query(scene, cuda, precision=precision)
    for k in ("pred_offset", "pred_prob_end", "pair_pred_pos", "pred_pos"):
        err = (got[k].cpu() - ref[k]).abs().max().item()
        assert err <= TOL, (k, err)
    assert (got["pred_prob_end_softmax"].cpu() - ref["pred_prob_end_softmax"]).abs().max().item() <= 1e-5
    # arg-max: identical unless the oracle's two best softmax values are within float noise
    gid, rid = got["max_pair_id"].cpu(), ref["max_pair_id"]
    bad = (gid != rid).nonzero().flatten()
    sm = ref["pred_prob_end_softmax"]
    P = scene["P"]
    for r in bad.tolist():
        assert gid[r] < P and rid[r] < P
        assert abs(sm[gid[r]] - sm[rid[r]]) <= 1e-6
    depth_ref = ref["pred_pos"][:, 2].reshape(scene["B"], scene["h"], scene["w"])
    l1 = (got["depth"].cpu() - depth_ref).abs().mean().item()
    assert l1 <= TOL


def test_decoders_strided_rows(cuda):
    """inp_feat given as a view with a row stride larger than D (ld_inp of the C ABI)."""
    from implicit_depth_amd import decoders_forward
    d = 385
    pp = orc.randomize_biases(orc.init_decoder("IMNET", d, 1, 5.0), 2)
    po = orc.randomize_biases(orc.init_decoder("IEF", d, 3, 5.0), 4)
    big = torch.randn(300, 400, generator=torch.Generator().manual_seed(6))
    x = big[:, :d]
    with torch.no_grad():
        gp, go = decoders_forward(big.to(cuda)[:, :d], make_module("IMNET", pp, d, cuda),
                                  make_module("IEF", po, d, cuda))
    assert (gp.cpu() - orc.imnet_forward(pp, x.contiguous())).abs().max().item() <= TOL
    assert (go.cpu() - orc.ief_forward(po, x.contiguous(), 2)).abs().max().item() <= TOL


def test_autograd_path_matches_hip(cuda):
    """With autograd enabled the modules run the library's training path: same values as the
    inference kernel, and gradients reach the parameters and the input (values of the gradients:
    tests/test_train_gpu.py)."""
    d = 385
    p = orc.randomize_biases(orc.init_decoder("IEF", d, 5, 5.0), 6)
    m = make_module("IEF", p, d, cuda)
    x = torch.randn(64, d, generator=torch.Generator().manual_seed(7)).to(cuda)
    with torch.no_grad():
        y_hip = m(x)
    m.train()
    xg = x.clone().requires_grad_(True)
    y = m(xg)
    assert y.requires_grad
    assert (y.detach() - y_hip).abs().max().item() <= TOL
    y.sum().backward()
    assert xg.grad is not None and torch.isfinite(xg.grad).all()
    assert m.linear_1.weight.grad is not None and m.offset_enc.weight.grad is not None
    from implicit_depth_amd import get_embedder
    fn, _ = get_embedder(4)
    pts = torch.randn(10, 3, device=cuda, requires_grad=True)
    e = fn(pts)
    e.sum().backward()
    assert pts.grad is not None
    with torch.no_grad():
        assert (fn(pts.detach()) - e.detach()).abs().max().item() <= 1e-6


def test_packed_weights_follow_parameter_updates(cuda):
    """The query caches its packed weight streams per parameter version: an in-place update, a
    load_state_dict and a replaced .data must all be picked up by the next call."""
    scene = orc.synthetic_scene(1, 8, 12, 4, seed=91)
    import copy
    from implicit_depth_amd import _lib
    from implicit_depth_amd.query import lidf_query
    from util import to_dev
    s = to_dev(scene, cuda)
    D = scene["D"]
    prob, off = make_module("IMNET", scene["prob_p"], D, cuda), make_module("IEF", scene["off_p"], D, cuda)

    def run():
        with torch.no_grad():
            return lidf_query(s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"],
                              s["pair_vox"], s["pair_t"], s["feat_grid"], s["vox_feat"], prob, off)
    a = run()
    cache0 = _lib.PACK_CACHE[prob][1]
    b = run()
    assert _lib.PACK_CACHE[prob][1] is cache0                 # reused
    twin = copy.deepcopy(prob)     # the cache (device blob + event) lives beside the module, not on it
    assert twin not in _lib.PACK_CACHE and "_lidf_pack_cache" not in twin.__dict__
    assert (a["pred_offset"] == b["pred_offset"]).all()
    with torch.no_grad():
        off.linear_2.weight.mul_(1.25)                                     # in-place (optimizer step)
    c = run()
    p2 = {k: v.clone() for k, v in scene["off_p"].items()}
    p2["linear_2.weight"] = p2["linear_2.weight"] * 1.25
    ref = oracle_query(dict(scene, off_p=p2))
    assert (c["pred_offset"].cpu() - ref["pred_offset"]).abs().max().item() <= TOL
    assert (c["pred_offset"] - a["pred_offset"]).abs().max().item() > 1e-4
    prob.load_state_dict({k: v * 0.5 for k, v in scene["prob_p"].items()})  # copy_ in place
    d = run()
    ref = oracle_query(dict(scene, off_p=p2, prob_p={k: v * 0.5 for k, v in scene["prob_p"].items()}))
    assert (d["pred_prob_end"].cpu() - ref["pred_prob_end"]).abs().max().item() <= TOL
    prob.linear_1.weight.data = prob.linear_1.weight.data * 2.0             # replaced storage
    e = run()
    assert (e["pred_prob_end"] - d["pred_prob_end"]).abs().max().item() > 1e-5
