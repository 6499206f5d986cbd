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


def _guard_state(entry):
    """(hash, dirty, valid) of a PackedEntry's device guard (struct LidfPackGuardState)."""
    import struct
    raw = bytes(entry.guard.cpu().numpy().tobytes())
    h, acc, ticket, dirty, valid = struct.unpack_from("<QQIii", raw)
    assert acc == 0 and ticket == 0          # re-armed by the block that finished last
    return h, dirty, valid


def test_packed_weights_follow_parameter_updates(cuda):
    """The query keeps its packed weight streams per module and re-validates them on the device
    (fingerprint of the raw parameter buffers): an in-place update, a load_state_dict, a replaced
    .data AND writes through `p.data` — which torch's version counter does not see —
    must all be picked up by the next call; an unchanged module must not be re-packed."""
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

    def ref_off(pp, po):
        return oracle_query(dict(scene, prob_p=pp, off_p=po))
    a = run()
    (entry,) = _lib.packed_entries(_lib.PACK_CACHE, prob)
    h0, dirty, valid = _guard_state(entry)
    assert dirty == 1 and valid == 1                          # first call packed
    b = run()
    assert _lib.packed_entries(_lib.PACK_CACHE, prob) == [entry] and _guard_state(entry) == (h0, 0, 1)   # reused, not re-packed
    twin = copy.deepcopy(prob)     # the cache (device blobs) lives beside the module, not on it
    assert twin not in _lib.PACK_CACHE and "_lidf_pack_cache" not in twin.__dict__
    assert (a["pred_offset"] == b["pred_offset"]).all()
    pp = {k: v.clone() for k, v in scene["prob_p"].items()}
    po = {k: v.clone() for k, v in scene["off_p"].items()}
    with torch.no_grad():
        off.linear_2.weight.mul_(1.25)                                     # in-place (optimizer step)
    po["linear_2.weight"] *= 1.25
    c = run()
    assert _guard_state(entry)[1] == 1
    assert (c["pred_offset"].cpu() - ref_off(pp, po)["pred_offset"]).abs().max().item() <= TOL
    assert (c["pred_offset"] - a["pred_offset"]).abs().max().item() > 1e-4
    # writes through .data: p._version does not move (the hazard of a version-keyed cache)
    v0 = off.linear_1.weight._version
    off.linear_1.weight.data.mul_(0.8)
    off.offset_enc.bias.data.add_(0.03)
    prob.linear_3.bias.data.copy_(torch.full_like(prob.linear_3.bias, 0.02))
    assert off.linear_1.weight._version == v0
    po["linear_1.weight"] *= 0.8
    po["offset_enc.bias"] += 0.03
    pp["linear_3.bias"] = torch.full_like(pp["linear_3.bias"], 0.02)
    d = run()
    assert _guard_state(entry)[1] == 1
    r = ref_off(pp, po)
    assert (d["pred_offset"].cpu() - r["pred_offset"]).abs().max().item() <= TOL
    assert (d["pred_prob_end"].cpu() - r["pred_prob_end"]).abs().max().item() <= TOL
    assert (d["pred_offset"] - c["pred_offset"]).abs().max().item() > 1e-4
    prob.load_state_dict({k: v * 0.5 for k, v in pp.items()})              # copy_ in place
    pp = {k: v * 0.5 for k, v in pp.items()}
    e = run()
    assert (e["pred_prob_end"].cpu() - ref_off(pp, po)["pred_prob_end"]).abs().max().item() <= TOL
    prob.linear_1.weight.data = prob.linear_1.weight.data * 2.0             # replaced storage
    f = run()
    assert (f["pred_prob_end"] - e["pred_prob_end"]).abs().max().item() > 1e-5
    # the IEF's constant initial offset is part of the streams' key (it is baked into the voxel rows)
    off.__dict__["_init_offset_f"] = 0.004
    g = run()
    assert _guard_state(entry)[1] == 1 and (g["pred_offset"] - f["pred_offset"]).abs().max().item() > 1e-6
    off.__dict__["_init_offset_f"] = 0.001
    # another offset decoder with the same prob_dec: the fingerprint covers both modules
    off2 = make_module("IEF", {k: v * 1.1 for k, v in scene["off_p"].items()}, D, cuda)
    off_keep, off = off, off2
    h = run()
    assert _guard_state(entry)[1] == 1
    off = off_keep
    i = run()
    assert torch.equal(i["pred_offset"], run()["pred_offset"]) and (h["pred_offset"] - i["pred_offset"]).abs().max() > 1e-5
    # frozen: packed once more, then trusted — no fingerprint launch, updates are NOT seen until
    # invalidate_packed (the documented contract of the opt-out)
    _lib.freeze_packed(prob)
    j = run()
    hj = _guard_state(_lib.packed_entries(_lib.PACK_CACHE, prob)[0])[0]
    off.linear_2.bias.data.add_(0.5)
    k = run()
    assert torch.equal(j["pred_offset"], k["pred_offset"]) and _guard_state(_lib.packed_entries(_lib.PACK_CACHE, prob)[0])[0] == hj
    _lib.invalidate_packed(prob)
    assert prob not in _lib.PACK_CACHE and prob not in _lib.FROZEN
    m = run()
    assert (m["pred_offset"] - k["pred_offset"]).abs().max().item() > 1e-4
