"""Training path of the decoders (first step): HIP forward that keeps activations +
HIP backward behind torch.autograd.Function, against torch autograd of the composite definition
(models/implicit_net.py:81-98 / :131-152) on the same device and against the CPU oracle."""
import pytest
import torch

from util import TOL, make_module, orc

pytestmark = pytest.mark.gpu


def _grads(module, x, fn, weight):
    for p in module.parameters():
        p.grad = None
    xg = x.clone().requires_grad_(True)
    y = fn(xg)
    (y.reshape(-1) * weight).sum().backward()
    g = {k: p.grad.detach().clone() for k, p in module.named_parameters()}
    g["input"] = xg.grad.detach().clone()
    return y.detach(), g


def _close(a, b, what):
    if b.numel() == 0:                     # (an empty batch: shapes only)
        assert a.shape == b.shape, what
        return
    scale = max(1.0, b.abs().max().item())
    err = (a - b).abs().max().item()
    assert err <= 2e-4 * scale, (what, err, scale)


@pytest.mark.parametrize("kind,d,n,n_iter,sig", [("IMNET", 385, 333, 1, False), ("IEF", 385, 333, 2, False),
                                                  ("IEF", 334, 129, 3, True), ("IMNET", 265, 64, 1, True),
                                                  ("IEF", 385, 5000, 2, False), ("IEF", 385, 1, 2, False),
                                                  ("IMNET", 27, 31, 1, False),
                                                  # (more than one middle pass of the IEF: the running sum of dZ1 over
                                                  # the passes takes the first pass processed, adds the middle ones in
                                                  # the encoding sweep and the last one in the chained launch)
                                                  ("IEF", 385, 700, 4, False), ("IEF", 102, 2500, 3, True),
                                                  # (32 t + 1 input columns over more than 2,048 rows: the input
                                                  # gradient's last column through the vector unit — 2, 7 and 8 tiles
                                                  # in front of it; 8 tiles leave no room for it: a ninth tile)
                                                  ("IMNET", 65, 3000, 1, False), ("IEF", 225, 2600, 2, True),
                                                  ("IMNET", 257, 2100, 1, False)])
def test_backward_matches_autograd(cuda, kind, d, n, n_iter, sig):
    p = orc.randomize_biases(orc.init_decoder(kind, d, 11, 5.0), 12)
    m = make_module(kind, p, d, cuda, n_iter=n_iter, use_sigmoid=sig).train()
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, d, generator=g).to(cuda)
    wgt = torch.randn(n, generator=g).to(cuda)          # a non-trivial upstream gradient
    y_hip, g_hip = _grads(m, x, m, wgt)                  # autograd.Function -> liblidf_hip
    y_ref, g_ref = _grads(m, x, m.forward_composite, wgt)
    assert (y_hip - y_ref).abs().max().item() <= TOL
    with torch.no_grad():
        assert (m(x) - y_hip).abs().max().item() <= 1e-5  # same values as the inference kernel
    assert set(g_hip) == set(g_ref)
    for k in g_ref:
        _close(g_hip[k], g_ref[k], k)


@pytest.mark.parametrize("d,n,n_iter,sig,use", [(385, 333, 2, False, "both"), (385, 5000, 2, False, "both"),
                                                 (334, 129, 3, True, "both"), (385, 1, 2, False, "both"),
                                                 (385, 700, 1, False, "both"), (27, 31, 2, False, "both"),
                                                 (97, 2300, 2, False, "both"), (257, 2100, 2, False, "both"), (385, 0, 2, False, "both"),
                                                 (385, 900, 2, False, "prob"), (385, 900, 2, False, "off")])
def test_decoder_pair_node_matches_the_two_modules(cuda, d, n, n_iter, sig, use):
    """decoders_forward_train (prob_dec(inp), offset_dec(inp) of pipeline.py:434-435 as ONE autograd node, the rows'
    gradient as one K = 512 product over both decoders' summed dZ1): values equal the two modules' bit for bit, the
    gradients equal the torch-op definition's to the tolerance of the single-module test and the two modules' own
    sum to float noise; an output the loss does not use contributes nothing. (Row counts are chosen away from
    leaky-ReLU kinks: at (257, 2200) one row has a layer-3 pre-activation of 1.4e-7, whose sign — and with it that
    row's gradient — differs between any two float32 evaluations.)"""
    from implicit_depth_amd import decoders_forward_train
    pp = orc.randomize_biases(orc.init_decoder("IMNET", d, 21, 5.0), 22)
    po = orc.randomize_biases(orc.init_decoder("IEF", d, 23, 5.0), 24)
    prob = make_module("IMNET", pp, d, cuda, use_sigmoid=sig).train()
    off = make_module("IEF", po, d, cuda, n_iter=n_iter, use_sigmoid=sig).train()
    g = torch.Generator().manual_seed(n + d)
    x = torch.randn(n, d, generator=g).to(cuda)
    wp = torch.randn(n, generator=g).to(cuda)
    wo = torch.randn(n, generator=g).to(cuda)

    def run(fn):
        for m in (prob, off):
            for q in m.parameters():
                q.grad = None
        xg = x.clone().requires_grad_(True)
        yp, yo = fn(xg)
        loss = 0.0
        if use in ("both", "prob"):
            loss = loss + (yp.reshape(-1) * wp).sum()
        if use in ("both", "off"):
            loss = loss + (yo.reshape(-1) * wo).sum()
        loss.backward()
        gr = {"prob." + k: (q.grad.detach().clone() if q.grad is not None else torch.zeros_like(q))
              for k, q in prob.named_parameters()}
        gr.update({"off." + k: (q.grad.detach().clone() if q.grad is not None else torch.zeros_like(q))
                   for k, q in off.named_parameters()})
        gr["input"] = xg.grad.detach().clone()
        return yp.detach(), yo.detach(), gr

    yp1, yo1, g1 = run(lambda t: decoders_forward_train(t, prob, off))
    yp2, yo2, g2 = run(lambda t: (prob(t), off(t)))
    yp3, yo3, g3 = run(lambda t: (prob.forward_composite(t), off.forward_composite(t)))
    assert torch.equal(yp1, yp2) and torch.equal(yo1, yo2)
    assert set(g1) == set(g2) == set(g3)
    for k in g3:
        _close(g1[k], g3[k], k)
        if k != "input":
            assert torch.equal(g1[k], g2[k]), k      # the parameter gradients are the same launches
    # one product over K = 512 against two over K = 256 and an add: summation order only
    if n:
        scale = max(1.0, g2["input"].abs().max().item())
        assert (g1["input"] - g2["input"]).abs().max().item() <= 2e-6 * scale
    # run-to-run identical
    _, _, g1b = run(lambda t: decoders_forward_train(t, prob, off))
    for k in g1:
        assert torch.equal(g1[k], g1b[k]), k
    # without anything to differentiate the call is the inference launch of both decoders
    with torch.no_grad():
        yp4, yo4 = decoders_forward_train(x, prob, off)
    assert not yp4.requires_grad and yp4.shape == yp1.shape
    if n:
        assert (yp4 - yp1).abs().max().item() <= 1e-5 and (yo4 - yo1).abs().max().item() <= 1e-5
    # frozen offset decoder (trainers/train_refine.py freezes stage 1 the same way): no gradient tensors for it
    for q in off.parameters():
        q.requires_grad_(False)
        q.grad = None
    for q in prob.parameters():
        q.grad = None
    yp5, yo5 = decoders_forward_train(x, prob, off)
    (yp5.reshape(-1) * wp).sum().backward()
    assert all(q.grad is None for q in off.parameters()) and all(q.grad is not None for q in prob.parameters())
    for q in off.parameters():
        q.requires_grad_(True)


def test_rows_backward_is_run_to_run_identical(cuda):
    """The decoders' backward on rows sums in a fixed order (slab reductions, no float atomics on the wide layers):
    two runs of the same step give the same bits for the input gradient and every wide parameter gradient."""
    d, n = 385, 4099
    p = orc.randomize_biases(orc.init_decoder("IEF", d, 31, 5.0), 32)
    m = make_module("IEF", p, d, cuda, n_iter=3).train()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(n, d, generator=g).to(cuda)
    wgt = torch.randn(n, generator=g).to(cuda)
    _, g1 = _grads(m, x, m, wgt)
    _, g2 = _grads(m, x, m, wgt)
    for k in ("input", "linear_1.weight", "linear_1.bias", "linear_2.weight", "linear_3.weight"):
        assert torch.equal(g1[k], g2[k]), k


def test_backward_matches_cpu_oracle(cuda):
    """The same gradients from the oracle's torch-CPU restatement (independent of the product's
    composite definition)."""
    d, n = 385, 257
    p = orc.randomize_biases(orc.init_decoder("IEF", d, 21, 5.0), 22)
    m = make_module("IEF", p, d, cuda, n_iter=2).train()
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(n, d, generator=gen)
    wgt = torch.randn(n, generator=gen)
    _, g_hip = _grads(m, x.to(cuda), m, wgt.to(cuda))
    pc = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xc = x.clone().requires_grad_(True)
    y = orc.ief_forward(pc, xc, 2)
    (y.reshape(-1) * wgt).sum().backward()
    _close(g_hip["input"].cpu(), xc.grad, "input")
    for k, v in pc.items():
        _close(g_hip[k].cpu(), v.grad, k)


def test_training_step_reduces_loss(cuda):
    """A few SGD steps through the HIP forward/backward fit a small regression target."""
    d, n = 385, 2048
    m = make_module("IEF", orc.init_decoder("IEF", d, 31, 5.0), d, cuda, n_iter=2).train()
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(n, d, generator=gen).to(cuda)
    target = torch.rand(n, 1, generator=gen).to(cuda) * 0.5 + 0.2
    opt = torch.optim.SGD(m.parameters(), lr=0.05)
    losses = []
    for _ in range(12):
        opt.zero_grad()
        loss = ((m(x) - target) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    # (plain SGD at this step size hovers around its floor: the best loss of the run is the criterion)
    assert min(losses) < 0.7 * losses[0] and losses[-1] < 0.8 * losses[0], losses


def test_empty_batch_and_no_input_grad(cuda):
    d = 385
    m = make_module("IMNET", orc.init_decoder("IMNET", d, 41, 5.0), d, cuda).train()
    y = m(torch.zeros(0, d, device=cuda))
    assert y.shape == (0, 1)
    y.sum().backward()
    assert all(p.grad is not None and (p.grad == 0).all() for p in m.parameters())
    x = torch.randn(50, d, device=cuda)                  # no grad on the input: weights only
    for p in m.parameters():
        p.grad = None
    m(x).sum().backward()
    assert m.linear_1.weight.grad.abs().sum().item() > 0


@pytest.mark.parametrize("kind,n_iter", [("IMNET", 1), ("IEF", 2)])
def test_backward_against_f64(cuda, kind, n_iter):
    """Tighter: the same gradients from the composite definition evaluated in float64 on the GPU.
    f32 rounding level is all that may separate them (seeded inputs without a pre-activation at
    the leaky-ReLU kink)."""
    d, n = 385, 300
    p = orc.randomize_biases(orc.init_decoder(kind, d, 61, 5.0), 62)
    m = make_module(kind, p, d, cuda, n_iter=n_iter).train()
    gen = torch.Generator().manual_seed(17)
    x = torch.randn(n, d, generator=gen).to(cuda)
    wgt = torch.randn(n, generator=gen).to(cuda)
    _, g_hip = _grads(m, x, m, wgt)
    md = make_module(kind, p, d, cuda, n_iter=n_iter).double().train()
    if kind == "IEF":
        md.init_offset = md.init_offset.double()
    _, g_ref = _grads(md, x.double(), md.forward_composite, wgt.double())
    for k in g_ref:
        scale = max(1e-3, g_ref[k].abs().max().item())
        err = (g_hip[k].double() - g_ref[k]).abs().max().item()
        assert err <= 2e-5 * scale, (k, err, scale)


# (70000, 300): more 128-point tiles than workgroups and voxels beyond one workgroup's 32-row pooling window;
# (20000, 5000): a handful of points per voxel — a workgroup's run of sorted points spans far more voxels than its
# window, most entries go through the global 64-bit maxima; (1000, 1): one voxel, one long run of rows (chunked
# row sums); (4000, 64) with a quarter of the rows left out: the sorted buffers' dead rows are written as zeros
@pytest.mark.parametrize("n,v,drop", [(3000, 40, False), (257, 9, True), (5, 3, False), (70000, 300, False),
                                      (20000, 5000, False), (1000, 1, False), (4000, 64, True)])
def test_pointnet_gradients(cuda, n, v, drop):
    """PointNet2Stage under autograd: HIP forward-with-activations + HIP backward against torch
    autograd on the CPU oracle (scatter-max routes the gradient to one arg row per pooled entry). Round 5: the
    register chains of lidf_pointnet_train.hip over voxel-sorted points; LIDF_PNET_TRAIN_CHAIN=0 selects the
    layer-by-layer path of rounds 2-4 (tests/test_train_gpu.py::test_pointnet_train_paths_agree)."""
    from util import make_pointnet
    g = torch.Generator().manual_seed(n + v)
    p = orc.init_pointnet(7, 1.5)
    inp = torch.randn(n, 6, generator=g)
    vox = torch.randint(0, v, (n,), generator=g)
    if drop:
        vox[torch.rand(n, generator=g) < 0.2] = -1          # rows left out of the poolings
    wgt = torch.randn(v, 128, generator=g)
    # oracle autograd
    pr = {k: t.clone().requires_grad_(True) for k, t in p.items()}
    xi = inp.clone().requires_grad_(True)
    keep = vox >= 0
    ref = orc.pointnet2stage(pr, xi[keep], vox[keep], v)
    (ref * wgt).sum().backward()
    m = make_pointnet(p, cuda).train()
    xd = inp.to(cuda).requires_grad_(True)
    out = m(xd, vox.to(cuda), n_vox=v)
    assert (out.detach().cpu() - ref.detach()).abs().max().item() <= 2e-5
    (out * wgt.to(cuda)).sum().backward()

    def close(a, b, what):
        scale = max(1e-2, b.abs().max().item())
        assert (a - b).abs().max().item() <= 5e-4 * scale, (what, (a - b).abs().max().item(), scale)
    close(xd.grad.cpu(), xi.grad, "inp")
    for k, t in pr.items():
        close(dict(m.named_parameters())[k].grad.cpu(), t.grad, k)
    # and the inference kernel gives the same values
    with torch.no_grad():
        assert (m(inp.to(cuda), vox.to(cuda), n_vox=v) - out.detach()).abs().max().item() <= 1e-6
    # the chains' sums run in a fixed order (sorted rows, slab reductions, arg rows by a 64-bit maximum): a second
    # run gives the same bits
    m2 = make_pointnet(p, cuda).train()
    xd2 = inp.to(cuda).requires_grad_(True)
    out2 = m2(xd2, vox.to(cuda), n_vox=v)
    (out2 * wgt.to(cuda)).sum().backward()
    assert torch.equal(out2, out) and torch.equal(xd2.grad, xd.grad)
    for (k, a), (_, b) in zip(m.named_parameters(), m2.named_parameters()):
        assert torch.equal(a.grad, b.grad), k


@pytest.mark.parametrize("multires", [0, 4, 8])
def test_embed_gradient(cuda, multires):
    from implicit_depth_amd import get_embedder
    fn, dim = get_embedder(multires)
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(500, 3, generator=g) - 0.5) * 4.0
    wgt = torch.randn(500, dim, generator=g)
    xr = x.clone().double().requires_grad_(True)
    outs = [xr] + [f(xr * 2.0 ** o) for o in range(multires) for f in (torch.sin, torch.cos)]
    (torch.cat(outs, -1) * wgt.double()).sum().backward()
    xd = x.to(cuda).requires_grad_(True)
    e = fn(xd)
    assert e.shape == (500, dim)
    (e * wgt.to(cuda)).sum().backward()
    scale = max(1.0, xr.grad.abs().max().item())
    assert (xd.grad.cpu().double() - xr.grad).abs().max().item() <= 2e-5 * scale


def test_refine_train_gradients_vs_oracle(cuda):
    """Stage-2 training step (RefineNet.forward with exp_type 'train', train_refine.py:393-399):
    lidf_refine_train's outputs and the gradients of every PointNet2Stage / IEF parameter against
    torch autograd through the oracle's refine_step chain on the CPU, with the train-only perturbation
    (pipeline.py:925-937) drawn by the reference's own np.random calls; the inference call
    (lidf_refine) must give the same forward values."""
    import numpy as np
    from implicit_depth_amd.query import lidf_query, lidf_refine, lidf_refine_train, refine_perturb_noise
    from util import make_module, make_pointnet, to_dev
    scene = orc.synthetic_scene(2, 10, 14, 5, seed=31, ragged=True)
    s = to_dev(scene, cuda)
    D = scene["D"]
    prob, off = make_module("IMNET", scene["prob_p"], D, cuda), make_module("IEF", scene["off_p"], D, cuda)
    with torch.no_grad():
        s1 = lidf_query(s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"],
                        s["pair_vox"], s["pair_t"], s["feat_grid"], s["vox_feat"], prob, off)
    g = torch.Generator().manual_seed(9)
    V = scene["V"]
    vb = torch.cat((scene["vox_center"] - 0.125, scene["vox_center"] + 0.125), 1)
    vbid = torch.arange(2).repeat_interleave(729).int()
    rgb = torch.randn(2, 3, 10, 14, generator=g)
    valid_inp = torch.randn(400, 6, generator=g) * 0.2
    valid_vox = torch.randint(0, V, (400,), generator=g).int()
    pnet_p = orc.init_pointnet(5, 1.5)
    offr_p = orc.randomize_biases(orc.init_decoder("IEF", 334, 77, 5.0), 78)
    np.random.seed(4)
    noise = refine_perturb_noise(perturb_prob=1.0)
    assert noise is not None and abs(noise) <= 0.1
    np.random.seed(4)            # the reference's draws, in its order (pipeline.py:926-936)
    assert np.random.random() < 1.0
    pr = np.random.random()
    ref_noise = (np.random.random() * 0.05 - 0.05 if pr < 0.5 else np.random.random() * 0.05 if pr < 0.8
                 else np.random.random() * 0.05 - 0.1 if pr < 0.9 else np.random.random() * 0.05 + 0.05)
    assert noise == ref_noise
    # --- oracle: autograd through two refine_step iterations on the CPU
    pn_ref = {k: v.clone().requires_grad_(True) for k, v in pnet_p.items()}
    of_ref = {k: v.clone().requires_grad_(True) for k, v in offr_p.items()}
    pos = s1["pred_pos"].cpu() + noise * scene["ray_dir"]
    for _ in range(2):
        pos, ev, _ = orc.refine_step(pos, scene["ray_dir"], scene["ray_pix"], scene["ray_bid"],
                                     scene["ray_flat"], s1["max_pair_id"].cpu(), scene["pair_vox"], vb, vbid,
                                     rgb, scene["feat_grid"], valid_inp, valid_vox, pn_ref, of_ref)
    wgt = torch.randn(pos.shape, generator=g)
    (pos * wgt).sum().backward()
    # --- product: the training path
    pnet, offr = make_pointnet(pnet_p, cuda).train(), make_module("IEF", offr_p, 334, cuda).train()
    args = (s["ray_dir"], s["ray_pix"], s["ray_bid"], s["ray_flat"], s1["pred_pos"], s1["max_pair_id"],
            s["pair_vox"], vb.to(cuda), vbid.to(cuda), rgb.to(cuda), s["feat_grid"], valid_inp.to(cuda),
            valid_vox.to(cuda), pnet, offr)
    got, gev = lidf_refine_train(*args, perturb_noise=noise)
    assert got.requires_grad and (gev.cpu().long() == ev).all()
    assert (got.detach().cpu() - pos.detach()).abs().max().item() <= TOL
    (got * wgt.to(cuda)).sum().backward()
    for mod, ref in ((pnet, pn_ref), (offr, of_ref)):
        for k, p in mod.named_parameters():
            gr = ref[k].grad
            assert p.grad is not None, k
            err = (p.grad.cpu() - gr).abs().max().item()
            assert err <= 5e-4 * max(gr.abs().max().item(), 1e-3), (k, err, gr.abs().max().item())
    # the inference call on the perturbed start gives the same forward values
    with torch.no_grad():
        inf, _ = lidf_refine(*args[:4], (s1["pred_pos"] + noise * s["ray_dir"]).contiguous(), *args[5:])
    assert (inf - got.detach()).abs().max().item() <= 2e-5
    with pytest.raises(RuntimeError, match="inference path"):
        lidf_refine(*args)       # autograd recording + trainable modules: refused, not silently detached


@pytest.mark.parametrize("pos_rel,pnet_pos_rel,kind,use_grid", [(False, True, "IEF", False), (True, False, "IEF", True),
                                                                (False, True, "IMNET", False)])
def test_refine_train_fused_step_vs_composed(cuda, pos_rel, pnet_pos_rel, kind, use_grid):
    """lidf_refine_train as two library calls (one autograd node: lidf_refine_train_forward_f32 / _backward_f32,
    the decoder factorised, parameter gradients summed over the iterations inside the call) against the same
    step composed from the modules' own autograd functions joined by torch ops (rounds 3-4, itself checked
    against the oracle's autograd above): positions, end voxels, the gradient of every PointNet2Stage / decoder
    parameter, of the incoming position and of the feature map (RoIAlign backward behind the per-ray layer-1
    table); three iterations; both position types; the end voxels through the cell table; and run to run the
    fused step's gradients are bit-identical."""
    from implicit_depth_amd.query import (_lidf_refine_train_composed, get_occ_vox_bound, lidf_refine_train)
    from util import make_module, make_pointnet
    g = torch.Generator().manual_seed(17)
    B, h, w = 2, 20, 24
    # occupied voxels from random points (so that the cell table of get_occ_vox_bound applies)
    pts = torch.rand(600, 3, generator=g) * torch.tensor([1.6, 1.6, 1.6]) + torch.tensor([-0.8, -0.8, 0.2])
    pb = torch.randint(0, B, (600,), generator=g).int()
    occ = get_occ_vox_bound(pts.to(cuda), pb.to(cuda), B, res=8)
    vb, vbid = occ["voxel_bound"], occ["occ_vox_bid"].int().contiguous()
    V = vb.shape[0]
    R = 700
    bid = torch.randint(0, B, (R,), generator=g)
    ctr = ((vb[:, :3] + vb[:, 3:]) / 2).cpu()
    own = [torch.nonzero(vbid.cpu() == b)[:, 0] for b in range(B)]
    pick = torch.stack([own[int(b)][torch.randint(0, own[int(b)].numel(), (1,), generator=g)][0] for b in bid])
    pos0 = ctr[pick] + (torch.rand(R, 3, generator=g) - 0.5) * 0.3
    ray_dir = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=1)
    flat = torch.randint(0, h * w, (R,), generator=g)
    ray_pix = torch.stack((flat % w, flat // w), 1).int()
    P = 900
    pair_vox = torch.randint(0, V, (P,), generator=g).int()
    mid = torch.randint(0, P + 1, (R,), generator=g)
    rgb = torch.randn(B, 3, h, w, generator=g)
    feat = torch.randn(B, 32, h, w, generator=g)
    valid_inp = torch.randn(500, 6, generator=g) * 0.2
    valid_vox = torch.randint(0, V, (500,), generator=g).int()
    wgt = torch.randn(R, 3, generator=g).to(cuda)
    pnet_p = orc.init_pointnet(5, 1.5)
    dec_p = orc.randomize_biases(orc.init_decoder(kind, 334, 77, 5.0), 78)

    def run(fn, **kw):
        pnet, dec = make_pointnet(pnet_p, cuda).train(), make_module(kind, dec_p, 334, cuda).train()
        pp = pos0.to(cuda).requires_grad_(True)
        fg = feat.to(cuda).requires_grad_(True)
        got, ev = fn(ray_dir.to(cuda), ray_pix.to(cuda), bid.int().to(cuda), flat.int().to(cuda), pp, mid.to(cuda),
                     pair_vox.to(cuda), vb, vbid, rgb.to(cuda), fg, valid_inp.to(cuda), valid_vox.to(cuda), pnet, dec,
                     forward_times=3, pos_rel=pos_rel, pnet_pos_rel=pnet_pos_rel, perturb_noise=0.03, **kw)
        (got * wgt).sum().backward()
        grads = {"pred_pos": pp.grad, "feat_grid": fg.grad}
        for name, mod in (("pnet", pnet), ("dec", dec)):
            for k, p in mod.named_parameters():
                assert p.grad is not None, k
                grads[name + "." + k] = p.grad
        return got.detach(), ev, grads

    ref_pos, ref_ev, ref_g = run(_lidf_refine_train_composed)
    got_pos, got_ev, got_g = run(lidf_refine_train, grid=occ if use_grid else None)
    assert torch.equal(got_ev, ref_ev)
    assert (got_pos - ref_pos).abs().max().item() <= TOL     # (three chained iterations, factorised vs materialised layer 1)
    for k, gr in ref_g.items():
        err = (got_g[k] - gr).abs().max().item()
        # (two f32 evaluation orders of the same function chained over three iterations; each is held to 5e-4 of
        # the oracle's autograd at two iterations by test_refine_train_gradients_vs_oracle)
        assert err <= 1e-3 * max(gr.abs().max().item(), 1e-3), (k, err, gr.abs().max().item())
    again_pos, _, again_g = run(lidf_refine_train, grid=occ if use_grid else None)
    assert torch.equal(again_pos, got_pos)
    for k in got_g:
        # (float atomics remain in the RoIAlign backward's border taps only — lidf_hip.h; everything else of the
        # step is a fixed-order sum: sorted rows, slab reductions, arg rows by a 64-bit maximum)
        if k != "feat_grid":
            assert torch.equal(again_g[k], got_g[k]), k
        else:
            assert (again_g[k] - got_g[k]).abs().max().item() <= 1e-5 * max(got_g[k].abs().max().item(), 1e-3), k


def test_pointnet_train_paths_agree(cuda):
    """The training chains (lidf_pointnet_train.hip) against the layer-by-layer training path of rounds 2-4
    (LIDF_PNET_TRAIN_CHAIN=0, a separate process: the switch is read once per process): outputs and every gradient
    within f32 re-association of each other on the same inputs."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from util import make_pointnet, orc
g = torch.Generator().manual_seed(11)
n, v = 30000, 120
p = orc.init_pointnet(7, 1.5)
inp = torch.randn(n, 6, generator=g); vox = torch.randint(0, v, (n,), generator=g); wgt = torch.randn(v, 128, generator=g)
m = make_pointnet(p, 'cuda').train()
x = inp.cuda().requires_grad_(True)
out = m(x, vox.cuda(), n_vox=v)
(out * wgt.cuda()).sum().backward()
res = {'out': out.detach().cpu(), 'x': x.grad.cpu()}
res.update({k: t.grad.cpu() for k, t in m.named_parameters()})
torch.save(res, sys.argv[1])
""" % (root, os.path.join(root, "tests"))
    outs = []
    with tempfile.TemporaryDirectory() as d:
        for flag in ("1", "0"):
            f = os.path.join(d, "r%s.pt" % flag)
            r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, LIDF_PNET_TRAIN_CHAIN=flag),
                               capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-3000:]
            outs.append(torch.load(f))
    a, b = outs
    for k in a:
        scale = max(1e-2, b[k].abs().max().item())
        assert (a[k] - b[k]).abs().max().item() <= 2e-4 * scale, (k, (a[k] - b[k]).abs().max().item(), scale)


def test_refine_train_node_frees_its_buffers_without_the_cycle_collector(cuda):
    """The stage-2 training node keeps no closure over its own output (out -> grad_fn -> ctx ->
    closure -> out kept ~1 GB of workspace per step alive until Python's cycle collector ran): with the
    collector disabled, dropping the outputs — with or without a backward — returns the device memory."""
    import gc
    from implicit_depth_amd.query import get_occ_vox_bound, lidf_refine_train
    from util import make_module, make_pointnet
    g = torch.Generator().manual_seed(27)
    B, h, w = 1, 16, 20
    pts = torch.rand(300, 3, generator=g) * 1.6 + torch.tensor([-0.8, -0.8, 0.2])
    occ = get_occ_vox_bound(pts.to(cuda), torch.zeros(300, dtype=torch.int32, device=cuda), B, res=8)
    vb, vbid = occ["voxel_bound"], occ["occ_vox_bid"].int().contiguous()
    V, R, P = vb.shape[0], 400, 500
    ctr = ((vb[:, :3] + vb[:, 3:]) / 2).cpu()
    pos0 = ctr[torch.randint(0, V, (R,), generator=g)] + (torch.rand(R, 3, generator=g) - 0.5) * 0.3
    flat = torch.randint(0, h * w, (R,), generator=g)
    args = dict(ray_dir=torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=1).to(cuda),
                ray_pix=torch.stack((flat % w, flat // w), 1).int().to(cuda),
                ray_bid=torch.zeros(R, dtype=torch.int32, device=cuda), ray_flat=flat.int().to(cuda),
                max_pair_id=torch.randint(0, P + 1, (R,), generator=g).to(cuda),
                pair_vox=torch.randint(0, V, (P,), generator=g).int().to(cuda), voxel_bound=vb, voxel_bid=vbid,
                rgb_img=torch.randn(B, 3, h, w, generator=g).to(cuda),
                feat_grid=torch.randn(B, 32, h, w, generator=g).to(cuda),
                valid_inp=(torch.randn(200, 6, generator=g) * 0.2).to(cuda),
                valid_vox=torch.randint(0, V, (200,), generator=g).int().to(cuda))
    pnet = make_pointnet(orc.init_pointnet(5, 1.5), cuda).train()
    dec = make_module("IEF", orc.randomize_biases(orc.init_decoder("IEF", 334, 77, 5.0), 78), 334, cuda).train()

    def step(backward):
        out, ev = lidf_refine_train(pred_pos=pos0.to(cuda), pnet_model=pnet, offset_dec=dec, grid=occ, **args)
        if backward:
            out.sum().backward()     # (an expanded stride-0 output gradient, too)
    step(True)                       # allocator pools, gradient buffers
    torch.cuda.synchronize()
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        base = torch.cuda.memory_allocated(cuda)
        for backward in (False, True, False):
            step(backward)
            torch.cuda.synchronize()
            assert torch.cuda.memory_allocated(cuda) <= base + 4096, (backward, torch.cuda.memory_allocated(cuda), base)
    finally:
        if was:
            gc.enable()


def test_grid_dict_that_does_not_describe_the_voxels_is_refused_when_validated(cuda, monkeypatch):
    """LIDF_VALIDATE_GRID=1: a grid dict whose origin is not the widened one of get_occ_vox_bound raises instead
    of silently giving other end voxels than the every-voxel test."""
    from implicit_depth_amd.query import _cell_lookup, get_occ_vox_bound
    g = torch.Generator().manual_seed(3)
    pts = torch.rand(200, 3, generator=g) * 1.6 + torch.tensor([-0.8, -0.8, 0.2])
    occ = get_occ_vox_bound(pts.to(cuda), torch.zeros(200, dtype=torch.int32, device=cuda), 1, res=8)
    V = occ["voxel_bound"].shape[0]
    monkeypatch.setenv("LIDF_VALIDATE_GRID", "1")
    assert _cell_lookup(occ, V, 1, cuda, occ["voxel_bound"]) is not None
    bad = dict(occ)
    xm = occ["xmin"]
    bad["xmin"] = [float(v) + 0.5 * float(occ["part_size"]) for v in (xm.tolist() if torch.is_tensor(xm) else xm)]
    with pytest.raises(RuntimeError):
        _cell_lookup(bad, V, 1, cuda, occ["voxel_bound"])
