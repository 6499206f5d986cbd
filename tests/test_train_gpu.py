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
    scale = max(1.0, b.abs().max().item())
    err = (a - b).abs().max().item()
    assert err <= 2e-4 * scale, (what, err, scale)


@pytest.mark.parametrize("kind,d,n,n_iter,sig", [("IMNET", 385, 333, 1, False), ("IEF", 385, 333, 2, False),
                                                  ("IEF", 334, 129, 3, True), ("IMNET", 265, 64, 1, True),
                                                  ("IEF", 385, 5000, 2, False), ("IEF", 385, 1, 2, False),
                                                  ("IMNET", 27, 31, 1, False)])
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


@pytest.mark.parametrize("n,v,drop", [(3000, 40, False), (257, 9, True), (5, 3, False)])
def test_pointnet_gradients(cuda, n, v, drop):
    """PointNet2Stage under autograd: HIP forward-with-activations + HIP backward against torch
    autograd on the CPU oracle (scatter-max routes the gradient to one arg row per pooled entry)."""
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
