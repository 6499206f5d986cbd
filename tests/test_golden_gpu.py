"""GPU: the HIP path against the golden vectors generated from the reference
(tests/golden/make_golden.py) — embed, both decoders, and the full stage-1 query trace."""
import numpy as np
import pytest
import torch

from test_oracle_golden import g2_cases, g3_inputs, load
from util import TOL, closed_form, closed_form_params, make_module

pytestmark = pytest.mark.gpu


def test_g1_embed_hip(cuda):
    from implicit_depth_amd import get_embedder
    g = load("g1_embed.npz")
    x = torch.from_numpy(g["x"]).to(cuda)
    for L in (8, 4):
        got = get_embedder(L)[0](x).cpu().numpy()
        assert np.abs(got - g["embed_L%d" % L]).max() <= 1e-6


def test_g2_decoders_hip(cuda):
    g = load("g2_decoders.npz")
    for key, kind, d, sig, seed in g2_cases(g):
        m = make_module(kind, closed_form_params(kind, d, seed=seed), d, cuda, 2, sig)
        x = closed_form((256, d), 0.5698402910, 0.1 * d, 1.0).to(cuda)
        with torch.no_grad():
            got = m(x).cpu().numpy()
        assert np.abs(got - g[key]).max() <= TOL, key


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_g3_pipeline_hip(cuda, precision):
    """ray_dirs -> compact ray/voxel pairs -> fused query, compared with the reference's own
    data_dict (voxel-major order) through the ray-major -> reference permutation."""
    from implicit_depth_amd.query import compute_ray_aabb, lidf_query, ray_dirs, to_reference_order
    g = load("g3_pipeline.npz")
    h, w, ray_dir, ray_pix, ray_bid, ray_flat, vb, vbid = g3_inputs(g)
    intr = torch.from_numpy(g["intr"]).to(cuda)
    d = ray_dirs(intr[:, 0], intr[:, 1], intr[:, 2], intr[:, 3], h, w).reshape(-1, 3)
    assert (d.cpu() - ray_dir).abs().max().item() <= 2e-7
    # use the reference's ray directions from here on so that t values are bit-comparable
    rd = ray_dir.to(cuda)
    off, pr, pv, pt = compute_ray_aabb(rd, vb.to(cuda), ray_bid.to(cuda), vbid.to(cuda))
    P = pr.shape[0]
    assert P == g["pair_pred_pos"].shape[0]
    perm = to_reference_order(pr, pv)
    assert (pv[perm].cpu().numpy() == g["occ_vox_intersect_idx"]).all()
    assert (pr[perm].cpu().numpy() == g["miss_ray_intersect_idx"]).all()
    assert np.abs(pt[perm, 0].cpu().numpy() - g["intersect_enter_dist"]).max() == 0.0  # bit-exact
    assert np.abs(pt[perm, 1].cpu().numpy() - g["intersect_leave_dist"]).max() == 0.0
    D = 385
    prob = make_module("IMNET", closed_form_params("IMNET", D, seed=21), D, cuda)
    offd = make_module("IEF", closed_form_params("IEF", D, seed=22), D, cuda)
    depth = torch.zeros((2, h, w), device=cuda)
    with torch.no_grad():
        out = lidf_query(rd, ray_pix.to(cuda), ray_bid.to(cuda), off, pr, pv, pt,
                         torch.from_numpy(g["full_rgb_feat"]).to(cuda),
                         torch.from_numpy(g["occ_voxel_feat"]).to(cuda), prob, offd,
                         offset_range=tuple(float(v) for v in g["offset_range"]),
                         part_size=float(g["part_size"]), ray_flat=ray_flat.to(cuda), depth=depth,
                         precision=precision)
    for k in ("pred_offset", "pred_prob_end", "pair_pred_pos", "pred_prob_end_softmax"):
        got = out[k][perm].cpu().numpy()
        assert np.abs(got - g[k]).max() <= TOL, k
    assert np.abs(out["pred_pos"].cpu().numpy() - g["pred_pos"]).max() <= TOL
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(P, device=cuda)
    gid = out["max_pair_id"]
    gid_ref = torch.where(gid < P, inv[gid.clamp(max=P - 1)], torch.full_like(gid, P)).cpu().numpy()
    sm = g["pred_prob_end_softmax"]
    for r in np.nonzero(gid_ref != g["max_pair_id"])[0]:
        assert abs(sm[gid_ref[r]] - sm[g["max_pair_id"][r]]) <= 1e-6  # only float-noise ties may differ
    dref = g["pred_pos"][:, 2].reshape(2, h, w)
    assert np.abs(depth.cpu().numpy() - dref).mean() <= TOL


def test_g5_decoder_grads_hip(cuda):
    """The library's training path (forward with kept activations + backward) against the
    gradients of the reference's own IMNet / IEF modules."""
    from test_oracle_golden import g5_cases, g5_inputs
    g = load("g5_decoder_grads.npz")
    for key, kind, d, n_iter, sig, seed in g5_cases(g):
        m = make_module(kind, closed_form_params(kind, d, seed=seed), d, cuda, n_iter=n_iter,
                        use_sigmoid=sig).train()
        x, wgt = g5_inputs(d)
        x = x.to(cuda).requires_grad_(True)
        y = m(x)
        (y * wgt.to(cuda)).sum().backward()
        assert np.abs(y.detach().cpu().numpy() - g[key + "_y"]).max() <= TOL, key
        ref = g[key + "_g_input"]
        assert np.abs(x.grad.cpu().numpy() - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max()), key
        for k, v in m.named_parameters():
            ref = g[key + "_g_" + k]
            assert np.abs(v.grad.cpu().numpy() - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max()), (key, k)
