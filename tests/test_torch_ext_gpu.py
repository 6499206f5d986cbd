"""The pybind11 torch-extension shim (csrc/lidf_torch_ext.cpp) against the ctypes binding: the same
C ABI underneath, so results are bitwise equal; plus its argument checks (TORCH_CHECK ->
RuntimeError, as the reference's CHECK_INPUT)."""
import pytest
import torch

from util import make_module, orc, to_dev

pytestmark = pytest.mark.gpu


def test_boxes_and_pairs(cuda):
    from implicit_depth_amd import torch_ext
    from implicit_depth_amd.extensions import pcl_aabb, ray_aabb
    from implicit_depth_amd.query import compute_ray_aabb
    m = torch_ext.ext()
    scene = orc.synthetic_scene(2, 12, 16, 4, seed=5)
    s = to_dev(scene, cuda)
    vb = torch.cat((s["vox_center"] - 0.125, s["vox_center"] + 0.125), 1).contiguous()
    vbid = torch.arange(2, device=cuda).repeat_interleave(729).int()
    mask, dist = m.ray_aabb(s["ray_dir"], vb, s["ray_bid"], vbid)
    rm, rd = ray_aabb.forward(s["ray_dir"], vb, s["ray_bid"], vbid)
    assert (mask == rm).all() and (dist == rd).all()
    pts = (s["ray_dir"] * 0.9).contiguous()
    assert (m.pcl_aabb(pts, vb, s["ray_bid"], vbid) == pcl_aabb.forward(pts, vb, s["ray_bid"], vbid)).all()
    a = m.compute_ray_aabb(s["ray_dir"], vb, s["ray_bid"], vbid)
    b = compute_ray_aabb(s["ray_dir"], vb, s["ray_bid"], vbid)
    assert all((x == y).all() for x, y in zip(a, b)) and a[1].shape[0] > 0
    with pytest.raises(RuntimeError):
        m.ray_aabb(s["ray_dir"].cpu(), vb, s["ray_bid"], vbid)
    with pytest.raises(RuntimeError):
        m.ray_aabb(s["ray_dir"], vb, s["ray_bid"].long(), vbid)


@pytest.mark.parametrize("precision", [0, 1])
def test_decoders_and_query(cuda, precision):
    from implicit_depth_amd import decoders_forward, torch_ext
    from implicit_depth_amd.query import lidf_query
    m = torch_ext.ext()
    scene = orc.synthetic_scene(2, 12, 16, 6, seed=6, ragged=True)
    s = to_dev(scene, cuda)
    D = scene["D"]
    prob, off = make_module("IMNET", scene["prob_p"], D, cuda), make_module("IEF", scene["off_p"], D, cuda)
    pw, ow = torch_ext.decoder_weights(prob), torch_ext.decoder_weights(off)
    name = ("f32", "f16x3")[precision]
    x = torch.randn(300, D, device=cuda)
    gp, go = m.forward_decoders(x, pw, ow, 2, 0.001, False, precision)
    with torch.no_grad():
        rp, ro = decoders_forward(x, prob, off, precision=name)
    assert (gp == rp).all() and (go == ro).all()
    depth = torch.zeros(2, 12, 16, device=cuda)
    out = m.forward_query(s["ray_dir"], s["ray_pix"].long(), s["ray_bid"].long(), s["ray_flat"].long(),
                          s["pair_off"], s["pair_ray"], s["pair_vox"], s["pair_t"], s["feat_grid"],
                          s["vox_feat"], None, pw, ow, 2, 0.001, False, 8, 4, 8, False, 0.0, 1.0, 0.25,
                          depth, precision)
    d2 = torch.zeros(2, 12, 16, device=cuda)
    with torch.no_grad():
        ref = lidf_query(s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"],
                         s["pair_vox"], s["pair_t"], s["feat_grid"], s["vox_feat"], prob, off,
                         ray_flat=s["ray_flat"], depth=d2, precision=name)
    keys = ("pred_offset", "pred_prob_end", "pair_pred_pos", "pred_prob_end_softmax", "max_pair_id", "pred_pos")
    for got, k in zip(out, keys):
        assert (got == ref[k]).all(), k
    assert (depth == d2).all()
    with pytest.raises(RuntimeError):   # depth without ray_flat
        m.forward_query(s["ray_dir"], s["ray_pix"], s["ray_bid"], None, s["pair_off"], s["pair_ray"],
                        s["pair_vox"], s["pair_t"], s["feat_grid"], s["vox_feat"], None, pw, ow, 2, 0.001,
                        False, 8, 4, 8, False, 0.0, 1.0, 0.25, depth, precision)
