"""CPU: properties of the two restatements whose third-party sources are absent (roi_align,
torch_scatter: parity unpinned), and the C box-test oracle against the numpy one."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import torch

from util import ROOT, orc


def test_roi_align_constant_and_block_means():
    feat = torch.full((1, 3, 12, 16), 2.5)
    pix = torch.tensor([[8, 6], [0, 0], [15, 11], [3, 10]])
    boxes = orc.roi_boxes(pix, torch.zeros(4, dtype=torch.long), 12, 16, 8)
    out = orc.roi_align(feat, boxes)
    assert torch.allclose(out, torch.full_like(out, 2.5))
    g = torch.Generator().manual_seed(0)
    feat = torch.randn(1, 2, 12, 16, generator=g)
    out = orc.roi_align(feat, boxes[:1])  # interior pixel (8,6): box x 4..12, y 2..10
    for ph in range(2):
        for pw in range(2):
            blk = feat[0, :, 2 + 4 * ph: 6 + 4 * ph, 4 + 4 * pw: 8 + 4 * pw].mean((1, 2))
            assert torch.allclose(out[0, :, ph, pw], blk, atol=1e-6)


def test_roi_align_fast_matches_loop():
    g = torch.Generator().manual_seed(1)
    feat = torch.randn(2, 4, 10, 14, generator=g)
    ys, xs = torch.meshgrid(torch.arange(10), torch.arange(14), indexing="ij")
    pix = torch.stack((xs.flatten(), ys.flatten()), 1).repeat(2, 1)
    bid = torch.arange(2).repeat_interleave(140)
    for bbox in (8, 7, 5, 2):
        boxes = orc.roi_boxes(pix, bid, 10, 14, bbox)
        assert (orc.roi_align(feat, boxes) - orc.roi_align_fast(feat, boxes)).abs().max() <= 2e-6


def test_scatter_softmax_and_max():
    g = torch.Generator().manual_seed(2)
    src = torch.randn(200, generator=g)
    idx = torch.randint(0, 17, (200,), generator=g)
    sm = orc.scatter_softmax(src, idx, 20)
    sums = torch.zeros(20).index_add_(0, idx, sm)
    present = torch.bincount(idx, minlength=20) > 0
    assert torch.allclose(sums[present], torch.ones(int(present.sum())), atol=1e-6)
    mx, arg = orc.scatter_max(sm, idx, 20)
    for r in range(20):
        sel = (idx == r).nonzero().flatten()
        if sel.numel() == 0:
            assert arg[r] == 200
        else:
            assert arg[r] == sel[torch.argmax(src[sel])]  # argmax(softmax) == argmax(logit)
    # ties: first index wins
    _, a = orc.scatter_max(torch.tensor([1.0, 3.0, 3.0, 2.0]), torch.tensor([0, 0, 0, 0]), 1)
    assert a[0] == 1


def _c_oracle():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    return C.CDLL(os.path.join(ROOT, "oracle", "libaabb_ref.so"))


def test_c_box_oracle_matches_numpy():
    L = _c_oracle()
    P = C.c_void_p
    rng = np.random.default_rng(0)
    R, V = 700, 50
    d = rng.normal(size=(R, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[0], d[1], d[2] = [0, 0, 1], [1, 0, 0], [0, -1, 0]  # axis-parallel rays: d == 0 components
    lo = (rng.integers(-4, 4, size=(V, 3)) * 0.25).astype(np.float32)
    vb = np.concatenate([lo, lo + 0.25], 1).astype(np.float32)
    rb = rng.integers(0, 2, R).astype(np.int32)
    vbid = np.sort(rng.integers(0, 2, V)).astype(np.int32)
    m = np.zeros((V, R), np.int32)
    dist = np.zeros((V, R, 2), np.float32)
    L.ray_aabb_ref(d.ctypes.data_as(P), vb.ctypes.data_as(P), rb.ctypes.data_as(P),
                   vbid.ctypes.data_as(P), C.c_int64(R), C.c_int64(V), m.ctypes.data_as(P),
                   dist.ctypes.data_as(P))
    m2, d2 = orc.ray_aabb(d, vb, rb, vbid)
    assert (m == m2).all() and (dist == d2).all() and m.sum() > 0
    pts = (rng.uniform(-1, 1, size=(R, 3))).astype(np.float32)
    pts[:10] = vb[:10, :3]  # points on a voxel corner / shared face: inclusive bounds
    pm = np.zeros((V, R), np.int32)
    L.pcl_aabb_ref(pts.ctypes.data_as(P), vb.ctypes.data_as(P), rb.ctypes.data_as(P),
                   vbid.ctypes.data_as(P), C.c_int64(R), C.c_int64(V), pm.ctypes.data_as(P))
    assert (pm == orc.pcl_aabb(pts, vb, rb, vbid)).all() and pm.sum() > 0


def test_depth_metrics_oracle_known_values():
    """Hand-checkable cases of the eval statistics (pipeline.py:577-627): a perfect prediction, a
    uniform +10 % error, and the nearest-neighbour source indices of the 320 -> 256 resize."""
    gt = torch.full((240, 320), 2.0)
    m = orc.depth_metrics(gt.clone(), gt, None)
    assert float(m["a1"]) == 1.0 and float(m["rmse"]) == 0.0 and float(m["count"]) == 144 * 256
    m = orc.depth_metrics(gt * 1.1, gt, None)
    assert float(m["a1"]) == 0.0 and float(m["a3"]) == 1.0        # ratio 1.1: above 1.05 (and 1.10 in f32)
    assert abs(float(m["abs_rel"]) - 0.1) < 1e-6 and abs(float(m["mae"]) - 0.2) < 1e-6
    assert abs(float(m["rmse_log"]) - math.log(1.1)) < 1e-6
    # columns: pred = source column index -> resized row 0 must read columns floor(1.25 x)
    pred = torch.arange(320, dtype=torch.float32).repeat(240, 1) + 1.0
    gt = pred.clone()
    gt[:, 5:] = 0.0                                                # only source columns 0..4 valid
    m = orc.depth_metrics(pred, gt, None)                          # dst x = 0..3 -> src 0,1,2,3 ; x = 4 -> src 5
    assert float(m["count"]) == 4 * 144
