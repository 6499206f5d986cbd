"""GPU tests of get_miss_ray: mask -> compacted rays, vs the reference's own
LIDF.get_miss_ray outputs (tests/golden/g6_miss_ray.npz, g3_pipeline.npz) and vs the oracle on
random masks of every accepted element type."""
import os

import numpy as np
import pytest
import torch

from util import orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _check(res, ref, R):
    assert res["total_miss_sample_num"] == R
    for k in ("miss_bid", "miss_flat_img_id", "miss_img_ind"):
        assert res[k].dtype == torch.int64
        assert (res[k].cpu().numpy() == np.asarray(ref[k])).all(), k
    assert (res["ray_bid"].cpu().long().numpy() == np.asarray(ref["miss_bid"])).all()
    assert (res["ray_flat"].cpu().long().numpy() == np.asarray(ref["miss_flat_img_id"])).all()
    assert (res["ray_pix"].cpu().long().numpy() == np.asarray(ref["miss_img_ind"])).all()
    if R:
        assert np.abs(res["miss_ray_dir"].cpu().numpy() - np.asarray(ref["miss_ray_dir"])).max() <= 2e-7


def test_g6_reference_masks(cuda):
    from implicit_depth_amd.query import get_miss_ray
    g = np.load(os.path.join(GOLD, "g6_miss_ray.npz"))
    for key in ("a", "b", "c"):
        mask = torch.from_numpy(g[key + "_mask"]).to(cuda)
        intr = torch.from_numpy(g[key + "_intr"]).to(cuda)
        res = get_miss_ray(mask, intr[:, 0], intr[:, 1], intr[:, 2], intr[:, 3])
        ref = {k: g[key + "_" + k] for k in ("miss_bid", "miss_flat_img_id", "miss_ray_dir", "miss_img_ind")}
        _check(res, ref, ref["miss_bid"].shape[0])


def test_g3_all_pixels(cuda):
    """mask_type 'all' (pred_mask = ones, pipeline.py:131): every pixel, the g3 trace's rays."""
    from implicit_depth_amd.query import get_miss_ray
    g = np.load(os.path.join(GOLD, "g3_pipeline.npz"))
    h, w = [int(v) for v in g["hw"]]
    intr = torch.from_numpy(g["intr"]).to(cuda)
    res = get_miss_ray(torch.ones((2, 1, h, w), device=cuda), intr[:, 0], intr[:, 1], intr[:, 2], intr[:, 3])
    _check(res, {k: g[k] for k in ("miss_bid", "miss_flat_img_id", "miss_ray_dir", "miss_img_ind")}, 2 * h * w)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bool, torch.uint8, torch.int32, torch.int64])
@pytest.mark.parametrize("shape", [(1, 7, 9), (4, 240, 320), (2, 3, 1025)])
def test_random_masks_vs_oracle(cuda, dtype, shape):
    from implicit_depth_amd.query import get_miss_ray
    B, h, w = shape
    g = torch.Generator().manual_seed(B * 1000 + h)
    mask = (torch.rand(shape, generator=g) < 0.4)
    if dtype == torch.float32:
        mask = mask.float() * (torch.rand(shape, generator=g) - 0.5)
        mask.view(-1)[3] = float("nan")    # NaN is non-zero for torch.nonzero
    else:
        mask = mask.to(dtype)
    fx = torch.full((B,), 0.9 * w) + torch.arange(B)
    fy = fx * 1.02
    cx = torch.full((B,), w / 2 - 0.5)
    cy = torch.full((B,), h / 2 - 0.25)
    ref = orc.get_miss_ray(mask, fx, fy, cx, cy)
    res = get_miss_ray(mask.to(cuda), fx.to(cuda), fy.to(cuda), cx.to(cuda), cy.to(cuda))
    _check(res, {k: v.numpy() for k, v in ref.items() if torch.is_tensor(v)}, ref["total_miss_sample_num"])


def test_empty_mask_and_feeds_query(cuda):
    """No miss ray (pipeline.py:676 early exit) and the int64 outputs accepted by lidf_query."""
    from implicit_depth_amd.query import get_miss_ray
    z = torch.zeros((2, 6, 8), device=cuda)
    one = torch.ones((2,), device=cuda)
    res = get_miss_ray(z, one * 8, one * 8, one * 3.5, one * 2.5)
    assert res["total_miss_sample_num"] == 0 and res["miss_ray_dir"].shape == (0, 3)
    with pytest.raises(RuntimeError):
        get_miss_ray(z.cpu(), one, one, one, one)
    with pytest.raises(RuntimeError):
        get_miss_ray(z.half(), one, one, one, one)


def test_train_window_matches_reference(cuda):
    """sample_miss_rays on the device output = the reference's train-time window (same seed, same
    np.random.choice calls): tests/golden/g6_miss_ray.npz 't_*'."""
    from implicit_depth_amd.query import get_miss_ray, sample_miss_rays
    g = np.load(os.path.join(GOLD, "g6_miss_ray.npz"))
    mask = torch.from_numpy(g["t_mask"]).to(cuda)
    intr = torch.from_numpy(g["t_intr"]).to(cuda)
    res = get_miss_ray(mask, intr[:, 0], intr[:, 1], intr[:, 2], intr[:, 3])
    np.random.seed(int(g["t_seed"]))
    sub = sample_miss_rays(res, mask.shape[0], int(g["t_miss_sample_num"]))
    ref = {k: g["t_" + k] for k in ("miss_bid", "miss_flat_img_id", "miss_ray_dir", "miss_img_ind")}
    _check(sub, ref, ref["miss_bid"].shape[0])
    assert sample_miss_rays(res, mask.shape[0], -1) is res          # miss_sample_num == -1: all rays
