"""Eval depth metrics on the device (models/pipeline.py:577-627) against the oracle."""
import math

import pytest
import torch

from util import orc

pytestmark = pytest.mark.gpu


def _maps(h, w, seed, holes=True):
    g = torch.Generator().manual_seed(seed)
    gt = 0.4 + 1.2 * torch.rand(h, w, generator=g)
    pred = gt * (1.0 + 0.08 * torch.randn(h, w, generator=g))
    seg = torch.rand(h, w, generator=g) < 0.4
    if holes:
        gt[torch.rand(h, w, generator=g) < 0.1] = 0.0
        gt[3, 5] = float("nan")
        gt[7, 9] = float("inf")
        gt[11, 2] = -0.5
    return pred, gt, seg


@pytest.mark.parametrize("h,w,out_size", [(240, 320, (144, 256)), (144, 256, (144, 256)),
                                          (480, 640, (144, 256)), (37, 53, None), (240, 320, (100, 77))])
def test_metrics_match_oracle(cuda, h, w, out_size):
    from implicit_depth_amd.query import METRIC_NAMES, depth_metrics
    pred, gt, seg = _maps(h, w, seed=h + w)
    ref = orc.depth_metrics(pred, gt, seg, out_size)
    got = depth_metrics(pred.to(cuda), gt.to(cuda), seg.to(cuda), out_size)
    assert float(got["count"]) == float(ref["count"])          # the same pixels are selected
    for k in METRIC_NAMES:
        r, v = float(ref[k]), float(got[k])
        assert abs(v - r) <= 2e-6 * max(1.0, abs(r)), (k, v, r)


def test_metrics_no_mask_and_empty(cuda):
    from implicit_depth_amd.query import METRIC_NAMES, depth_metrics
    pred, gt, _ = _maps(60, 80, seed=3)
    ref = orc.depth_metrics(pred, gt, None, (30, 40))
    got = depth_metrics(pred.to(cuda), gt.to(cuda), None, (30, 40))
    for k in METRIC_NAMES:
        assert abs(float(got[k]) - float(ref[k])) <= 2e-6 * max(1.0, abs(float(ref[k]))), k
    got = depth_metrics(pred.to(cuda), torch.zeros_like(gt).to(cuda), None, (30, 40))
    assert float(got["count"]) == 0 and all(math.isnan(float(got[k])) for k in METRIC_NAMES)


def test_metrics_argument_errors(cuda):
    from implicit_depth_amd.query import depth_metrics
    a = torch.ones(4, 4)
    with pytest.raises(RuntimeError):
        depth_metrics(a, a)                                    # CPU tensors
    with pytest.raises(RuntimeError):
        depth_metrics(a.to(cuda), torch.ones(4, 5, device=cuda))


def test_metrics_match_reference_compute_loss(cuda):
    """lidf_depth_metrics_f32 against the reference's own compute_loss statistics
    (tests/golden/g8_metrics.npz, models/pipeline.py:577-618)."""
    from implicit_depth_amd.query import depth_metrics
    from test_oracle_golden import G8_KEYS, load
    g8 = load("g8_metrics.npz")
    got = depth_metrics(torch.from_numpy(g8["pred_depth"]).to(cuda), torch.from_numpy(g8["gt_depth"]).to(cuda),
                        torch.from_numpy(g8["seg_mask"]).to(cuda), tuple(int(v) for v in g8["out_size"]))
    for k in G8_KEYS:
        assert abs(float(got[k]) - float(g8[k])) <= 2e-6 * max(1.0, abs(float(g8[k]))), k
