"""GPU: occupied-voxel build (batch_get_occupied_idx + get_occ_vox_bound) — indices and f32
outputs bit-exact against the oracle and the reference trace."""
import numpy as np
import pytest
import torch

from test_oracle_golden import load
from util import orc

pytestmark = pytest.mark.gpu


def check(res, ref):
    assert res["part_size"] == ref["part_size"]
    for k in ("revidx", "valid_v_pid", "occ_vox_bid", "occ_vox_global_coord"):
        assert (res[k].cpu() == ref[k]).all(), k
    for k in ("valid_v_rel_coord", "voxel_bound"):
        assert res[k].shape == ref[k].shape and (res[k].cpu() == ref[k]).all(), k
    assert (res["xmin"].cpu() == ref["xmin"]).all()


def test_voxelize_golden(cuda):
    from implicit_depth_amd.query import get_occ_vox_bound
    g = load("g3_pipeline.npz")
    xyz = torch.from_numpy(g["valid_xyz"])
    bid = torch.from_numpy(g["valid_bid"])
    res = get_occ_vox_bound(xyz.to(cuda), bid.int().to(cuda), batch=2)
    assert (res["voxel_bound"].cpu().numpy() == g["voxel_bound"]).all()
    assert (res["revidx"].cpu().numpy() == g["revidx"]).all()
    assert (res["occ_vox_bid"].cpu().numpy() == g["occ_vox_bid"]).all()
    check(res, orc.occupied_voxels(xyz, bid))


@pytest.mark.parametrize("n,B", [(20000, 3), (1, 1), (0, 2), (5000, 1)])
def test_voxelize_random(cuda, n, B):
    from implicit_depth_amd.query import get_occ_vox_bound
    g = torch.Generator().manual_seed(n + B)
    xyz = (torch.rand(n, 3, generator=g) - 0.5) * 3.0 + torch.tensor([0.0, 0.0, 1.0])  # some outside
    if n > 10:
        xyz[:5] = torch.tensor([-1.125, -1.125, -0.125])          # exactly on the lower grid corner
        xyz[5:8] = torch.tensor([1.125, 0.0, 1.0])                # exactly on the upper face: outside
    bid = torch.randint(0, B, (n,), generator=g)
    ref = orc.occupied_voxels(xyz, bid)
    res = get_occ_vox_bound(xyz.to(cuda), bid.int().to(cuda), batch=B)
    check(res, ref)
