"""GPU: box tests (ray/voxel slab test, point/voxel inside test), scan and per-ray reduction
against the oracle — integer outputs and t values must be BIT-EXACT."""
import numpy as np
import pytest
import torch

from util import orc

pytestmark = pytest.mark.gpu


def scene(R, V, seed, B=2):
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(R, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    if R >= 3:
        d[0], d[1], d[2] = [0, 0, 1], [1, 0, 0], [0, -1, 0]  # zero components (d + 1e-12 path)
    lo = (rng.integers(-4, 4, size=(V, 3)) * 0.25).astype(np.float32)
    vb = np.concatenate([lo, lo + 0.25], 1).astype(np.float32)
    rb = rng.integers(0, B, R).astype(np.int32)
    vbid = np.sort(rng.integers(0, B, V)).astype(np.int32)
    return d, vb, rb, vbid


@pytest.mark.parametrize("R,V", [(1000, 60), (257, 300), (5, 1)])
def test_ray_aabb_dense_and_compact(cuda, R, V):
    from implicit_depth_amd.extensions import ray_aabb
    from implicit_depth_amd.query import compute_ray_aabb
    d, vb, rb, vbid = scene(R, V, R + V)
    m_ref, t_ref = orc.ray_aabb(d, vb, rb, vbid)
    t = [torch.from_numpy(a).to(cuda) for a in (d, vb, rb, vbid)]
    mask, dist = ray_aabb.forward(*t)
    assert mask.dtype == torch.int32 and mask.shape == (V, R) and dist.shape == (V, R, 2)
    assert (mask.cpu().numpy() == m_ref).all()
    assert (dist.cpu().numpy() == t_ref).all()
    off, pr, pv, pt = compute_ray_aabb(*t)
    pr_ref, pv_ref, pt_ref, off_ref = orc.pairs_from_dense(m_ref, t_ref)
    assert (off.cpu().long() == off_ref).all()
    assert (pr.cpu().long() == pr_ref).all() and (pv.cpu().long() == pv_ref).all()
    assert (pt.cpu() == pt_ref).all()


def test_ray_aabb_empty(cuda):
    from implicit_depth_amd.query import compute_ray_aabb
    d, vb, rb, vbid = scene(64, 4, 0)
    t = [torch.from_numpy(a).to(cuda) for a in (d, vb, rb, vbid)]
    off, pr, pv, pt = compute_ray_aabb(t[0][:0], t[1], t[2][:0], t[3])  # no rays
    assert off.tolist() == [0] and pr.numel() == 0
    off, pr, pv, pt = compute_ray_aabb(t[0], t[1][:0], t[2], t[3][:0])  # no voxels
    assert off.shape[0] == 65 and int(off[-1]) == 0 and pt.shape == (0, 2)
    far = torch.tensor([[50.0, 50, 50, 50.25, 50.25, 50.25]], device=cuda)
    off, pr, pv, pt = compute_ray_aabb(t[0], far, torch.zeros(64, dtype=torch.int32, device=cuda),
                                       torch.zeros(1, dtype=torch.int32, device=cuda))
    assert pr.numel() >= 0 and int(off[-1]) == pr.numel()


def grid_scene(R, B, res, fill, seed):
    """Occupied cells of the reference's widened grid (models/pipeline.py:167-173 with
    utils/point_utils.py:40-76: bound_min = xmin + coord * part_size in f32), sorted by
    (frame, x, y, z) like torch.unique sorts them; rays through, beside and along the grid."""
    g = torch.Generator().manual_seed(seed)
    ps = torch.tensor(2.0 / 8, dtype=torch.float32)
    xmin = torch.tensor([-1.0, -1.0, 0.0]) - 0.5 * ps
    key = torch.nonzero(torch.rand(B * res ** 3, generator=g) < fill).squeeze(1)
    bid = key // res ** 3
    coord = torch.stack(((key // res ** 2) % res, (key // res) % res, key % res), 1)
    lo = xmin + coord.float() * ps
    vb = torch.cat((lo, lo + ps), 1).contiguous()
    d = torch.randn(R, 3, generator=g)
    d = d / d.norm(dim=1, keepdim=True)
    if R >= 8:
        d[0], d[1], d[2] = torch.tensor([0.0, 0, 1]), torch.tensor([1.0, 0, 0]), torch.tensor([0.0, -1, 0])
        d[3] = torch.tensor([0.0, 0.6, 0.8])           # in a cell-boundary plane x = const? (x = 0)
        d[4] = torch.tensor([-0.125, -0.125, 0.125]) / 0.2165063509   # through cell corners
        d[5] = torch.tensor([0.6, 0.0, -0.8])          # pointing away from the grid (no t >= 0 test)
        d[6] = -d[7]                                    # a line and its reverse hit the same voxels
    rb = torch.randint(0, B, (R,), generator=g).int()
    return d.contiguous(), vb, rb, bid.int().contiguous(), coord.int().contiguous()


@pytest.mark.parametrize("R,B,fill", [(3000, 1, 0.15), (2000, 3, 0.6), (900, 2, 1.0), (700, 2, 0.01)])
def test_ray_aabb_grid_walk(cuda, R, B, fill):
    """lidf_ray_aabb_grid_*: bit-identical to the voxel-by-voxel compact path and to the oracle."""
    from implicit_depth_amd.query import compute_ray_aabb
    res = 9
    d, vb, rb, vbid, coord = grid_scene(R, B, res, fill, seed=R + B)
    t = [a.to(cuda) for a in (d, vb, rb, vbid)]
    ref = compute_ray_aabb(*t)
    got = compute_ray_aabb(*t, voxel_coord=coord.to(cuda), grid_dims=(res, res, res), batch=B)
    for a, b in zip(got, ref):
        assert a.dtype == b.dtype and torch.equal(a, b)
    # int64 coordinates (data_dict['occ_vox_global_coord']) are narrowed
    got = compute_ray_aabb(*t, voxel_coord=coord.long().to(cuda), grid_dims=(res, res, res), batch=B)
    assert torch.equal(got[2], ref[2]) and torch.equal(got[3], ref[3])
    if vb.shape[0] <= 600:
        m_ref, t_ref = orc.ray_aabb(d.numpy(), vb.numpy(), rb.numpy(), vbid.numpy())
        pr_ref, pv_ref, pt_ref, off_ref = orc.pairs_from_dense(m_ref, t_ref)
        assert (got[0].cpu().long() == off_ref).all() and (got[2].cpu().long() == pv_ref).all()
        assert (got[3].cpu() == pt_ref).all()


def test_ray_aabb_grid_walk_edges(cuda):
    from implicit_depth_amd.query import compute_ray_aabb
    d, vb, rb, vbid, coord = grid_scene(64, 2, 9, 0.3, seed=5)
    t = [a.to(cuda) for a in (d, vb, rb, vbid)]
    c = coord.to(cuda)
    kw = dict(grid_dims=(9, 9, 9), batch=2)
    off, pr, pv, pt = compute_ray_aabb(t[0][:0], t[1], t[2][:0], t[3], voxel_coord=c, **kw)   # no rays
    assert off.tolist() == [0] and pr.numel() == 0
    off, pr, pv, pt = compute_ray_aabb(t[0], t[1][:0], t[2], t[3][:0], voxel_coord=c[:0], **kw)  # no voxels
    assert off.shape[0] == 65 and int(off[-1]) == 0 and pt.shape == (0, 2)
    # a non-cubic grid and rays of a frame that has no voxel at all
    g = torch.Generator().manual_seed(9)
    key = torch.nonzero(torch.rand(5 * 7 * 11, generator=g) < 0.4).squeeze(1)
    coord = torch.stack((key // 77, (key // 11) % 7, key % 11), 1)
    lo = torch.tensor([-0.6, -0.9, 0.1]) + coord.float() * torch.tensor(0.2)
    vb = torch.cat((lo, lo + torch.tensor(0.2)), 1).contiguous().to(cuda)
    vbid = torch.ones(key.numel(), dtype=torch.int32, device=cuda)           # every voxel in frame 1
    rb = (torch.arange(64) % 3).int().to(cuda)                               # frames 0, 1, 2
    ref = compute_ray_aabb(t[0], vb, rb, vbid)
    got = compute_ray_aabb(t[0], vb, rb, vbid, voxel_coord=coord.int().to(cuda), grid_dims=(5, 7, 11), batch=3)
    for a, b in zip(got, ref):
        assert torch.equal(a, b)
    assert int(ref[0][-1]) > 0
    with pytest.raises(RuntimeError):
        compute_ray_aabb(t[0], vb, rb, vbid, voxel_coord=coord.int().to(cuda))   # dims missing


@pytest.mark.parametrize("N,V", [(900, 70), (3, 300)])
def test_pcl_aabb(cuda, N, V):
    from implicit_depth_amd.extensions import pcl_aabb
    d, vb, rb, vbid = scene(N, V, 7 * N + V)
    rng = np.random.default_rng(5)
    pts = rng.uniform(-1.1, 1.1, size=(N, 3)).astype(np.float32)
    k = min(N, V)
    pts[:k] = vb[:k, :3]  # on a corner: inclusive bounds, several voxels may contain it
    ref = orc.pcl_aabb(pts, vb, rb, vbid)
    t = [torch.from_numpy(a).to(cuda) for a in (pts, vb, rb, vbid)]
    mask = pcl_aabb.forward(*t)
    assert (mask.cpu().numpy() == ref).all()
    last = pcl_aabb.last_voxel(*t).cpu().numpy()
    exp = np.where(ref.any(0), V - 1 - np.argmax(ref[::-1], axis=0), -1)
    assert (last == exp).all()


@pytest.mark.parametrize("n", [0, 1, 1023, 1024, 1025, 300000])
def test_exclusive_scan(cuda, n):
    from implicit_depth_amd import _lib
    g = torch.Generator().manual_seed(n)
    x = torch.randint(0, 30, (n,), generator=g, dtype=torch.int32)
    xd = x.to(cuda)
    out = torch.empty((n + 1,), dtype=torch.int32, device=cuda)
    L = _lib.lib()
    wsb = L.lidf_exclusive_scan_workspace_bytes(n)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=cuda)
    _lib.check(L.lidf_exclusive_scan_i32(_lib.ptr(xd), n, _lib.ptr(out), _lib.ptr(ws), wsb,
                                         _lib.current_stream(cuda)))
    ref = torch.zeros(n + 1, dtype=torch.int64)
    ref[1:] = torch.cumsum(x.long(), 0)
    assert (out.cpu().long() == ref).all()


def test_ray_reduce_ties_and_empties(cuda):
    from implicit_depth_amd import _lib
    # ray 0: 3 pairs with a tie between pair 1 and 2 -> lowest index (1); ray 1: empty;
    # ray 2: 70 pairs (more than one wavefront pass); ray 3: single pair
    cnt = [3, 0, 70, 1]
    off = torch.tensor([0, 3, 3, 73, 74], dtype=torch.int32)
    g = torch.Generator().manual_seed(0)
    prob = torch.randn(74, generator=g)
    prob[1] = prob[2] = 5.0
    prob[3 + 69] = 9.0
    pos = torch.randn(74, 3, generator=g)
    ray = torch.repeat_interleave(torch.arange(4), torch.tensor(cnt))
    sm_ref = orc.scatter_softmax(prob, ray, 4)
    _, id_ref = orc.scatter_max(sm_ref, ray, 4)
    sm = torch.empty(74, device=cuda)
    mid = torch.empty(4, dtype=torch.int64, device=cuda)
    pp = torch.empty(4, 3, device=cuda)
    depth = torch.full((1, 2, 2), -1.0, device=cuda)
    bid = torch.zeros(4, dtype=torch.int32, device=cuda)
    flat = torch.arange(4, dtype=torch.int32, device=cuda)
    t = [a.to(cuda) for a in (prob, pos, off)]
    _lib.check(_lib.lib().lidf_ray_reduce_f32(_lib.ptr(t[0]), _lib.ptr(t[1]), _lib.ptr(t[2]), 4, 74,
                                              _lib.ptr(bid), _lib.ptr(flat), 4, _lib.ptr(sm),
                                              _lib.ptr(mid), _lib.ptr(pp), _lib.ptr(depth),
                                              _lib.current_stream(cuda)))
    assert mid.cpu().tolist() == id_ref.tolist() == [1, 74, 72, 73]
    assert (sm.cpu() - sm_ref).abs().max().item() <= 1e-6
    exp = torch.cat((pos, torch.zeros(1, 3)))[id_ref]
    assert (pp.cpu() == exp).all()
    assert (depth.cpu().flatten() == exp[:, 2]).all()
