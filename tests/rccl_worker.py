"""Worker of tests/test_rccl_gpu.py, launched by torch.distributed.run (one process per GPU,
backend nccl = RCCL): the depth all-gather of implicit_depth_amd.dist on real device buffers —
equal shards, ragged shards, and the depth map of a small lidf_query."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    share_gpu = os.environ.get("LIDF_TEST_SHARE_GPU") == "1"   # every rank on cuda:0, gloo (1-GPU box)
    if share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if share_gpu:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from implicit_depth_amd.dist import (all_gather_depth, all_gather_depth_ragged, all_gather_depth_rows,
                                         crop_rows, shard_frames, shard_rays, slice_rays)
    from util import orc, run_query

    # equal shards
    loc = torch.full((2, 6, 8), float(rank + 1), device=dev)
    full = all_gather_depth(loc)
    assert full.shape == (2 * world, 6, 8)
    for r in range(world):
        assert (full[2 * r:2 * r + 2] == r + 1).all()
    # ragged shards: 2*world + 1 frames
    n = 2 * world + 1
    lo, hi = shard_frames(n, world, rank)
    loc = torch.arange(lo, hi, device=dev, dtype=torch.float32).view(-1, 1, 1).expand(hi - lo, 3, 4).contiguous()
    full = all_gather_depth_ragged(loc, n)
    assert full.shape == (n, 3, 4) and (full[:, 0, 0] == torch.arange(n, device=dev)).all()
    # the depth map of a query, gathered
    scene = orc.synthetic_scene(1, 12, 16, 8, seed=50 + rank)
    got = run_query(scene, dev)
    full = all_gather_depth(got["depth"])
    assert (full[rank] == got["depth"][0]).all() and torch.isfinite(full).all()
    # ONE frame, image rows sharded over the ranks (fewer frames than GPUs): every rank
    # queries its rows of the same ragged scene; the gathered rows equal the unsharded query's map
    h, w = 13, 16
    whole = orc.synthetic_scene(1, h, w, 8, seed=77, ragged=True)
    lo, hi = shard_rays(h, world, rank)
    part = run_query(dict(slice_rays(whole, lo * w, hi * w), B=1, h=h, w=w), dev)
    rows = all_gather_depth_rows(part["depth"][0, lo:hi], h)
    ref = run_query(whole, dev)
    assert rows.shape == (h, w) and torch.equal(rows, ref["depth"][0])
    # the same with the feature map cut to the rank's rows + the RoIAlign halo (dist.crop_rows): the per-ray
    # features, and so the whole map, are those of the uncut query bit for bit
    mine = slice_rays(whole, lo * w, hi * w)
    cut, fg, r0 = crop_rows(mine, whole["feat_grid"], lo, hi, 4)
    part = run_query(dict(cut, feat_grid=fg, B=1, h=fg.shape[2], w=w), dev)
    rows = all_gather_depth_rows(part["depth"][0, lo - r0:hi - r0], h)
    assert rows.shape == (h, w) and torch.equal(rows, ref["depth"][0])
    dist.barrier()
    if rank == 0:
        print("RCCL_WORKER_OK world=%d" % world, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
