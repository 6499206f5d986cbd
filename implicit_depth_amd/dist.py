"""dist.py — multi-GPU sharding of the LIDF query (one process per GPU, torch.distributed).

The query is embarrassingly parallel over frames (every gather is within one image:
ray_bid == voxel_bid filter, extensions/ray_aabb/ray_aabb_cuda_kernel.cu:26), so frames are
sharded across ranks exactly as the reference's DDP splits its batch (trainers/train_lidf.py:163)
and the ONLY collective on the path is the all-gather of the per-rank depth maps (new
functionality: the reference refuses multi-GPU evaluation, trainers/train_lidf.py:693-694).
Backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests of the sharding logic.
"""
import torch
import torch.distributed as dist


def shard_frames(n_frames, world_size, rank):
    """Contiguous, balanced frame range [lo, hi) of `rank`: the first n_frames % world_size ranks
    take one extra frame."""
    if world_size <= 0 or not (0 <= rank < world_size) or n_frames < 0:
        raise ValueError("bad shard arguments")
    per, rem = divmod(n_frames, world_size)
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


def all_gather_depth(local_depth, out=None, group=None):
    """All-gather equally sized per-rank depth maps [B_local,h,w] into [world*B_local,h,w]
    (rank-major = global frame order under shard_frames with equal shards). One
    all_gather_into_tensor: 307 KB per frame, latency-bound on xGMI."""
    if not (dist.is_available() and dist.is_initialized()):
        return local_depth if out is None else out.copy_(local_depth)
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world * local_depth.shape[0],) + tuple(local_depth.shape[1:]),
                          dtype=local_depth.dtype, device=local_depth.device)
    dist.all_gather_into_tensor(out, local_depth.contiguous(), group=group)
    return out


def all_gather_depth_ragged(local_depth, n_frames, group=None):
    """Unequal shards (n_frames not divisible by world): pad to the largest shard, gather, strip."""
    if not (dist.is_available() and dist.is_initialized()):
        return local_depth
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [shard_frames(n_frames, world, r) for r in range(world)]
    bmax = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((bmax,) + tuple(local_depth.shape[1:]), dtype=local_depth.dtype,
                      device=local_depth.device)
    lo, hi = sizes[rank]
    pad[: hi - lo] = local_depth
    full = all_gather_depth(pad, group=group)
    parts = [full[r * bmax: r * bmax + (sizes[r][1] - sizes[r][0])] for r in range(world)]
    return torch.cat(parts, 0)
