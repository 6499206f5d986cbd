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


def shard_rays(height, world_size, rank):
    """Row range [lo, hi) of ONE image owned by `rank` when there are fewer frames than ranks
    ("for B < 8 shard rows of rays of one image instead"): whole image rows, contiguous
    and balanced like shard_frames, so a rank's rays are pixels [lo*w, hi*w) — a contiguous slice of
    the all-pixel ray list (and of its ray-major candidate list). The feature map, the voxel features
    and the weights are replicated (9.8 MB + 0.4 MB + 1.1 MB)."""
    return shard_frames(height, world_size, rank)


def slice_rays(rays, lo, hi):
    """The rays [lo, hi) of a ray-major query input set with their candidate list re-based: `rays` is
    a dict with ray_dir / ray_pix / ray_bid / ray_flat [R,...] and the CSR pairs pair_off [R+1],
    pair_ray / pair_vox [P], pair_t [P,2] (host or device tensors; plumbing only: views and two
    subtractions). pair_ray of the slice counts from 0, as lidf_query expects."""
    off = rays["pair_off"]
    p0, p1 = int(off[lo]), int(off[hi])
    out = dict(rays)
    for k in ("ray_dir", "ray_pix", "ray_bid", "ray_flat"):
        if rays.get(k) is not None:
            out[k] = rays[k][lo:hi].contiguous()
    out["pair_off"] = (off[lo:hi + 1] - off[lo]).contiguous()
    out["pair_ray"] = (rays["pair_ray"][p0:p1] - lo).contiguous()
    out["pair_vox"] = rays["pair_vox"][p0:p1].contiguous()
    out["pair_t"] = rays["pair_t"][p0:p1].contiguous()
    out["R"], out["P"] = hi - lo, p1 - p0
    return out


def crop_rows(rays, feat_grid, lo, hi, halo):
    """Row shard [lo, hi) of ONE image without the rows it never reads: the feature map cut to rows
    [lo - halo, hi + halo) (clamped to the image), the rays' pixel rows and flat pixel indices re-based to
    the cut. halo >= roi_inp_bbox // 2: a ray's RoIAlign box is pixel +- roi_inp_bbox // 2, clamped on the
    IMAGE (models/pipeline.py:374-380) — with that halo a box reaches the cut's first / last row only where
    it is the image's own, so the per-ray features are those of the whole map, bit for bit, while the
    box-sum image and the depth map of a rank cover (hi - lo + 2 halo) rows instead of all of them (at 8 GPUs
    38 of 240: the part of a row-sharded step that did not shrink with the shard before).
    `rays`: slice_rays' output (ray_pix [R,2] (x, y), ray_flat [R]); returns (rays', feat_grid', row0):
    row r of the cut is image row row0 + r; the rank's own rows are [lo - row0, hi - row0) of its local map."""
    h, w = feat_grid.shape[2], feat_grid.shape[3]
    r0, r1 = max(0, lo - halo), min(h, hi + halo)
    out = dict(rays)
    pix = rays["ray_pix"].clone()
    pix[:, 1] -= r0
    out["ray_pix"] = pix
    out["ray_flat"] = rays["ray_flat"] - r0 * w
    return out, feat_grid[:, :, r0:r1].contiguous(), r0


def all_gather_depth_rows(local_rows, height, group=None):
    """All-gather of the row shards of one depth map: rank r holds rows shard_rays(height, world, r)
    of a [height, w] map ([rows_r, w] f32); returns the whole [height, w] map on every rank. Shards
    differ by at most one row, so every rank pads to the largest shard and one
    all_gather_into_tensor moves the map (307 KB at 240x320: latency-bound on xGMI)."""
    if not (dist.is_available() and dist.is_initialized()):
        return local_rows
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    spans = [shard_rays(height, world, r) for r in range(world)]
    rmax = max(hi - lo for lo, hi in spans)
    lo, hi = spans[rank]
    if tuple(local_rows.shape[:1]) != (hi - lo,):
        raise ValueError("rank %d owns rows [%d,%d) but holds %d" % (rank, lo, hi, local_rows.shape[0]))
    w = local_rows.shape[1]
    pad = torch.zeros((rmax, w), dtype=local_rows.dtype, device=local_rows.device)
    pad[: hi - lo] = local_rows
    full = torch.empty((world * rmax, w), dtype=local_rows.dtype, device=local_rows.device)
    dist.all_gather_into_tensor(full, pad, group=group)
    if all(b - a == rmax for a, b in spans):
        return full
    return torch.cat([full[r * rmax: r * rmax + (spans[r][1] - spans[r][0])] for r in range(world)], 0)
