"""ray_aabb.forward — dense drop-in for the reference's ray/voxel slab-test extension
(extensions/ray_aabb/ray_aabb_cuda.cpp:20-37, kernel ray_aabb_cuda_kernel.cu:10-126).

forward(ray_dir [R,3] f32, voxel_bound [V,6] f32, ray_bid [R] i32, voxel_bid [V] i32)
  -> (mask [V,R] i32, dist [V,R,2] f32), zero where the ray misses the voxel.
The compact ray-major form the fused query consumes is implicit_depth_amd.query.compute_ray_aabb.
"""
import torch

from .. import _lib


def forward(ray_dir, voxel_bound, ray_bid, voxel_bid):
    _lib.require_cuda(ray_dir, voxel_bound, ray_bid, voxel_bid,
                      names=["ray_dir", "voxel_bound", "ray_bid", "voxel_bid"])
    if ray_dir.dtype != torch.float32 or voxel_bound.dtype != torch.float32:
        raise RuntimeError("ray_dir and voxel_bound must be float32")
    if ray_bid.dtype != torch.int32 or voxel_bid.dtype != torch.int32:
        raise RuntimeError("ray_bid and voxel_bid must be int32 (the reference calls .int())")
    R, V = ray_dir.shape[0], voxel_bound.shape[0]
    dev = ray_dir.device
    mask = torch.zeros((V, R), dtype=torch.int32, device=dev)
    dist = torch.zeros((V, R, 2), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().lidf_ray_aabb_dense_f32(
            _lib.ptr(ray_dir), _lib.ptr(voxel_bound), _lib.ptr(ray_bid), _lib.ptr(voxel_bid), R, V,
            _lib.ptr(mask), _lib.ptr(dist), _lib.current_stream(dev)))
    return mask, dist
