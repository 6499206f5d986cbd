"""pcl_aabb.forward — dense drop-in for the reference's point/voxel inside-test extension
(extensions/pcl_aabb/pcl_aabb_cuda.cpp:20-37, kernel pcl_aabb_cuda_kernel.cu:10-80).

forward(pcl_pos [N,3] f32, voxel_bound [V,6] f32, pcl_bid [N] i32, voxel_bid [V] i32)
  -> mask [V,N] i32 (inclusive bounds: a point on a shared face belongs to both voxels).
`last_voxel` is the compact form stage 2 needs (models/pipeline.py:939-944): per point the largest
index of a containing voxel, -1 if none.
"""
import torch

from .. import _lib


def _check(pcl_pos, voxel_bound, pcl_bid, voxel_bid):
    _lib.require_cuda(pcl_pos, voxel_bound, pcl_bid, voxel_bid,
                      names=["pcl_pos", "voxel_bound", "pcl_bid", "voxel_bid"])
    if pcl_pos.dtype != torch.float32 or voxel_bound.dtype != torch.float32:
        raise RuntimeError("pcl_pos and voxel_bound must be float32")
    if pcl_bid.dtype != torch.int32 or voxel_bid.dtype != torch.int32:
        raise RuntimeError("pcl_bid and voxel_bid must be int32 (the reference calls .int())")


def forward(pcl_pos, voxel_bound, pcl_bid, voxel_bid):
    _check(pcl_pos, voxel_bound, pcl_bid, voxel_bid)
    N, V = pcl_pos.shape[0], voxel_bound.shape[0]
    dev = pcl_pos.device
    mask = torch.zeros((V, N), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().lidf_pcl_aabb_dense_f32(
            _lib.ptr(pcl_pos), _lib.ptr(voxel_bound), _lib.ptr(pcl_bid), _lib.ptr(voxel_bid), N, V,
            _lib.ptr(mask), _lib.current_stream(dev)))
    return mask


def last_voxel(pcl_pos, voxel_bound, pcl_bid, voxel_bid):
    _check(pcl_pos, voxel_bound, pcl_bid, voxel_bid)
    N, V = pcl_pos.shape[0], voxel_bound.shape[0]
    dev = pcl_pos.device
    out = torch.empty((N,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().lidf_pcl_aabb_last_f32(
            _lib.ptr(pcl_pos), _lib.ptr(voxel_bound), _lib.ptr(pcl_bid), _lib.ptr(voxel_bid), N, V,
            _lib.ptr(out), _lib.current_stream(dev)))
    return out
