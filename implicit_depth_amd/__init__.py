"""implicit_depth_amd — MI355X-native LIDF per-point implicit-depth query path.

Host-side mirror of the reference interface for this path (names follow the reference):
  decoders.get_embedder / IMNet / IEF        <- models/implicit_net.py
  pointnet.PointNet2Stage                    <- models/pointnet.py
  extensions.ray_aabb.forward / pcl_aabb     <- extensions/{ray_aabb,pcl_aabb}
  query.get_miss_ray / compute_ray_aabb / lidf_query / lidf_refine / get_occ_vox_bound /
        depth_metrics / lidf_query_train     <- models/pipeline.py:162-466, 577-627, 922-1041
  pipeline.lidf_forward / refine_forward     <- LIDF.forward / RefineNet.forward, eval flavour
  torch_ext.ext()                            <- the pybind11 operator module (extensions/*/jit.py)
All compute goes through csrc/liblidf_hip.so (C ABI in include/lidf_hip.h).
"""
from . import _lib  # noqa: F401
from .decoders import (IEF, IMNet, Embedder, decoders_forward, decoders_forward_train,  # noqa: F401
                       get_embedder)
from .pointnet import PointNet2Stage  # noqa: F401

__all__ = ["IEF", "IMNet", "Embedder", "PointNet2Stage", "decoders_forward", "decoders_forward_train",
           "get_embedder"]
