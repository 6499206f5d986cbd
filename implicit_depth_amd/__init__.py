"""implicit_depth_amd — MI355X-native LIDF per-point implicit-depth query path.

Host-side mirror of the reference interface for this path (names follow the reference):
  decoders.get_embedder / IMNet / IEF        <- models/implicit_net.py
  pointnet.PointNet2Stage                    <- models/pointnet.py
  extensions.ray_aabb.forward / pcl_aabb     <- extensions/{ray_aabb,pcl_aabb}
  query.lidf_query / get_miss_ray / ...      <- models/pipeline.py:203-466, 593-596
All compute goes through csrc/liblidf_hip.so (C ABI in include/lidf_hip.h).
"""
from . import _lib  # noqa: F401
from .decoders import IEF, IMNet, Embedder, decoders_forward, get_embedder  # noqa: F401
from .pointnet import PointNet2Stage  # noqa: F401

__all__ = ["IEF", "IMNet", "Embedder", "PointNet2Stage", "decoders_forward", "get_embedder"]
