"""Drop-ins for the reference's two native extensions (extensions/ray_aabb, extensions/pcl_aabb):
`ray_aabb.forward(...)` / `pcl_aabb.forward(...)` with the reference's argument order, dtypes,
dense outputs and error behaviour (CUDA + contiguous required), backed by liblidf_hip.so."""
from . import pcl_aabb, ray_aabb  # noqa: F401
