"""Ad-hoc timing of the box-test kernels at full size (development aid)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from implicit_depth_amd.synthetic import synthetic_scene
from implicit_depth_amd.query import compute_ray_aabb
from implicit_depth_amd.extensions import ray_aabb, pcl_aabb
dev = torch.device("cuda:0")
def timeit(fn, k=5):
    fn(); torch.cuda.synchronize(); t = time.time()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.time() - t) / k * 1e3
sc = synthetic_scene(1, 240, 320, 1, seed=1)
rd = sc["ray_dir"].to(dev); rb = sc["ray_bid"].to(dev)
for V in (100, 729):
    vb = torch.cat((sc["vox_center"][:V] - 0.125, sc["vox_center"][:V] + 0.125), 1).to(dev).contiguous()
    vbid = torch.zeros(V, dtype=torch.int32, device=dev)
    ms = timeit(lambda: compute_ray_aabb(rd, vb, rb, vbid))
    off, pr, pv, pt = compute_ray_aabb(rd, vb, rb, vbid)
    md = timeit(lambda: ray_aabb.forward(rd, vb, rb, vbid))
    mp = timeit(lambda: pcl_aabb.forward(rd, vb, rb, vbid))
    print("V=%d R=%d: compact pairs=%d %.3f ms | dense ray_aabb %.3f ms (%.0f MB out) | dense pcl_aabb %.3f ms" % (V, rd.shape[0], pr.shape[0], ms, md, V * rd.shape[0] * 12 / 1e6, mp))
