"""Development aid: compute_ray_aabb, voxel-by-voxel vs grid walk, on the geometry-derived frame
and on a fully occupied grid (run on the GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from implicit_depth_amd import pipeline as pl, query as Q
from implicit_depth_amd.synthetic import synthetic_batch
dev = torch.device("cuda:0")
for B in (1, 4):
    batch, feat = synthetic_batch(B, 240, 320, seed=77)
    batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
    opt = pl.LidfOptions()
    dd = pl.prepare_data(batch, opt); pl.get_valid_points(dd, opt)
    occ = Q.get_occ_vox_bound(dd["valid_xyz"].contiguous(), dd["valid_bid"].to(torch.int32).contiguous(), B, opt.xmin, opt.xmax, opt.grid_res)
    dd.update(Q.get_miss_ray(dd["pred_mask"], dd["fx"], dd["fy"], dd["cx"], dd["cy"]))
    vb, vbid = occ["voxel_bound"], occ["occ_vox_bid"].to(torch.int32).contiguous()
    cases = [("scene V=%d" % vb.shape[0], vb, vbid, occ["voxel_coord"])]
    # fully occupied grid
    res = occ["grid_dims"][0]
    key = torch.arange(B * res ** 3, device=dev)
    coord = torch.stack(((key // res ** 2) % res, (key // res) % res, key % res), 1).int().contiguous()
    lo = occ["xmin"] + coord.float() * occ["part_size"]
    cases.append(("full V=%d" % key.numel(), torch.cat((lo, lo + occ["part_size"]), 1).contiguous(), (key // res ** 3).int().contiguous(), coord))
    for name, vbb, vbi, co in cases:
        for mode in ("voxels", "grid"):
            kw = dict(voxel_coord=co, grid_dims=occ["grid_dims"], batch=B) if mode == "grid" else {}
            for _ in range(3):
                out = Q.compute_ray_aabb(dd["miss_ray_dir"], vbb, dd["ray_bid"], vbi, **kw)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20):
                out = Q.compute_ray_aabb(dd["miss_ray_dir"], vbb, dd["ray_bid"], vbi, **kw)
            torch.cuda.synchronize()
            print("B=%d %-14s %-7s %.3f ms  pairs %d" % (B, name, mode, (time.perf_counter() - t0) / 20 * 1e3, out[1].numel()))
