"""Development aid: N split-f16 queries at the headline shape (target of rocprofv3 --pmc runs)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import orc, to_dev, make_module
from implicit_depth_amd.query import lidf_query
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
scene = orc.synthetic_scene(1, 240, 320, 64, seed=1235)
s = to_dev(scene, dev)
prob = make_module("IMNET", scene["prob_p"], 385, dev); off = make_module("IEF", scene["off_p"], 385, dev)
ws = None
with torch.no_grad():
    for _ in range(n):
        o = lidf_query(s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"], s["pair_vox"], s["pair_t"],
                       s["feat_grid"], s["vox_feat"], prob, off, ray_flat=s["ray_flat"], workspace=ws, precision=prec)
        ws = o["workspace"]
torch.cuda.synchronize()
