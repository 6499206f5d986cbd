"""Ad-hoc timing of the fused query at the headline shape (development aid, not the bench)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from util import orc, to_dev, make_module
from implicit_depth_amd.query import lidf_query

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
scene = orc.synthetic_scene(1, 240, 320, N, seed=1235)
s = to_dev(scene, dev)
D = 385
prob = make_module("IMNET", scene["prob_p"], D, dev)
off = make_module("IEF", scene["off_p"], D, dev)
ws = None
def run():
    global ws
    with torch.no_grad():
        o = lidf_query(s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"],
                       s["pair_vox"], s["pair_t"], s["feat_grid"], s["vox_feat"], prob, off,
                       ray_flat=s["ray_flat"], workspace=ws)
    ws = o["workspace"]
    return o
for _ in range(2):
    run()
torch.cuda.synchronize()
t = time.time()
K = 5
for _ in range(K):
    o = run()
torch.cuda.synchronize()
dt = (time.time() - t) / K
P = scene["P"]
print("P=%d  %.3f ms/query  %.1f Mpts/s  F_alg %.1f TFLOP/s" % (P, dt * 1e3, P / dt / 1e6, P / dt * 853952 / 1e12))
print("finite:", bool(torch.isfinite(o["pred_pos"]).all()), float(o["pred_pos"][:, 2].mean()))
