"""Development aid: split-f16 query vs the f32 query vs the oracle (small scene), then timing of
both at the headline shape."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import orc, to_dev, make_module, run_query, oracle_query

dev = torch.device("cuda:0")
small = orc.synthetic_scene(2, 24, 32, 16, seed=7)
ref = oracle_query(small)
for prec in ("f32", "f16x3"):
    out = run_query(small, dev, precision=prec)
    for k in ("pred_offset", "pred_prob_end", "pair_pred_pos", "pred_pos"):
        a, b = out[k].cpu(), ref[k]
        print(prec, k, "max|d| = %.3g" % float((a - b).abs().max()), " finite", bool(torch.isfinite(a).all()))
if len(sys.argv) > 1 and sys.argv[1] == "small":
    sys.exit(0)

from implicit_depth_amd.query import lidf_query
scene = orc.synthetic_scene(1, 240, 320, 64, seed=1235)
s = to_dev(scene, dev)
prob = make_module("IMNET", scene["prob_p"], 385, dev)
off = make_module("IEF", scene["off_p"], 385, dev)
res = {}
for prec in ("f32", "f16x3"):
    ws = None
    def run():
        global ws
        with torch.no_grad():
            o = lidf_query(s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"],
                           s["pair_vox"], s["pair_t"], s["feat_grid"], s["vox_feat"], prob, off,
                           ray_flat=s["ray_flat"], workspace=ws, precision=prec)
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
    res[prec] = o
    print("%s: P=%d  %.3f ms/query  %.1f Mpts/s" % (prec, scene["P"], dt * 1e3, scene["P"] / dt / 1e6))
for k in ("pred_offset", "pred_prob_end", "pair_pred_pos", "pred_pos"):
    print(k, "f16x3 vs f32 max|d| = %.3g" % float((res["f16x3"][k] - res["f32"][k]).abs().max()))
