"""Development aid: does alternating two streams (the small kernels of query i+1 beside the
decoder kernel of query i) raise the query throughput?"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import orc, to_dev, make_module
from implicit_depth_amd.query import lidf_query
dev = torch.device("cuda:0")
scene = orc.synthetic_scene(1, 240, 320, 64, seed=1235)
s = to_dev(scene, dev)
prob = make_module("IMNET", scene["prob_p"], 385, dev); off = make_module("IEF", scene["off_p"], 385, dev)
for prec in ("f32", "f16x3"):
    for nstream in (1, 2, 3):
        streams = [torch.cuda.Stream() for _ in range(nstream)]
        ws = [None] * nstream
        def run(i):
            k = i % nstream
            with torch.cuda.stream(streams[k]), torch.no_grad():
                o = lidf_query(s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"], s["pair_vox"], s["pair_t"],
                               s["feat_grid"], s["vox_feat"], prob, off, ray_flat=s["ray_flat"], workspace=ws[k], precision=prec)
            ws[k] = o["workspace"]
        for i in range(6): run(i)
        torch.cuda.synchronize(); t = time.time()
        K = 30
        for i in range(K): run(i)
        torch.cuda.synchronize(); dt = (time.time() - t) / K
        print("%s, %d stream(s): %.3f ms/query  %.1f Mpts/s" % (prec, nstream, dt * 1e3, scene["P"] / dt / 1e6))
