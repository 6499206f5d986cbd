import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from util import orc, to_dev, make_module
from implicit_depth_amd.query import lidf_query
from implicit_depth_amd import _lib
dev = torch.device("cuda:0")
scene = orc.synthetic_scene(1, 240, 320, 64, seed=1235)
s = to_dev(scene, dev)
prob = make_module("IMNET", scene["prob_p"], 385, dev); off = make_module("IEF", scene["off_p"], 385, dev)
with torch.no_grad():
    for _ in range(3):
        o = lidf_query(s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"], s["pair_vox"], s["pair_t"], s["feat_grid"], s["vox_feat"], prob, off, ray_flat=s["ray_flat"])
torch.cuda.synchronize()
# locate rayfeat scratch inside the workspace: replicate query_ws offsets is fiddly; scan instead
ws = o["workspace"]
# rayfeat offset = total - align(R*(128+27)*4)
import ctypes
R = scene["R"]
def al(x): return (x + 255) // 256 * 256
total = _lib.lib().lidf_query_workspace_bytes(R, scene["V"], 0)
# workspace for actual L=8,Lv=4 is laid out with those sizes: recompute
E=51; Ed=27
def l1q_f(L): return 24*((L+1)//2)+8
stream_pts = al(2*(l1q_f(8)+168)*256*4); aux = al(2*72*4)
KH=64; KQ1=(KH+1+3)//4; stream_vox = al(2*KQ1*8*256*4)
D=128+Ed; KH=(D+1)//2; KQ1=(KH+1+3)//4; stream_ray = al(2*KQ1*8*256*4)
voxpart = al(scene["V"]*512*4); raypart = al(R*512*4)
off_rayfeat = stream_pts+aux+stream_vox+stream_ray+voxpart+raypart
t = ws[off_rayfeat:off_rayfeat+64].view(torch.int64).cpu().tolist()
names = ["loop-top+stores", "geo/next-prep", "base-init", "L1 loop+tail", "pass head (L4 of prev, u, gen0)", "L4 + out_act (last pass)", "layer 2", "layer 3"]
tot = sum(t[:8])
for n, v in zip(names, t[:8]):
    print("%-34s %12d ticks  %5.1f%%  per tile %8.0f" % (n, v, 100.0*v/max(tot,1), v/150.0))
print("total ticks", tot, "per tile", tot/150.0)
