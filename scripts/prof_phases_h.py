"""Development aid: per-phase cycle counters of the split-f16 per-point kernel. Needs a library
built with -DLIDF_PROFILE (LIDF_HIP_LIB=<path>): wavefront 0 of workgroup 0 writes its counters
into the rayfeat rows."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import orc, to_dev, make_module
from implicit_depth_amd.query import lidf_query
dev = torch.device("cuda:0")
scene = orc.synthetic_scene(1, 240, 320, 64, seed=1235)
s = to_dev(scene, dev)
prob = make_module("IMNET", scene["prob_p"], 385, dev); off = make_module("IEF", scene["off_p"], 385, dev)
sys.path.insert(0, ROOT)
from bench import HipEvents
hev = HipEvents(); e0, e1 = hev.create(), hev.create()
with torch.no_grad():
    for _ in range(3):
        o = lidf_query(s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"], s["pair_vox"], s["pair_t"],
                       s["feat_grid"], s["vox_feat"], prob, off, ray_flat=s["ray_flat"], want_rayfeat=True, precision="f16x3", profile_events=(e0.value, e1.value))
torch.cuda.synchronize()
t = o["rayfeat"].view(-1)[:16].view(torch.int64).cpu().tolist()
names = ["loop top + stores", "geometry + PE operands + rays", "net 0 (1 pass)", "net 1 (2 passes)", "-", "-", "-", "-"]
tot = sum(t[:6])
print("workgroup 0: %d shader-clock ticks in %d ticks of the 100 MHz wall clock -> %.2f GHz" % (t[7], t[6], t[7] / max(t[6], 1) * 0.1))
for n, v in zip(names[:6], t[:6]):
    print("%-30s %12d ticks  %5.1f%%  per tile %8.0f" % (n, v, 100.0 * v / max(tot, 1), v / 75.0))
ms = hev.elapsed_ms(e0, e1)
print("points kernel %.3f ms -> %.2f GHz by the tick count of workgroup 0" % (ms, tot / ms / 1e6))
print("total ticks", tot, "per tile", tot / 75.0)

import numpy as np
w = o["rayfeat"].view(-1)[32:32 + 8 * 512].view(torch.int64).cpu().numpy().reshape(512, 4)
t0 = w[:, 0].min()
st, en = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0   # microseconds
print("workgroup start (us): min %.0f  median %.0f  max %.0f ; end: min %.0f median %.0f max %.0f" % (st.min(), np.median(st), st.max(), en.min(), np.median(en), en.max()))
late = st > 100
print("workgroups starting later than 100 us:", int(late.sum()), " their duration median %.0f us ; early ones %.0f us" % (np.median((en - st)[late]) if late.any() else 0, np.median((en - st)[~late])))
hw = w[:, 2]; xcc = w[:, 3] & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = xcc * 1000 + se * 100 + sh * 16 + cu
import collections
cnt = collections.Counter(key.tolist())
print("distinct (xcc,se,sh,cu):", len(cnt), " workgroups per CU histogram:", collections.Counter(cnt.values()))
dur = en - st
for x in range(8):
    m = xcc == x
    print("xcc %d: %3d workgroups, duration min %.0f median %.0f max %.0f us" % (x, int(m.sum()), dur[m].min(), np.median(dur[m]), dur[m].max()))
for q in range(8):
    m = (np.arange(512) // 64) == q
    print("blockIdx %3d..%3d: duration median %.0f us, xcc set %s" % (q * 64, q * 64 + 63, np.median(dur[m]), sorted(set(xcc[m].tolist()))))
