import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import orc, run_query
dev = torch.device("cuda:0")
scene = orc.synthetic_scene(1, 16, 24, 16, seed=7)
out = run_query(scene, dev, precision="f16x3")
ws = out["workspace"]
st = ws[: 2 * 288 * 1024].view(torch.float16).reshape(2, 288, 64, 8).float().cpu()
w1 = scene["prob_p"]["linear_1.weight"]
for q in (15, 16, 17, 18):
    a = st[0, q]
    print("quad", q, "nonzero per element:", [(int((a[:, i] != 0).sum())) for i in range(8)], "max abs per element:", ["%.2e" % float(a[:, i].abs().max()) for i in range(8)])
# expected XB e0 for lane o=0..3, h=0: lo piece of w1[o, 256]
w = w1[:4, 256]
hi = w.half().float(); lo = (w - hi).half().float()
print("expected wl:", lo.tolist(), " stream XB e0:", st[0, 17, :4, 0].tolist())
print("expected wh:", hi.tolist(), " stream XA e0:", st[0, 16, :4, 0].tolist(), " XA e3:", st[0, 16, :4, 3].tolist())
for net in (0, 1):
    for T in range(8):
        q = 4 + 13 if T == 0 else 4 + 14 + (T - 1) * 30 + 13
        a = st[net, q]
        print("net", net, "tile", T, "XB quad", q, "nonzero per element:", [(int((a[:, i] != 0).sum())) for i in range(8)])
