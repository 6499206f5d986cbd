"""Development aid: localise a precision loss of the split-f16 kernel by zeroing groups of layer-1
input columns (voxel | rgb | enter xyz | enter sincos | leave xyz | leave sincos | dir)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import orc, run_query, oracle_query
dev = torch.device("cuda:0")
groups = {"none": [], "vox": [(0, 128)], "rgb+dir": [(128, 256), (358, 385)], "enter xyz": [(256, 259)], "enter sincos": [(259, 307)],
          "leave xyz": [(307, 310)], "leave sincos": [(310, 358)], "all PE": [(256, 358)]}
for name, cols in groups.items():
    scene = orc.synthetic_scene(1, 16, 24, 16, seed=7)
    for key in ("prob_p", "off_p"):
        w = scene[key]["linear_1.weight"]
        keep = torch.zeros(w.shape[1], dtype=torch.bool)
        for a, b in cols:
            keep[a:b] = True
        if name != "none":
            w[:, ~keep] = 0     # keep ONLY this group
    ref = oracle_query(scene)
    out = run_query(scene, dev, precision="f16x3")
    print("%-14s only: prob err %.3g  off err %.3g" % (name, float((out["pred_prob_end"].cpu() - ref["pred_prob_end"]).abs().max()),
          float((out["pred_offset"].cpu() - ref["pred_offset"]).abs().max())))
