"""Development aid: per-phase s_memtime counters of the f32 per-point kernel (wavefront 0 of
workgroup 0). Builds scripts/liblidf_prof.so with -DLIDF_PROFILE if needed (run on the GPU box:
`python scripts/prof_phases.py [dense|scene|ragged]`)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
LIB = os.path.join(ROOT, "scripts", "liblidf_prof.so")
if "--build" in sys.argv or not os.path.exists(LIB):
    from implicit_depth_amd.csrc import build as B
    src = [os.path.join(B.HERE, s) for s in B.SOURCES]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                    "-ffp-contract=off", "-fvisibility=hidden", "-DLIDF_PROFILE", "-mllvm",
                    "-pragma-unroll-threshold=8000000", "-mllvm", "-amdgpu-mfma-vgpr-form", "-I", os.path.join(ROOT, "include"), "-I", B.HERE,
                    "-o", LIB] + src, check=True)
    if "--build" in sys.argv:
        sys.exit(0)
os.environ["LIDF_HIP_LIB"] = LIB
import torch  # noqa: E402
from util import make_module, orc, to_dev  # noqa: E402
from implicit_depth_amd.query import lidf_query  # noqa: E402
from bench import HipEvents  # noqa: E402

kind = [a for a in sys.argv[1:] if not a.startswith("--")]
kind = kind[0] if kind else "dense"
dev = torch.device("cuda:0")
scene = orc.synthetic_scene(1, 240, 320, 64, seed=1235, ragged=kind == "ragged")
s = to_dev(scene, dev)
prob = make_module("IMNET", scene["prob_p"], 385, dev)
off = make_module("IEF", scene["off_p"], 385, dev)
if kind == "scene":
    # rays, voxels and pairs as the candidate generator produces them on a geometry-derived frame
    from implicit_depth_amd import PointNet2Stage, pipeline as pl
    from implicit_depth_amd.synthetic import synthetic_batch
    batch, feat = synthetic_batch(int(os.environ.get("FRAMES", "1")), 240, 320, seed=77)
    torch.manual_seed(3)
    pn = PointNet2Stage(6, 128, 32).to(dev).eval()
    with torch.no_grad():
        ok, dd = pl.lidf_forward({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()},
                                 feat.to(dev), pn, prob, off)
    s.update({"ray_dir": dd["miss_ray_dir"], "ray_pix": dd["ray_pix"], "ray_bid": dd["ray_bid"],
              "ray_flat": dd["ray_flat"], "pair_off": dd["pair_off"], "pair_ray": dd["pair_ray"],
              "pair_vox": dd["pair_vox"], "pair_t": dd["pair_t"], "feat_grid": dd["full_rgb_feat"],
              "vox_feat": dd["occ_voxel_feat"]})
    scene = dict(scene, P=int(dd["pair_ray"].shape[0]))
hev = HipEvents()
e0, e1 = hev.create(), hev.create()
with torch.no_grad():
    for _ in range(3):
        o = lidf_query(s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"], s["pair_vox"],
                       s["pair_t"], s["feat_grid"], s["vox_feat"], prob, off, ray_flat=s["ray_flat"],
                       want_rayfeat=True, profile_events=(e0, e1))
torch.cuda.synchronize()
t = o["rayfeat"].view(-1)[:16].view(torch.int64).cpu().tolist()
names = ["loop top, outputs", "geometry, PE operands", "rank-1 row requests", "layer 1 (PE + rank-1 MFMAs)",
         "pass: u + first H1", "pass tail (layer 4) + LDS->base", "pass: layer 2", "pass: layer 3"]
ntile = (scene["P"] + 127) // 128
per = ntile // 256 + (1 if ntile % 256 else 0)
tot = sum(t)
for n, v in zip(names, t):
    print("%-34s %12d cycles %5.1f%%  per wave-tile %9.0f" % (n, v, 100.0 * v / max(tot, 1), v / per))
print("total %d cycles, per wave-tile %.0f (MFMA floor 2778 x 64 = 177792)" % (tot, tot / per))
print("points kernel %.3f ms" % hev.elapsed_ms(e0, e1))
# per-wavefront start / end (100 MHz wall clock) and shader-clock totals
nw = min(256, ntile) * 4
w = o["rayfeat"].view(-1)[32:32 + 8 * nw].view(torch.int64).cpu().view(nw, 4)
t0 = w[:, 0].min()
end = (w[:, 1] - t0).double() / 100.0   # us
start = (w[:, 0] - t0).double() / 100.0
cyc = w[:, 2].double()
import numpy as np
print("wavefront end times us: min %.1f  median %.1f  p90 %.1f  max %.1f; start max %.1f" % (
    end.min(), end.median(), np.percentile(end.numpy(), 90), end.max(), start.max()))
print("shader cycles per wavefront: min %.0f median %.0f max %.0f; effective clock of the slowest %.2f GHz" % (
    cyc.min(), cyc.median(), cyc.max(), cyc[end.argmax()] / ((end.max() - start[end.argmax()]) * 1e3)))
