import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from implicit_depth_amd.synthetic import synthetic_scene
from oracle import lidf_oracle as orc
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch default threads", torch.get_num_threads())
scene = synthetic_scene(1, 240, 320, 64, seed=1235)
w, N = 320, 64
def run(rows):
    R = rows * w; P = R * N
    t0 = time.time()
    orc.query(scene["ray_dir"][:R], scene["ray_pix"][:R], scene["ray_bid"][:R],
              scene["pair_ray"][:P].long(), scene["pair_vox"][:P].long(), scene["pair_t"][:P],
              scene["pair_off"][:R + 1], scene["feat_grid"], scene["vox_feat"],
              scene["prob_p"], scene["off_p"], fast_roi=True)
    return P, time.time() - t0
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    run(1)
    P, t = run(8)
    print(th, "threads:", P, "pts", round(t, 3), "s", round(P / t / 1e6, 4), "Mpts/s", flush=True)
