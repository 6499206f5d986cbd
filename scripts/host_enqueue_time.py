"""Host time of one FrameRunner.run() + metrics() (enqueue only, no sync) against the GPU time of a frame: whether
frames pipelined over several streams are bound by the enqueueing thread. usage: python scripts/host_enqueue_time.py"""
import sys
import time

import torch

sys.path.insert(0, ".")
from implicit_depth_amd import IEF, IMNet, PointNet2Stage, pipeline as pl   # noqa: E402
from implicit_depth_amd.synthetic import init_decoder_params, synthetic_batch   # noqa: E402

dev = torch.device("cuda", 0)
batch, feat = synthetic_batch(1, 240, 320, seed=77)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
feat = feat.to(dev)
torch.manual_seed(3)
pnet, pnet_r = PointNet2Stage(6, 128, 32).to(dev).eval(), PointNet2Stage(6, 128, 32).to(dev).eval()
prob = IMNet(385, 1, 64).to(dev).eval()
prob.load_state_dict(init_decoder_params("IMNET", 385, 7, 5.0))
off = IEF(dev, 385, 1, 64, n_iter=2).to(dev).eval()
off.load_state_dict(init_decoder_params("IEF", 385, 8, 5.0))
offr = IEF(dev, 334, 1, 64, n_iter=2).to(dev).eval()
offr.load_state_dict(init_decoder_params("IEF", 334, 9, 5.0))
opt = pl.LidfOptions(valid_stride=6)
r = pl.FrameRunner(1, 240, 320, dev, pnet, prob, off, opt, pnet_r, offr, side_stream=False)
with torch.no_grad():
    for _ in range(20):
        r.run(batch, feat)
        r.metrics(batch)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):          # (short bursts: the launch queue never fills, the host never waits)
            r.run(batch, feat)
            r.metrics(batch)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        best = min(best, (t1 - t0) / 20)
        print("host enqueue per frame %.3f ms   (burst of 20 incl. drain: %.3f ms per frame)" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
print("best host enqueue per frame %.3f ms" % (best * 1e3))
