import sys, os, faulthandler
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import torch
from util import make_module, make_pointnet, orc
from implicit_depth_amd import pipeline as pl
from implicit_depth_amd.synthetic import init_decoder_params, synthetic_batch
cuda = torch.device('cuda:0')
refine = int(sys.argv[1]); B,h,w = 1,240,320
pnet = make_pointnet(orc.init_pointnet(3, 1.5), cuda)
pnet_r = make_pointnet(orc.init_pointnet(4, 1.5), cuda)
prob = make_module("IMNET", init_decoder_params("IMNET", 385, 7, 5.0), 385, cuda)
off = make_module("IEF", init_decoder_params("IEF", 385, 8, 5.0), 385, cuda)
offr = make_module("IEF", init_decoder_params("IEF", 334, 9, 5.0), 334, cuda)
opt = pl.LidfOptions()
runner = pl.FrameRunner(B,h,w,cuda,pnet,prob,off,opt, pnet_r if refine else None, offr if refine else None)
batch, feat = synthetic_batch(B,h,w,seed=77)
batch = {k:(v.to(cuda) if torch.is_tensor(v) else v) for k,v in batch.items()}; feat=feat.to(cuda)
with torch.no_grad():
    runner.run(batch, feat)
    torch.cuda.synchronize(); print("frame done", runner.counts(), flush=True)
    ok, dd = runner.result()
    x = dd["pnet_inp"].clone(); idx = dd["revidx"].clone(); V = dd["counts"]["V"]
    print("calling pnet", x.shape, idx.shape, V, int(idx.max()), flush=True)
    o = pnet(x, idx, n_vox=V)
    torch.cuda.synchronize(); print("pnet ok", float(o.abs().sum()), flush=True)
    ok2, ref = pl.lidf_forward(batch, feat, pnet, prob, off, opt)
    torch.cuda.synchronize(); print("stepwise ok", flush=True)
    print("occ feat equal", torch.equal(ref["occ_voxel_feat"], dd["occ_voxel_feat"]), torch.equal(ref["pred_pos"], dd["pred_pos"]))
