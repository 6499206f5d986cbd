"""Worker of tests/test_ddp_gpu.py, launched by torch.distributed.run: the training steps of the path under the
reference's own wrapper — DistributedDataParallel(find_unused_parameters=True), trainers/train_lidf.py:115-121 —
with the product modules (IMNet / IEF / PointNet2Stage) inside one nn.Module per stage, as the reference's LIDF /
RefineNet are, and the fused training calls (lidf_query_train, lidf_refine_train) in its forward. Every rank runs
its own frame of a 2-frame batch; the parameters' reduced gradients are compared with a single-process step over
the whole batch (the product itself, no DDP). Steps: a normal one; one where a rank has no (ray, voxel) pair and
takes part with an empty list; the reference's success-flag protocol (models/pipeline.py:662-701: the flags are
all-reduced and every rank leaves the step without a backward) followed by a normal step."""
import os
import sys

import torch
import torch.distributed as dist
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class QueryStage(nn.Module):
    """Stage 1 as DDP sees it in the reference (LIDF: pnet_model, prob_dec, offset_dec as sub-modules)."""

    def __init__(self, pnet, prob, off):
        super().__init__()
        self.pnet_model, self.prob_dec, self.offset_dec = pnet, prob, off

    def forward(self, s):
        from implicit_depth_amd.query import lidf_query_train
        vox_feat = self.pnet_model(s["pn_inp"], s["pn_vox"], n_vox=s["V"])
        o = lidf_query_train(s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"], s["pair_vox"],
                             s["pair_t"], s["feat_grid"], vox_feat, self.prob_dec, self.offset_dec)
        return o["pred_pos"], o["pred_prob_end"]


class RefineStage(nn.Module):
    """Stage 2 (RefineNet: pnet_model, offset_dec)."""

    def __init__(self, pnet, off):
        super().__init__()
        self.pnet_model, self.offset_dec = pnet, off

    def forward(self, s, pred_pos, max_pair_id):
        from implicit_depth_amd.query import lidf_refine_train
        pos, _ = lidf_refine_train(s["ray_dir"], s["ray_pix"], s["ray_bid"], s["ray_flat"], pred_pos, max_pair_id,
                                   s["pair_vox"], s["voxel_bound"], s["voxel_bid"], s["rgb_img"], s["feat_grid"],
                                   s["valid_inp"], s["valid_vox"], self.pnet_model, self.offset_dec, forward_times=2)
        return pos


def frame_inputs(scene, b, dev, gen, empty=False):
    """Frame b of an oracle synthetic_scene as a rank's own batch of one frame (indices re-based)."""
    from implicit_depth_amd.dist import slice_rays
    hw, Vb = scene["h"] * scene["w"], scene["V"] // scene["B"]
    s = slice_rays(scene, b * hw, (b + 1) * hw)
    s["ray_bid"] = s["ray_bid"] - b
    s["pair_vox"] = s["pair_vox"] - b * Vb
    s["feat_grid"] = scene["feat_grid"][b:b + 1].contiguous()
    s["V"] = Vb
    n = 400
    s["pn_inp"] = torch.randn(n, 6, generator=gen) * 0.3
    s["pn_vox"] = torch.randint(0, Vb, (n,), generator=gen).int()
    s["gt_pos"] = torch.randn(hw, 3, generator=gen)
    s["w_prob"] = torch.randn(s["P"], generator=gen)
    ctr = scene["vox_center"][b * Vb:(b + 1) * Vb]
    s["voxel_bound"] = torch.cat((ctr - 0.125, ctr + 0.125), 1)
    s["voxel_bid"] = torch.zeros(Vb, dtype=torch.int32)
    s["rgb_img"] = torch.randn(1, 3, scene["h"], scene["w"], generator=gen)
    s["valid_inp"] = torch.randn(300, 6, generator=gen) * 0.2
    s["valid_vox"] = torch.randint(0, Vb, (300,), generator=gen).int()
    # stage 2's inputs arrive detached from a frozen stage 1 (trainers/train_refine.py:73): positions near voxel
    # centres, a selection that names pairs of this frame's list or its dummy row
    # (every position strictly inside an occupied voxel: a ray without a candidate takes its end voxel from the dummy
    # row = voxel 0 OF THE BATCH when its point lies in no voxel, models/pipeline.py:924,942-943 — a cross-frame
    # quirk of the reference that a per-frame shard cannot and need not reproduce)
    s["pos0"] = ctr[torch.randint(0, Vb, (hw,), generator=gen)] + (torch.rand(hw, 3, generator=gen) - 0.5) * 0.2
    s["mid"] = torch.randint(0, s["P"] + 1, (hw,), generator=gen)
    if empty:    # no ray meets a voxel in this frame
        s["pair_off"] = torch.zeros_like(s["pair_off"])
        for k in ("pair_ray", "pair_vox"):
            s[k] = s[k][:0].contiguous()
        s["pair_t"], s["w_prob"], s["P"] = s["pair_t"][:0].contiguous(), s["w_prob"][:0], 0
        s["mid"] = torch.zeros_like(s["mid"])
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in s.items()}


def whole_batch(frames):
    """The rank frames joined back into one batch (what a single process would be handed)."""
    dev = frames[0]["ray_dir"].device
    out, r0, p0, v0, n0 = {}, 0, 0, 0, 0
    cat = {k: [] for k in ("ray_dir", "ray_pix", "ray_bid", "ray_flat", "pair_ray", "pair_vox", "pair_t", "feat_grid",
                           "pn_inp", "pn_vox", "gt_pos", "w_prob", "voxel_bound", "voxel_bid", "rgb_img", "valid_inp",
                           "valid_vox", "pos0")}
    offs = [torch.zeros(1, dtype=torch.int32, device=dev)]
    for b, f in enumerate(frames):
        cat["ray_dir"].append(f["ray_dir"]), cat["ray_pix"].append(f["ray_pix"]), cat["ray_flat"].append(f["ray_flat"])
        cat["ray_bid"].append(f["ray_bid"] + b), cat["voxel_bid"].append(f["voxel_bid"] + b)
        cat["pair_ray"].append(f["pair_ray"] + r0), cat["pair_vox"].append(f["pair_vox"] + v0)
        cat["pn_vox"].append(f["pn_vox"] + v0), cat["valid_vox"].append(f["valid_vox"] + v0)
        for k in ("pair_t", "feat_grid", "pn_inp", "gt_pos", "w_prob", "voxel_bound", "rgb_img", "valid_inp", "pos0"):
            cat[k].append(f[k])
        offs.append(f["pair_off"][1:] + p0)
        r0, p0, v0 = r0 + f["ray_dir"].shape[0], p0 + f["P"], v0 + f["V"]
    out = {k: torch.cat(v, 0).contiguous() for k, v in cat.items()}
    out["pair_off"] = torch.cat(offs).contiguous()
    out["V"], out["P"] = v0, p0
    # a frame's dummy row (index P_frame) is the whole list's dummy row (index P)
    mids, q0 = [], 0
    for f in frames:
        mids.append(torch.where(f["mid"] >= f["P"], torch.full_like(f["mid"], p0), f["mid"] + q0))
        q0 += f["P"]
    out["mid"] = torch.cat(mids).contiguous()
    return out


def stage1_loss(pred_pos, prob, s):
    # the reference's structure (models/pipeline.py:468-490): L1 on pred_pos + a term on the logits
    return (pred_pos - s["gt_pos"]).abs().sum() + (prob[:, 0] * s["w_prob"]).sum()


def grads_of(mod):
    return {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in mod.named_parameters()}


def check(got, ref, what, rel=2e-4):
    for k, g in ref.items():
        a = got[k]
        assert (a is None) == (g is None), (what, k)
        if g is None:
            continue
        scale = max(g.abs().max().item(), 1e-3)
        err = (a - g).abs().max().item()
        assert err <= rel * scale, (what, k, err, scale)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    share_gpu = os.environ.get("LIDF_TEST_SHARE_GPU") == "1"   # every rank on cuda:0, gloo (1-GPU box)
    if share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if share_gpu:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from torch.nn.parallel import DistributedDataParallel as DDP
    from util import make_module, make_pointnet, orc

    scene = orc.synthetic_scene(world, 10, 12, 5, seed=301, ragged=True)
    D = scene["D"]
    pn_p, pnr_p = orc.init_pointnet(5, 1.5), orc.init_pointnet(6, 1.5)
    offr_p = orc.randomize_biases(orc.init_decoder("IEF", 334, 77, 5.0), 78)

    def modules():
        q = QueryStage(make_pointnet(pn_p, dev), make_module("IMNET", scene["prob_p"], D, dev),
                       make_module("IEF", scene["off_p"], D, dev)).train()
        r = RefineStage(make_pointnet(pnr_p, dev), make_module("IEF", offr_p, 334, dev)).train()
        return q, r

    for case in ("normal", "one rank without pairs", "after a skipped step"):
        gen = torch.Generator().manual_seed(7)
        empty_rank = world - 1 if case == "one rank without pairs" and world > 1 else -1
        frames = [frame_inputs(scene, b, dev, gen, empty=(b == empty_rank)) for b in range(world)]
        mine, allf = frames[rank], whole_batch(frames)
        # ---- single process over the whole batch (no DDP): the mean of the ranks' losses
        q1, r1 = modules()
        pos, prob = q1(allf)
        (stage1_loss(pos, prob, allf) / world).backward()
        posr = r1(allf, allf["pos0"], allf["mid"])
        ((posr - allf["gt_pos"]).abs().sum() / world).backward()
        ref_q, ref_r = grads_of(q1), grads_of(r1)
        # ---- the same modules under the reference's wrapper, one frame per rank (DDP averages the gradients)
        q2, r2 = modules()
        dq = DDP(q2, device_ids=[local], find_unused_parameters=True)
        dr = DDP(r2, device_ids=[local], find_unused_parameters=True)
        if case == "after a skipped step":
            # the reference's protocol when a rank finds no pair: flags all-reduced inside forward, every rank
            # returns without a backward (pipeline.py:689-701); the next step must reduce as usual
            pos, prob = dq(mine)
            posr = dr(mine, mine["pos0"], mine["mid"])
            flag = torch.tensor([0.0 if rank == world - 1 else 1.0], device=dev)
            dist.all_reduce(flag)
            assert flag.item() == world - 1
            dq.zero_grad(set_to_none=True), dr.zero_grad(set_to_none=True)
            del pos, prob, posr
        pos, prob = dq(mine)
        stage1_loss(pos, prob, mine).backward()
        posr = dr(mine, mine["pos0"], mine["mid"])
        (posr - mine["gt_pos"]).abs().sum().backward()
        check(grads_of(q2), ref_q, "stage 1, " + case)
        check(grads_of(r2), ref_r, "stage 2, " + case)
        dist.barrier()
    if rank == 0:
        print("DDP_WORKER_OK world=%d" % world, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
