"""Every BASELINE.json configuration on the GPU, HIP path (through the C ABI) against the oracle.

  configs[0]  64x64 grid, 16 candidates/ray: the whole frame against the oracle (P = 65,536)
  configs[1]  240x320x64, one frame: tests/test_edge_gpu.py::test_full_size_properties
  configs[2]  the per-GPU shard of the 32-frame batch over 8 GPUs: 4 frames x 240x320x64
  configs[3]  stage 1 + stage 2 at 240x320x64: test_config3_refine_full_size below
  configs[4]  256 candidates/ray: one frame, and the per-GPU shard of the 32-frame batch (4 frames)
At the full sizes the oracle cannot run the whole frame in seconds, so the checks are
size-independent properties (softmax sums to 1 per ray, the arg-max is a maximal logit of its own
ray, select and depth are pure gathers) plus >= 96 random WHOLE rays per frame against the oracle.
"""
import functools

import pytest
import torch

from util import TOL, make_module, make_pointnet, oracle_query, orc, run_query, to_dev

pytestmark = pytest.mark.gpu


@functools.lru_cache(maxsize=1)
def _scene(B, h, w, N, seed, ragged=False):
    return orc.synthetic_scene(B, h, w, N, seed=seed, ragged=ragged)


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("ragged", [False, True])
def test_config0_64x64x16_whole_frame(cuda, precision, ragged):
    scene = _scene(1, 64, 64, 16, 1234, ragged)
    ref = oracle_query(scene, fast_roi=True)
    got = run_query(scene, cuda, precision=precision)
    for k in ("pred_offset", "pred_prob_end", "pair_pred_pos", "pred_pos"):
        assert (got[k].cpu() - ref[k]).abs().max().item() <= TOL, k
    assert (got["pred_prob_end_softmax"].cpu() - ref["pred_prob_end_softmax"]).abs().max().item() <= 1e-5
    depth_ref = torch.zeros(64 * 64)
    depth_ref[scene["ray_flat"].long()] = ref["pred_pos"][:, 2]
    assert (got["depth"].cpu().reshape(-1) - depth_ref).abs().mean().item() <= TOL  # depth L1 vs ref
    gid, rid = got["max_pair_id"].cpu(), ref["max_pair_id"]
    sm = ref["pred_prob_end_softmax"]
    for r in (gid != rid).nonzero().flatten().tolist():   # only float-noise ties may differ
        assert gid[r] < scene["P"] and rid[r] < scene["P"] and abs(sm[gid[r]] - sm[rid[r]]) <= 1e-6


@pytest.mark.parametrize("ragged", [False, True])
def test_config0_offsets_for_the_selected_pairs_only(cuda, ragged):
    """lidf_query(offsets="selected") (opt-in, LidfQueryArgs.offsets_selected) on configs[0], whole frame, against
    the oracle and against the default call: the offset decoder runs on the arg-max pair of every ray only —
    pred_prob_end, softmax, max_pair_id, pred_pos and the depth map are bit-identical to the default (the
    reference reads per-pair offsets only through pred_pos, models/pipeline.py:452-454), pred_offset /
    pair_pred_pos hold the default's values at the selected rows (<= 1e-4 of the oracle) and NaN elsewhere."""
    scene = _scene(1, 64, 64, 16, 1234, ragged)
    ref = oracle_query(scene, fast_roi=True)
    full = run_query(scene, cuda)
    got = run_query(scene, cuda, offsets="selected")
    for k in ("pred_prob_end", "pred_prob_end_softmax", "max_pair_id", "pred_pos", "depth"):
        assert torch.equal(got[k], full[k]), k
    P = scene["P"]
    mid = got["max_pair_id"]
    has = mid < P                                   # rays with at least one candidate
    sel = mid[has]
    assert bool(has.all()) == (not ragged)
    for k in ("pred_offset", "pair_pred_pos"):
        assert torch.equal(got[k][sel], full[k][sel]), k
        assert (got[k][sel].cpu() - ref[k][sel.cpu()]).abs().max().item() <= TOL, k
        rest = torch.ones(P, dtype=torch.bool, device=cuda)
        rest[sel] = False
        assert torch.isnan(got[k][rest]).all(), k   # rows no launch writes do not look like results
    assert (got["pred_pos"].cpu() - ref["pred_pos"]).abs().max().item() <= TOL


def check_full_size(scene, got, cuda, rays_per_frame=96):
    B, N, R, P = scene["B"], scene["N"], scene["R"], scene["P"]
    assert P == R * N
    sm = got["pred_prob_end_softmax"]
    assert torch.isfinite(got["pair_pred_pos"]).all() and torch.isfinite(sm).all()
    assert (sm.reshape(R, N).sum(1) - 1).abs().max().item() <= 1e-5
    mid = got["max_pair_id"]
    ar = torch.arange(R, device=cuda)
    assert ((mid >= ar * N) & (mid < (ar + 1) * N)).all()                  # inside its own ray
    logit = got["pred_prob_end"][:, 0].reshape(R, N)
    assert (logit.gather(1, (mid - ar * N).unsqueeze(1))[:, 0] >= logit.max(1).values - 1e-6).all()
    assert (got["pred_pos"] == got["pair_pred_pos"][mid]).all()
    assert (got["depth"].reshape(-1) == got["pred_pos"][:, 2]).all()
    g = torch.Generator().manual_seed(0)
    hw = scene["h"] * scene["w"]
    rows = torch.cat([b * hw + torch.randperm(hw, generator=g)[:rays_per_frame] for b in range(B)]).sort().values
    n = rows.shape[0]
    pidx = (rows.unsqueeze(1) * N + torch.arange(N)).reshape(-1)
    ref = orc.query(scene["ray_dir"][rows], scene["ray_pix"][rows], scene["ray_bid"][rows],
                    torch.arange(n).repeat_interleave(N), scene["pair_vox"][pidx].long(),
                    scene["pair_t"][pidx], None, scene["feat_grid"], scene["vox_feat"],
                    scene["prob_p"], scene["off_p"], fast_roi=True)
    for k in ("pred_offset", "pred_prob_end", "pair_pred_pos"):
        assert (got[k][pidx.to(cuda)].cpu() - ref[k]).abs().max().item() <= TOL, k
    assert (got["pred_pos"][rows.to(cuda)].cpu() - ref["pred_pos"]).abs().max().item() <= TOL
    dsel = got["depth"].reshape(-1)[rows.to(cuda)].cpu()
    assert (dsel - ref["pred_pos"][:, 2]).abs().mean().item() <= TOL            # depth L1 vs ref


@pytest.mark.parametrize("gf", [32, 128])
def test_config1_at_other_decoder_widths_through_the_chain_launch(cuda, gf):
    """configs[1] (240x320x64, P = 4,915,200) with decoders of gf_dim 32 / 128 — one register-chained launch per decoder
    and slab (lidf_decoder_chain_f32; 8 slabs at gf 32, one at gf 128): the size-independent properties over the whole
    frame and 96 random whole rays against the oracle at that width."""
    from implicit_depth_amd import IEF, IMNet, generic
    from implicit_depth_amd.query import lidf_query
    scene = dict(_scene(1, 240, 320, 64, 1235))
    D = scene["D"]
    scene["prob_p"], scene["off_p"] = orc.init_decoder("IMNET", D, 7, 5.0, gf=gf), orc.init_decoder("IEF", D, 8, 5.0, gf=gf)
    prob, off = IMNet(D, 1, gf), IEF(cuda, D, 1, gf, n_iter=2)
    prob.load_state_dict(scene["prob_p"]), off.load_state_dict(scene["off_p"])
    prob, off = prob.to(cuda).eval(), off.to(cuda).eval()
    assert generic.chain_ok(prob) and generic.chain_ok(off)
    s = to_dev(scene, cuda)
    depth = torch.zeros((1, 240, 320), device=cuda)
    with torch.no_grad():
        got = lidf_query(s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"], s["pair_vox"],
                         s["pair_t"], s["feat_grid"], s["vox_feat"], prob, off, ray_flat=s["ray_flat"], depth=depth)
    got["depth"] = depth
    check_full_size(scene, got, cuda)


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_config2_shard_4_frames(cuda, precision):
    """configs[2]: 32 frames over 8 GPUs = 4 frames of 240x320x64 per GPU (P = 19,660,800)."""
    scene = _scene(4, 240, 320, 64, 1236)
    got = run_query(scene, cuda, precision=precision)
    check_full_size(scene, got, cuda)


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_config4_256_candidates(cuda, precision):
    """configs[4]: dense resample, 256 candidates per ray, one 240x320 frame (P = 19,660,800)."""
    scene = _scene(1, 240, 320, 256, 1238)
    got = run_query(scene, cuda, precision=precision)
    check_full_size(scene, got, cuda)


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_full_size_ragged(cuda, precision):
    """240x320 rays with U{0..64} candidates each (P ~ 2.46 M, not a multiple of the 32-point
    wave-tile): the list is long enough for the dynamic tile hand-out of the per-point kernel, has
    empty rays and tiles that hold up to 32 rays. Size-independent properties plus 160 random WHOLE
    rays against the oracle."""
    scene = _scene(1, 240, 320, 64, 1240, True)
    R, P = scene["R"], scene["P"]
    assert P % 32 != 0 and P > 32 * 1024 * 32
    got = run_query(scene, cuda, precision=precision)
    off = scene["pair_off"].long()
    cnt = off[1:] - off[:-1]
    sm = got["pred_prob_end_softmax"]
    assert torch.isfinite(got["pair_pred_pos"]).all() and torch.isfinite(sm).all()
    ray = scene["pair_ray"].long().to(cuda)
    sums = torch.zeros(R, device=cuda).index_add_(0, ray, sm)
    assert (sums[(cnt > 0).to(cuda)] - 1).abs().max().item() <= 1e-5
    mid = got["max_pair_id"]
    empty = (cnt == 0).to(cuda)
    assert (mid[empty] == P).all() and (got["pred_pos"][empty] == 0).all()
    ne = ~empty
    assert (ray[mid[ne]] == torch.arange(R, device=cuda)[ne]).all()        # inside its own ray
    assert (got["pred_pos"][ne] == got["pair_pred_pos"][mid[ne]]).all()
    assert (got["depth"].reshape(-1) == got["pred_pos"][:, 2]).all()
    g = torch.Generator().manual_seed(1)
    rows = torch.randperm(R, generator=g)[:160].sort().values
    rows = torch.cat((rows, torch.tensor([R - 1]))).unique()                # the ray of the partial last tile too
    c = cnt[rows]
    pidx = torch.cat([torch.arange(off[r], off[r + 1]) for r in rows.tolist()])
    local_off = torch.zeros(rows.numel() + 1, dtype=torch.int32)
    local_off[1:] = torch.cumsum(c, 0).int()
    ref = orc.query(scene["ray_dir"][rows], scene["ray_pix"][rows], scene["ray_bid"][rows],
                    torch.arange(rows.numel()).repeat_interleave(c), scene["pair_vox"][pidx].long(),
                    scene["pair_t"][pidx], local_off, scene["feat_grid"], scene["vox_feat"],
                    scene["prob_p"], scene["off_p"], fast_roi=True)
    for k in ("pred_offset", "pred_prob_end", "pair_pred_pos"):
        assert (got[k][pidx.to(cuda)].cpu() - ref[k]).abs().max().item() <= TOL, k
    assert (got["pred_pos"][rows.to(cuda)].cpu() - ref["pred_pos"]).abs().max().item() <= TOL
    assert (sm[pidx.to(cuda)].cpu() - ref["pred_prob_end_softmax"]).abs().max().item() <= 1e-5


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_config4_shard_4_frames_256_candidates(cuda, precision):
    """configs[4] at its per-GPU shape: 32 frames over 8 GPUs = 4 frames of 240x320 with 256
    candidates per ray (P = 78,643,200 points per GPU)."""
    scene = _scene(4, 240, 320, 256, 1239)
    got = run_query(scene, cuda, precision=precision)
    check_full_size(scene, got, cuda, rays_per_frame=32)


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_config3_refine_full_size(cuda, precision):
    """configs[3] at its stated shape: dense stage 1 (240x320 rays x 64 candidates, V = 729) with the
    per-ray feature rows kept, then 2 x get_pred_refine (RefineNet.forward, models/pipeline.py:
    922-1041) with 10,000 valid points: PointNet2Stage over 86,800 points, IEF D = 334 on every ray.
    Stage 1 is checked by check_full_size; each refine iteration is compared over the WHOLE frame with
    orc.refine_step fed the state the HIP path was fed (end voxel ids exact, positions <= 1e-4),
    and the chained 2-iteration call must reproduce the two single iterations bit for bit."""
    from implicit_depth_amd.query import lidf_query, lidf_refine
    from implicit_depth_amd.synthetic import init_decoder_params
    scene = _scene(1, 240, 320, 64, 1237)
    B, h, w, V, R = scene["B"], scene["h"], scene["w"], scene["V"], scene["R"]
    assert (B, h, w, V, scene["N"]) == (1, 240, 320, 729, 64)
    s = to_dev(scene, cuda)
    prob = make_module("IMNET", scene["prob_p"], scene["D"], cuda)
    off = make_module("IEF", scene["off_p"], scene["D"], cuda)
    depth = torch.zeros((B, h, w), device=cuda)
    with torch.no_grad():
        s1 = lidf_query(s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"],
                        s["pair_vox"], s["pair_t"], s["feat_grid"], s["vox_feat"], prob, off,
                        ray_flat=s["ray_flat"], depth=depth, want_rayfeat=True, precision=precision)
    s1["depth"] = depth
    check_full_size(scene, s1, cuda)
    # stage-2 inputs as bench.py's refine_setup draws them (valid_sample_num = 10,000)
    g = torch.Generator().manual_seed(4321)
    vb = torch.cat((scene["vox_center"] - 0.125, scene["vox_center"] + 0.125), 1)
    vbid = torch.zeros(V, dtype=torch.int32)
    rgb = torch.randn(B, 3, h, w, generator=g)
    valid_inp = torch.randn(10000, 6, generator=g) * 0.2
    valid_vox = torch.randint(0, V, (10000,), generator=g).int()
    pnet_p = orc.init_pointnet(5, 1.5)
    offr_p = init_decoder_params("IEF", 334, 9, 5.0)
    pnet, offr = make_pointnet(pnet_p, cuda), make_module("IEF", offr_p, 334, cuda)
    # the per-ray ROI rows: the HIP rows against the restatement first, then shared by both sides
    boxes = orc.roi_boxes(scene["ray_pix"].long(), scene["ray_bid"].long(), h, w, 8)
    ray_rgb = orc.roi_align_fast(scene["feat_grid"], boxes).reshape(R, -1)
    assert (s1["rayfeat"][:, :128].cpu() - ray_rgb).abs().max().item() <= 2e-5
    mid = s1["max_pair_id"].cpu()

    def hip(pos, times):
        with torch.no_grad():
            return lidf_refine(s["ray_dir"], s["ray_pix"], s["ray_bid"], s["ray_flat"], pos,
                               s1["max_pair_id"], s["pair_vox"], vb.to(cuda), vbid.to(cuda), rgb.to(cuda),
                               s["feat_grid"], valid_inp.to(cuda), valid_vox.to(cuda), pnet, offr,
                               forward_times=times, rayfeat=s1["rayfeat"], precision=precision)
    cur = s1["pred_pos"]
    for it in range(2):
        got, gev = hip(cur, 1)
        ref, ev, _ = orc.refine_step(cur.cpu(), scene["ray_dir"], scene["ray_pix"], scene["ray_bid"],
                                     scene["ray_flat"], mid, scene["pair_vox"], vb, vbid, rgb,
                                     scene["feat_grid"], valid_inp, valid_vox, pnet_p, offr_p,
                                     ray_rgb=ray_rgb)
        assert (gev.cpu().long() == ev).all(), it                           # end voxel ids: exact
        assert (got.cpu() - ref).abs().max().item() <= TOL, it
        assert (got - cur).abs().max().item() > 1e-3                        # the iteration moved the points
        cur = got
    both, bev = hip(s1["pred_pos"], 2)
    assert torch.equal(both, cur) and torch.equal(bev, gev)
