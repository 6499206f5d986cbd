"""The evaluation path of a batch of frames as ONE library call without a host round trip
(lidf_frame_f32 / pipeline.FrameRunner, device-side list lengths, capacity-sized buffers, HIP graph
replay) against the stepwise path (pipeline.lidf_forward + refine_forward, which reads every size on
the host and is itself checked against the oracle chain in test_e2e_gpu.py) and against the oracle."""
import pytest
import torch

from util import TOL, make_module, make_pointnet, orc

pytestmark = pytest.mark.gpu


def _dev(batch, dev):
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}


def _models(cuda):
    from implicit_depth_amd.synthetic import init_decoder_params
    pnet = make_pointnet(orc.init_pointnet(3, 1.5), cuda)
    pnet_r = make_pointnet(orc.init_pointnet(4, 1.5), cuda)
    prob = make_module("IMNET", init_decoder_params("IMNET", 385, 7, 5.0), 385, cuda)
    off = make_module("IEF", init_decoder_params("IEF", 385, 8, 5.0), 385, cuda)
    offr = make_module("IEF", init_decoder_params("IEF", 334, 9, 5.0), 334, cuda)
    return pnet, prob, off, pnet_r, offr


SAME = ("voxel_bound", "valid_v_rel_coord", "miss_ray_dir", "pair_t", "pnet_inp", "occ_voxel_feat",
        "pred_offset", "pred_prob_end", "pair_pred_pos", "pred_prob_end_softmax", "pred_pos", "rayfeat",
        "pred_depth", "pred_pos_refine", "pred_depth_refine")
SAME_INT = ("occ_vox_bid", "revidx", "valid_v_pid", "ray_bid", "ray_flat", "ray_pix", "pair_off", "pair_ray",
            "pair_vox", "max_pair_id", "end_voxel_id")


def _stepwise(batch, feat, models, opt, precision="f32", valid_idx=None):
    from implicit_depth_amd import pipeline as pl
    pnet, prob, off, pnet_r, offr = models
    with torch.no_grad():
        ok, dd = pl.lidf_forward(batch, feat, pnet, prob, off, opt, precision=precision, valid_idx=valid_idx)
        if ok:
            pl.refine_forward(dd, pnet_r, offr, opt, precision=precision)
    return ok, dd


def _compare(dd, ref):
    for k in SAME:
        assert torch.equal(dd[k], ref[k]), k                 # the same kernels on the same lists: bit-equal
    for k in SAME_INT:
        assert torch.equal(dd[k].long(), ref[k].long()), k


@pytest.mark.parametrize("shape,stride,use_all_pix,graph", [((1, 240, 320), 6, True, True),
                                                            ((1, 240, 320), None, True, False),
                                                            ((2, 48, 64), None, False, True),
                                                            ((4, 60, 80), 3, True, True),
                                                            ((4, 240, 320), 6, True, True),
                                                            ((4, 240, 320), None, True, False)])
def test_frame_equals_stepwise_path(cuda, shape, stride, use_all_pix, graph):
    from implicit_depth_amd import pipeline as pl
    from implicit_depth_amd.synthetic import synthetic_batch
    B, h, w = shape
    models = _models(cuda)
    opt = pl.LidfOptions(valid_stride=stride, refine_use_all_pix=use_all_pix)
    runner = pl.FrameRunner(B, h, w, cuda, models[0], models[1], models[2], opt, models[3], models[4])
    batch, feat = synthetic_batch(B, h, w, seed=77)
    batch, feat = _dev(batch, cuda), feat.to(cuda)
    with torch.no_grad():
        runner.load(batch, feat)
        if graph:
            runner.capture()
        runner.run()
    ok, dd = runner.result()
    ok_ref, ref = _stepwise(batch, feat, models, opt)
    assert ok and ok_ref
    c = dd["counts"]
    assert (c["R"], c["P"], c["V"]) == (ref["miss_ray_dir"].shape[0], ref["pair_ray"].shape[0],
                                        ref["voxel_bound"].shape[0])
    assert c["NV"] == ref["revidx"].shape[0] and c["NPN"] == c["NV"] + c["R"] and c["OVERFLOW"] == 0
    _compare(dd, ref)
    # a second, different batch through the same runner (and the same graph): nothing stale
    batch2, feat2 = synthetic_batch(B, h, w, seed=78, hole_frac=1.3)
    batch2, feat2 = _dev(batch2, cuda), feat2.to(cuda)
    with torch.no_grad():
        runner.run(batch2, feat2)
    ok2, dd2 = runner.result()
    ok_ref2, ref2 = _stepwise(batch2, feat2, models, opt)
    assert ok2 and ok_ref2 and dd2["counts"]["P"] != c["P"]
    _compare(dd2, ref2)
    m = runner.metrics(batch2)
    m_ref = pl.eval_metrics(ref2, "pred_depth_refine")
    for k in m_ref:
        assert float(m[k]) == float(m_ref[k]) or (m[k] != m[k] and m_ref[k] != m_ref[k]), k


def test_frame_vs_oracle_and_reference_dtypes(cuda):
    """The frame path straight against the oracle chain on a small frame (geometry exact, predictions
    within 1e-4), with the reference's int64 index tensors."""
    from implicit_depth_amd import pipeline as pl
    from implicit_depth_amd.synthetic import init_decoder_params, synthetic_batch
    B, h, w = 2, 48, 64
    batch, feat = synthetic_batch(B, h, w, seed=77)
    pnet_p = orc.init_pointnet(3, 1.5)
    prob_p, off_p = init_decoder_params("IMNET", 385, 7, 5.0), init_decoder_params("IEF", 385, 8, 5.0)
    ok_ref, ref = orc.lidf_forward(batch, feat, pnet_p, prob_p, off_p)
    assert ok_ref
    pnet = make_pointnet(pnet_p, cuda)
    prob, off = make_module("IMNET", prob_p, 385, cuda), make_module("IEF", off_p, 385, cuda)
    runner = pl.FrameRunner(B, h, w, cuda, pnet, prob, off)
    with torch.no_grad():
        runner.run(_dev(batch, cuda), feat.to(cuda))
    ok, dd = runner.result(reference_dtypes=True)
    assert ok and dd["miss_bid"].dtype == torch.int64 and "pred_pos_refine" not in dd
    assert (dd["voxel_bound"].cpu() == ref["voxel_bound"]).all()
    assert (dd["revidx"].cpu() == ref["revidx"]).all() and (dd["valid_v_pid"].cpu() == ref["valid_v_pid"]).all()
    assert (dd["valid_v_rel_coord"].cpu() == ref["valid_v_rel_coord"]).all()
    assert (dd["miss_flat_img_id"].cpu() == ref["miss_flat_img_id"]).all()
    assert (dd["pair_ray"].cpu().long() == ref["pair_ray"]).all()
    assert (dd["pair_vox"].cpu().long() == ref["pair_vox"]).all()
    assert (dd["occ_voxel_feat"].cpu() - ref["occ_voxel_feat"]).abs().max().item() <= 2e-5
    for k in ("pred_offset", "pred_prob_end", "pair_pred_pos"):
        assert (dd[k].cpu() - ref[k]).abs().max().item() <= TOL, k
    same = dd["max_pair_id"].cpu() == ref["max_pair_id"]
    assert (~same).sum().item() <= 2
    assert (dd["pred_depth"].cpu() - ref["pred_depth"]).abs().reshape(-1)[same].mean().item() <= TOL


def test_frame_pred_mask_early_exits_and_overflow(cuda):
    from implicit_depth_amd import pipeline as pl
    from implicit_depth_amd.synthetic import synthetic_batch
    B, h, w = 1, 24, 32
    models = _models(cuda)
    batch, feat = synthetic_batch(B, h, w, seed=5)
    b, feat = _dev(batch, cuda), feat.to(cuda)
    # mask_type 'pred': rays only where the predicted mask is set, valid points elsewhere
    opt = pl.LidfOptions(mask_type="pred")
    pm = (torch.rand(B, h, w, generator=torch.Generator().manual_seed(1)) < 0.3).float().to(cuda)
    runner = pl.FrameRunner(B, h, w, cuda, models[0], models[1], models[2], opt, models[3], models[4])
    with torch.no_grad():
        runner.run(b, feat, pm)
        ok, dd = runner.result()
        ok_ref, ref = pl.lidf_forward(b, feat, models[0], models[1], models[2], opt, pred_mask=pm)
        pl.refine_forward(ref, models[3], models[4], opt)
    assert ok and ok_ref and dd["counts"]["R"] == int(pm.sum().item())
    _compare(dd, ref)
    with torch.no_grad():
        # no miss ray (pipeline.py:686-687)
        runner.run(b, feat, torch.zeros(B, h, w, device=cuda))
        ok, dd = runner.result()
        assert not ok and dd["counts"]["R"] == 0 and dd["counts"]["P"] == 0
        # no occupied voxel (:671-672): every point outside the grid
        far = dict(b)
        far["xyz_corrupt"] = b["xyz_corrupt"] + 50.0
        runner.run(far, feat, pm)
        ok, dd = runner.result()
        assert not ok and dd["counts"]["V"] == 0 and dd["counts"]["P"] == 0 and dd["counts"]["NV"] == 0
        # and a good frame again through the same buffers
        runner.run(b, feat, pm)
        ok, dd = runner.result()
        assert ok
        _compare(dd, ref)
    # a pair list longer than its capacity is cut and flagged, never written out of bounds
    small = pl.FrameRunner(B, h, w, cuda, models[0], models[1], models[2], pl.LidfOptions(), max_pairs=100)
    with torch.no_grad():
        small.run(b, feat)
    assert small.counts()["OVERFLOW"] == 1 and small.counts()["P"] == 100
    with pytest.raises(RuntimeError, match="max_pairs"):
        small.result()
    with pytest.raises(RuntimeError, match="inference path"):
        small.run(b, feat)                                  # autograd recording: refused


def test_frames_pipelined_over_streams(cuda):
    """Evaluation loops run independent frames: several FrameRunners (own buffers, own packed-weight
    entries — nothing shared but the read-only modules and inputs) take the frames in turn on their own
    streams, so one frame's low-occupancy stretches are filled by its neighbour's kernels. Every frame's
    result must equal what a single runner on the default stream produces."""
    from implicit_depth_amd import _lib, pipeline as pl
    from implicit_depth_amd.synthetic import synthetic_batch
    B, h, w = 1, 120, 160
    models = _models(cuda)
    opt = pl.LidfOptions(valid_stride=2)
    frames = []
    for seed in (77, 78, 79, 80, 81, 82):
        batch, feat = synthetic_batch(B, h, w, seed=seed, hole_frac=1.0 + 0.1 * (seed % 3))
        frames.append((_dev(batch, cuda), feat.to(cuda)))
    keys = ("pred_pos", "pred_pos_refine", "pair_pred_pos", "pred_depth_refine", "occ_voxel_feat")
    ref = []
    single = pl.FrameRunner(B, h, w, cuda, models[0], models[1], models[2], opt, models[3], models[4])
    with torch.no_grad():
        for batch, feat in frames:
            single.run(batch, feat)
            ok, dd = single.result()
            assert ok
            ref.append({k: dd[k].clone() for k in keys})
    S = 3
    runners = [pl.FrameRunner(B, h, w, cuda, models[0], models[1], models[2], opt, models[3], models[4])
               for _ in range(S)]
    lanes = [torch.cuda.Stream(cuda) for _ in range(S)]
    torch.cuda.synchronize()
    got = [None] * len(frames)
    with torch.no_grad():
        for rnd in range(0, len(frames), S):
            for k in range(S):                               # enqueue S frames, one per stream, no sync between
                with torch.cuda.stream(lanes[k]):
                    runners[k].run(*frames[rnd + k])
            for k in range(S):
                with torch.cuda.stream(lanes[k]):
                    ok, dd = runners[k].result()
                    assert ok
                    got[rnd + k] = {key: dd[key].clone() for key in keys}
    torch.cuda.synchronize()
    for g, r in zip(got, ref):
        for k in keys:
            assert torch.equal(g[k], r[k]), k
    # every runner owns its packed weight streams and their fingerprints: nothing shared between streams,
    # nothing left in the per-module caches (a captured graph must not reference memory another call frees)
    blobs = {r.pack_blob.data_ptr() for r in runners + [single]} | {r.pack_guard.data_ptr() for r in runners}
    assert len(blobs) == 2 * S + 1
    assert all(m not in _lib.PACK_CACHE and m not in _lib.PACK_CACHE_REFINE for m in models)


@pytest.mark.parametrize("pos,rpos,rpnet", [("rel", "abs", "rel"), ("rel", "rel", "abs"), ("abs", "rel", "abs")])
def test_frame_position_types(cuda, pos, rpos, rpnet):
    """intersect_pos_type / refine.intersect_pos_type / refine.pnet_pos_type 'rel' and 'abs'
    (models/pipeline.py:355-360, :975-986, :1019-1023) through the sync-free call: bit-equal to the stepwise
    path, which test_e2e_gpu.py checks against the oracle."""
    from implicit_depth_amd import pipeline as pl
    from implicit_depth_amd.synthetic import synthetic_batch
    B, h, w = 2, 48, 64
    models = _models(cuda)
    opt = pl.LidfOptions(intersect_pos_type=pos, refine_intersect_pos_type=rpos, refine_pnet_pos_type=rpnet)
    batch, feat = synthetic_batch(B, h, w, seed=91)
    batch, feat = _dev(batch, cuda), feat.to(cuda)
    runner = pl.FrameRunner(B, h, w, cuda, models[0], models[1], models[2], opt, models[3], models[4])
    with torch.no_grad():
        runner.run(batch, feat)
    ok, dd = runner.result()
    ok_ref, ref = _stepwise(batch, feat, models, opt)
    assert ok and ok_ref
    _compare(dd, ref)
    # and the option does something: another result than the shipped combination
    if pos == "rel":
        ok0, ref0 = _stepwise(batch, feat, models, pl.LidfOptions())
        assert not torch.equal(ref0["pred_offset"], ref["pred_offset"])


@pytest.mark.parametrize("shape,graph", [((1, 240, 320), True), ((3, 37, 53), False)])
def test_frame_offsets_for_selected_pairs_only(cuda, shape, graph):
    """offsets='selected' (opt-in): offset_dec on the arg-max pair of every ray only. Everything the reference
    reads downstream of get_pred (pred_prob_end, softmax, max_pair_id, pred_pos — models/pipeline.py:453-454 —,
    depth, stage 2, the statistics) is bit-identical to the default; pred_offset / pair_pred_pos agree at the
    selected pairs; frame and stepwise paths agree with each other in this mode too."""
    from implicit_depth_amd import pipeline as pl
    from implicit_depth_amd.synthetic import synthetic_batch
    B, h, w = shape
    models = _models(cuda)
    opt = pl.LidfOptions(valid_stride=3)
    batch, feat = synthetic_batch(B, h, w, seed=83)
    batch, feat = _dev(batch, cuda), feat.to(cuda)
    full = pl.FrameRunner(B, h, w, cuda, *models[:3], opt, models[3], models[4])
    sel = pl.FrameRunner(B, h, w, cuda, *models[:3], opt, models[3], models[4], offsets="selected")
    with torch.no_grad():
        full.run(batch, feat)
        sel.load(batch, feat)
        if graph:
            sel.capture()
        sel.run()
    ok, a = full.result()
    ok2, b = sel.result()
    assert ok and ok2 and a["counts"] == b["counts"]
    for k in ("pred_prob_end", "pred_prob_end_softmax", "pred_pos", "pred_depth", "pred_pos_refine",
              "pred_depth_refine", "occ_voxel_feat", "rayfeat"):
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(a["max_pair_id"], b["max_pair_id"]) and torch.equal(a["end_voxel_id"], b["end_voxel_id"])
    P = a["counts"]["P"]
    m = a["max_pair_id"]
    m = m[m < P]
    assert m.numel() > 0
    assert torch.equal(a["pred_offset"][m], b["pred_offset"][m])
    assert torch.equal(a["pair_pred_pos"][m], b["pair_pred_pos"][m])
    rest = torch.ones(P, dtype=torch.bool, device=cuda)
    rest[m] = False
    assert torch.isnan(b["pred_offset"][rest]).all()          # never written: not mistaken for results
    ma, mb = full.metrics(batch), sel.metrics(batch)
    for k in ma:
        assert float(ma[k]) == float(mb[k]) or (ma[k] != ma[k] and mb[k] != mb[k]), k
    # the stepwise path in the same mode
    with torch.no_grad():
        ok3, c = pl.lidf_forward(batch, feat, models[0], models[1], models[2], opt, offsets="selected")
        pl.refine_forward(c, models[3], models[4], opt)
    assert ok3
    for k in ("pred_prob_end", "pred_pos", "pred_depth", "pred_pos_refine"):
        assert torch.equal(b[k], c[k]), k
    assert torch.equal(b["pred_offset"][m], c["pred_offset"][m])


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("offsets", ["all", "selected"])
def test_frame_with_a_side_stream(cuda, graph, offsets):
    """side_stream=True (LidfFrameArgs.aux_stream / ev_fork / ev_join): the weight-stream guard, the box sums and
    the per-ray features run beside the head, the pairs and the PointNet. Same kernels on the same data: every
    output is bit-identical to the one-stream call, eager and through a captured graph, over several frames
    with different inputs on the same runner (FrameRunner's default: side stream for eager calls only)."""
    from implicit_depth_amd import pipeline as pl
    from implicit_depth_amd.synthetic import synthetic_batch
    B, h, w = 2, 60, 80
    models = _models(cuda)
    opt = pl.LidfOptions(valid_stride=2)
    one = pl.FrameRunner(B, h, w, cuda, *models[:3], opt, models[3], models[4], offsets=offsets,
                         side_stream=False)
    two = pl.FrameRunner(B, h, w, cuda, *models[:3], opt, models[3], models[4], offsets=offsets,
                         side_stream=True)
    for it, seed in enumerate((5, 6, 7)):
        batch, feat = synthetic_batch(B, h, w, seed=seed)
        batch, feat = _dev(batch, cuda), feat.to(cuda)
        with torch.no_grad():
            one.run(batch, feat)
            two.load(batch, feat)
            if graph and it == 0:
                two.capture()
            two.run()
        ok, a = one.result()
        ok2, b = two.result()
        assert ok and ok2 and a["counts"] == b["counts"]
        P = a["counts"]["P"]
        for k in ("pred_prob_end", "pred_prob_end_softmax", "pred_pos", "pred_depth", "pred_pos_refine",
                  "pred_depth_refine", "occ_voxel_feat", "rayfeat", "max_pair_id", "end_voxel_id"):
            assert torch.equal(a[k], b[k]), (k, seed)
        for k in ("pair_ray", "pair_vox", "pair_t"):
            assert torch.equal(a[k][:P], b[k][:P]), (k, seed)
        if offsets == "all":
            assert torch.equal(a["pred_offset"][:P], b["pred_offset"][:P])


@pytest.mark.parametrize("stage", [1, 2, 3, 4])
def test_side_stream_is_joined_on_every_error_exit(cuda, stage, monkeypatch):
    """A frame that fails between its fork and its joins (LidfFrameArgs.fail_after: the return a failed launch of
    that stage would take — stage 1: second fork just recorded, per-ray features queued on the side stream and
    no join recorded; 3: join recorded but not awaited) must leave nothing of itself running on the side
    stream beside the caller's next work: lidf_frame_f32 records ev_join behind the side stream's queue and
    makes the caller's stream wait for it on every non-zero return. Checked where it shows: the failed frame's
    side launches write `rayfeat` / the box sums of ITS inputs; the good frame that follows on the same runner
    (other inputs, same buffers) must be bit-identical to a fresh runner's — repeatedly, since a race is a
    matter of timing."""
    from implicit_depth_amd import pipeline as pl
    from implicit_depth_amd.synthetic import synthetic_batch
    B, h, w = 1, 240, 320
    models = _models(cuda)
    opt = pl.LidfOptions(valid_stride=4)
    bad_in = synthetic_batch(B, h, w, seed=31, hole_frac=1.4)
    good_in = synthetic_batch(B, h, w, seed=32)
    bad, bad_feat = _dev(bad_in[0], cuda), bad_in[1].to(cuda)
    good, good_feat = _dev(good_in[0], cuda), good_in[1].to(cuda)
    fresh = pl.FrameRunner(B, h, w, cuda, *models[:3], opt, models[3], models[4], side_stream=True)
    with torch.no_grad():
        fresh.run(good, good_feat)
    ok, ref = fresh.result()
    assert ok
    runner = pl.FrameRunner(B, h, w, cuda, *models[:3], opt, models[3], models[4], side_stream=True)
    monkeypatch.setenv("LIDF_TEST_FAULTS", "1")    # (the hook is ignored in a process without it)
    for rep in range(4):
        runner._fail_after = stage
        with torch.no_grad(), pytest.raises(RuntimeError):
            runner.run(bad, bad_feat)
        runner._fail_after = 0
        with torch.no_grad():
            runner.run(good, good_feat)     # no sync in between: ordered by the streams alone
        ok2, dd = runner.result()
        assert ok2 and dd["counts"] == ref["counts"]
        for k in ("rayfeat", "pred_prob_end", "pred_offset", "pred_pos", "pred_depth", "pred_pos_refine",
                  "pred_depth_refine", "occ_voxel_feat", "max_pair_id", "end_voxel_id"):
            assert torch.equal(dd[k], ref[k]), (k, stage, rep)


def test_fault_hook_is_ignored_without_the_environment_switch(cuda, monkeypatch):
    """LidfFrameArgs.fail_after only acts in a process with LIDF_TEST_FAULTS=1: a caller that left the trailing
    field of the grown struct uninitialised gets its frame, not a spurious LIDF_ERR_HIP."""
    from implicit_depth_amd import pipeline as pl
    from implicit_depth_amd.synthetic import synthetic_batch
    B, h, w = 1, 48, 64
    models = _models(cuda)
    opt = pl.LidfOptions(valid_stride=2)
    batch, feat = synthetic_batch(B, h, w, seed=33)
    batch, feat = _dev(batch, cuda), feat.to(cuda)
    monkeypatch.delenv("LIDF_TEST_FAULTS", raising=False)
    ref = pl.FrameRunner(B, h, w, cuda, *models[:3], opt, models[3], models[4])
    runner = pl.FrameRunner(B, h, w, cuda, *models[:3], opt, models[3], models[4])
    runner._fail_after = 3
    with torch.no_grad():
        ref.run(batch, feat)
        runner.run(batch, feat)
    (ok, a), (ok2, b) = ref.result(), runner.result()
    assert ok and ok2 and a["counts"] == b["counts"] and torch.equal(a["pred_depth_refine"], b["pred_depth_refine"])


@pytest.mark.parametrize("graph", [False, True])
def test_frame_weight_streams_follow_parameter_updates(cuda, graph):
    """The runner's own packed streams (one fingerprint launch over every module per frame): in-place
    updates, `p.data` writes the version counter misses, load_state_dict — picked up by the next frame,
    eager and through a captured graph; guard_every > 1 trusts the blob between checks until the N-th
    frame or invalidate(); an unchanged frame re-packs nothing (the guards' verdicts read back)."""
    from implicit_depth_amd import pipeline as pl
    from implicit_depth_amd.synthetic import synthetic_batch
    B, h, w = 1, 48, 64
    models = _models(cuda)
    opt = pl.LidfOptions()
    batch, feat = synthetic_batch(B, h, w, seed=77)
    batch, feat = _dev(batch, cuda), feat.to(cuda)
    keys = ("pred_offset", "pred_prob_end", "occ_voxel_feat", "pred_pos_refine")

    def frame(r):
        with torch.no_grad():
            r.run(batch, feat)
        ok, dd = r.result()
        assert ok
        return {k: dd[k].clone() for k in keys}

    def fresh():
        ok, ref = _stepwise(batch, feat, models, opt)
        assert ok
        return {k: ref[k].clone() for k in keys}

    def dirty(r):   # verdicts of the four guards (query, pnet, pnet_refine, off_refine): int32 at byte 20
        return r.pack_guard.view(torch.int32).reshape(4, 16)[:, 5].tolist()

    runner = pl.FrameRunner(B, h, w, cuda, *models[:3], opt, models[3], models[4])
    if graph:
        with torch.no_grad():
            runner.load(batch, feat)
            runner.capture()
    a = frame(runner)
    ref = fresh()
    assert all(torch.equal(a[k], ref[k]) for k in keys)
    a2 = frame(runner)
    assert dirty(runner) == [0, 0, 0, 0] and all(torch.equal(a[k], a2[k]) for k in keys)   # nothing re-packed
    # every way a parameter can change, one module at a time
    edits = [
        (0, lambda: models[1].linear_2.weight.data.mul_(1.25)),                  # prob_dec, version untouched
        (0, lambda: models[2].offset_enc.bias.data.add_(0.01)),                  # offset_dec
        (1, lambda: models[0].point_lin3.weight.mul_(0.9)),                      # PointNet, in place
        (2, lambda: models[3].load_state_dict({k: v * 1.1 for k, v in models[3].state_dict().items()})),
        (3, lambda: models[4].linear_1.weight.data.copy_(models[4].linear_1.weight.data * 0.95)),
    ]
    for grp, edit in edits:
        with torch.no_grad():
            edit()
        b = frame(runner)
        d = dirty(runner)
        assert d[grp] == 1 and sum(d) == 1, (grp, d)
        ref = fresh()
        assert all(torch.equal(b[k], ref[k]) for k in keys), grp
        assert not all(torch.equal(b[k], a[k]) for k in keys)
        a = b
    # guard_every = 3: frames 2 and 3 after a check trust the blob, the 4th validates again
    lazy = pl.FrameRunner(B, h, w, cuda, *models[:3], opt, models[3], models[4], guard_every=3)
    if graph:
        with torch.no_grad():
            lazy.load(batch, feat)
            lazy.capture()
        assert lazy.graph_trusted is not None
    c0 = frame(lazy)
    assert all(torch.equal(c0[k], a[k]) for k in keys)
    with torch.no_grad():
        models[1].linear_3.bias.data.add_(0.05)
    c1 = frame(lazy)                                   # trusted: still the old weights
    assert all(torch.equal(c1[k], c0[k]) for k in keys)
    lazy.invalidate()
    c2 = frame(lazy)
    ref = fresh()
    assert all(torch.equal(c2[k], ref[k]) for k in keys) and not torch.equal(c2["pred_prob_end"], c1["pred_prob_end"])
    with torch.no_grad():
        models[1].linear_3.bias.data.sub_(0.02)
    seen = [frame(lazy) for _ in range(3)]             # frames 2, 3 trusted; frame 4 validates
    assert torch.equal(seen[0]["pred_prob_end"], c2["pred_prob_end"])
    assert torch.equal(seen[1]["pred_prob_end"], c2["pred_prob_end"])
    ref = fresh()
    assert all(torch.equal(seen[2][k], ref[k]) for k in keys)


@pytest.mark.parametrize("seed", range(10))
def test_frame_fuzz_against_stepwise(cuda, seed):
    """Randomised shapes / options of the frame path against the stepwise path (bit-equal): batch size,
    image size (not a multiple of the 1024-pixel workgroups), valid stride, mask type with random
    predicted masks, use_all_pix, hole sizes (down to no hole at all: no invalid pixel), pair capacity
    close to the need."""
    import random
    from implicit_depth_amd import pipeline as pl
    from implicit_depth_amd.synthetic import synthetic_batch
    rnd = random.Random(1000 + seed)
    B = rnd.choice([1, 1, 2, 3, 5])
    h, w = rnd.choice([(24, 32), (37, 53), (48, 64), (60, 81), (96, 128)])
    stride = rnd.choice([None, 2, 3, 7])
    mask_type = rnd.choice(["all", "all", "pred"])
    use_all = rnd.choice([True, False])
    hole = rnd.choice([0.0, 0.6, 1.0, 1.6])
    models = _models(cuda)
    opt = pl.LidfOptions(valid_stride=stride, refine_use_all_pix=use_all, mask_type=mask_type)
    batch, feat = synthetic_batch(B, h, w, seed=200 + seed, hole_frac=max(hole, 1e-3))
    batch, feat = _dev(batch, cuda), feat.to(cuda)
    pm = None
    if mask_type == "pred":
        g = torch.Generator().manual_seed(seed)
        pm = (torch.rand(B, h, w, generator=g) < rnd.choice([0.05, 0.4, 1.0])).float().to(cuda)
    with torch.no_grad():
        ok_ref, ref = pl.lidf_forward(batch, feat, models[0], models[1], models[2], opt, pred_mask=pm)
        if ok_ref:
            pl.refine_forward(ref, models[3], models[4], opt)
    need = ref["pair_ray"].shape[0] if ok_ref else 0
    runner = pl.FrameRunner(B, h, w, cuda, models[0], models[1], models[2], opt, models[3], models[4],
                            max_pairs=max(need + rnd.choice([0, 1, 100]), 1))
    with torch.no_grad():
        runner.run(batch, feat, pm)
    ok, dd = runner.result()
    assert ok == ok_ref, (B, h, w, stride, mask_type, use_all, hole)
    if ok:
        _compare(dd, ref)
    else:   # the same early exit: the counts say which
        c = dd["counts"]
        assert c["V"] == ref["voxel_bound"].shape[0]
        if c["V"] > 0:
            assert c["R"] == ref.get("total_miss_sample_num", 0)


@pytest.mark.parametrize("shape,graph", [((1, 120, 160), False), ((3, 48, 64), True)])
def test_frame_split_f16_equals_stepwise(cuda, shape, graph):
    """precision = "f16x3" through the frame path (split-f16 per-point kernel, per-ray rows kernel and
    stage-2 IEF with device-side counts) against the stepwise path at the same precision (bit-equal) and
    against the f32 frame (within the f16x3 accuracy)."""
    from implicit_depth_amd import pipeline as pl
    from implicit_depth_amd.synthetic import synthetic_batch
    B, h, w = shape
    models = _models(cuda)
    opt = pl.LidfOptions(valid_stride=2)
    batch, feat = synthetic_batch(B, h, w, seed=91)
    batch, feat = _dev(batch, cuda), feat.to(cuda)
    runner = pl.FrameRunner(B, h, w, cuda, models[0], models[1], models[2], opt, models[3], models[4],
                            precision="f16x3")
    with torch.no_grad():
        runner.load(batch, feat)
        if graph:
            runner.capture()
        runner.run()
    ok, dd = runner.result()
    ok_ref, ref = _stepwise(batch, feat, models, opt, precision="f16x3")
    assert ok and ok_ref
    _compare(dd, ref)
    ok32, d32 = _stepwise(batch, feat, models, opt)
    for k in ("pred_offset", "pred_prob_end"):
        assert (dd[k] - d32[k]).abs().max().item() <= 2e-5, k
    same = dd["max_pair_id"] == d32["max_pair_id"]          # (a float-noise tie may pick another pair)
    assert (~same).sum().item() <= 2
    # (two opt-in split-f16 refine iterations on top of a split-f16 stage 1, random weights: each stage is
    # within 2e-5 of f32 — above —, the chained positions within 2e-4; parity of the f16x3 path against the
    # oracle has its own tests, tests/test_split_f16_gpu.py)
    assert (dd["pred_pos_refine"] - d32["pred_pos_refine"])[same].abs().max().item() <= 2 * TOL


def _sampled_valid_idx(batch, n_per_image, seed, B, h, w):
    """[B * n, 2] (image, flat pixel) int64 as LIDF.get_valid_points keeps it with valid_sample_num != -1:
    per image n entries drawn from its valid pixels, in shuffled order, with repeats when the image has
    fewer than n valid pixels (utils/point_utils.py:98-106)."""
    g = torch.Generator().manual_seed(seed)
    valid = (batch["depth_corrupt"].reshape(B, h * w) != 0).cpu()
    out = []
    for b in range(B):
        pix = torch.nonzero(valid[b]).flatten()
        if pix.numel() >= n_per_image:
            sel = pix[torch.randperm(pix.numel(), generator=g)[:n_per_image]]
        else:
            extra = pix[torch.randint(0, pix.numel(), (n_per_image - pix.numel(),), generator=g)]
            sel = torch.cat([pix, extra])
        out.append(torch.stack([torch.full_like(sel, b), sel], 1))
    return torch.cat(out, 0)


@pytest.mark.parametrize("shape,n,graph", [((1, 240, 320), 10000, True), ((2, 48, 64), 700, False),
                                           ((3, 24, 32), 700, True)])
def test_frame_with_listed_valid_points(cuda, shape, n, graph):
    """load(valid_idx=): the valid points as the reference's get_valid_points hands them on when
    grid.valid_sample_num != -1 (a sampled, shuffled list with repeats) — the frame call against the
    stepwise path fed the same list, bit for bit, through a graph replay and a second batch."""
    from implicit_depth_amd import pipeline as pl
    from implicit_depth_amd.synthetic import synthetic_batch
    B, h, w = shape
    models = _models(cuda)
    opt = pl.LidfOptions()
    runner = pl.FrameRunner(B, h, w, cuda, models[0], models[1], models[2], opt, models[3], models[4])
    for k, seed in enumerate((31, 32)):
        batch, feat = synthetic_batch(B, h, w, seed=seed, hole_frac=1.0 + 0.2 * k)
        vidx = _sampled_valid_idx(batch, n, 5 + seed, B, h, w)      # (3, 24, 32): more than the valid pixels -> repeats
        batch, feat = _dev(batch, cuda), feat.to(cuda)
        with torch.no_grad():
            runner.load(batch, feat, valid_idx=vidx.to(cuda) if k else vidx)   # host or device indices
            if graph and k == 0:
                runner.capture()
            runner.run()
        ok, dd = runner.result()
        ok_ref, ref = _stepwise(batch, feat, models, opt, valid_idx=vidx.to(cuda))
        assert ok and ok_ref
        c = dd["counts"]
        assert c["NVS"] == B * n and c["NV"] == ref["revidx"].shape[0] and c["V"] == ref["voxel_bound"].shape[0]
        assert torch.equal(dd["valid_flat_img_id"].long(), vidx[:, 1].to(cuda))
        assert torch.equal(dd["valid_xyz"], ref["valid_xyz"]) and torch.equal(dd["valid_rgb"], ref["valid_rgb"])
        _compare(dd, ref)
    if graph:   # the captured graph replays its own list length
        with pytest.raises(RuntimeError):
            runner.load(batch, feat, valid_idx=vidx[:-1])
    with pytest.raises(RuntimeError):
        pl.FrameRunner(B, h, w, cuda, models[0], models[1], models[2], opt).load(
            batch, feat, valid_idx=torch.zeros((B * h * w + 1, 2), dtype=torch.int64))


def test_frame_pipeline_helper(cuda):
    """pipeline.FramePipeline: frames submitted back to back over 3 streams come back in order, each
    equal to what a single runner produces."""
    from implicit_depth_amd import pipeline as pl
    from implicit_depth_amd.synthetic import synthetic_batch
    B, h, w = 1, 96, 128
    models = _models(cuda)
    opt = pl.LidfOptions(valid_stride=2)
    frames = []
    for seed in range(60, 67):                                  # 7 frames over 3 slots
        batch, feat = synthetic_batch(B, h, w, seed=seed, hole_frac=1.0 + 0.1 * (seed % 4))
        frames.append((_dev(batch, cuda), feat.to(cuda)))
    keys = ("pred_pos_refine", "pred_depth_refine", "pair_pred_pos")
    single = pl.FrameRunner(B, h, w, cuda, *models[:3], opt, models[3], models[4])
    ref = []
    with torch.no_grad():
        for batch, feat in frames:
            single.run(batch, feat)
            ok, dd = single.result()
            ref.append(({k: dd[k].clone() for k in keys}, {k: float(v) for k, v in single.metrics(batch).items()}))
    pipe = pl.FramePipeline(3, B, h, w, cuda, *models[:3], opt, models[3], models[4])
    got = []

    def take():
        ok, dd, m = pipe.collect()
        assert ok
        got.append(({k: dd[k].clone() for k in keys}, {k: float(v) for k, v in m.items()}))
    for i, (batch, feat) in enumerate(frames):
        assert pipe.full == (i >= 3)
        if pipe.full:
            take()
        pipe.submit(batch, feat)
    with pytest.raises(RuntimeError):
        pipe.submit(*frames[0])                                  # 3 in flight
    while pipe.pending:
        take()
    assert len(got) == len(frames) and not pipe.pending
    for (g, gm), (r, rm) in zip(got, ref):
        for k in keys:
            assert torch.equal(g[k], r[k]), k
        for k in rm:
            assert gm[k] == rm[k] or (gm[k] != gm[k] and rm[k] != rm[k]), k
