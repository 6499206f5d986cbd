"""CPU: host logic — the C-ABI library loads and exports every symbol include/lidf_hip.h declares,
the drop-in modules keep the reference's constructor signatures and state-dict keys, CPU tensors
are refused (no CPU product path), and the frame sharding + depth all-gather work under gloo."""
import os
import re
import socket

import pytest
import torch
import torch.multiprocessing as mp

from util import ROOT


def header_functions():
    src = open(os.path.join(ROOT, "include", "lidf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lidf_[a-z0-9_]+)\s*\(", src)))


def header_abi_version():
    src = open(os.path.join(ROOT, "include", "lidf_hip.h")).read()
    return int(re.search(r"#define\s+LIDF_ABI_VERSION\s+(\d+)", src).group(1))


def test_stale_library_is_refused_at_load(tmp_path):
    """SURVEY 8b error convention: a liblidf_hip.so built for another ABI (the .so is a git-ignored
    artefact that travels outside history) must raise at load, before any struct is marshalled."""
    import subprocess
    import sys
    src = tmp_path / "stub.c"
    src.write_text("int lidf_version(void) { return 5; }\n")
    so = tmp_path / "libstub.so"
    subprocess.run(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)], check=True)
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from implicit_depth_amd import _lib\n"
            "try:\n    _lib.lib()\nexcept RuntimeError as e:\n"
            "    assert 'ABI 5' in str(e) and 'ABI %%d' %% _lib.ABI in str(e), e; print('refused')\n"
            "else:\n    raise SystemExit('a stale library was accepted')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LIDF_HIP_LIB=str(so)),
                       capture_output=True, text=True)
    assert r.returncode == 0 and "refused" in r.stdout, r.stdout + r.stderr
    # a library without the symbol at all (not ours) is refused the same way
    src.write_text("int other(void) { return 0; }\n")
    subprocess.run(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)], check=True)
    r = subprocess.run([sys.executable, "-c", code.replace("'ABI 5'", "'ABI None'")],
                       env=dict(os.environ, LIDF_HIP_LIB=str(so)), capture_output=True, text=True)
    assert r.returncode == 0 and "refused" in r.stdout, r.stdout + r.stderr


def test_library_exports_every_declared_symbol():
    from implicit_depth_amd import _lib
    names = header_functions()
    assert len(names) >= 17
    L = _lib.lib()  # loads without a GPU (no compute call is made here)
    for n in names:
        assert hasattr(L, n), n
        assert n in _lib.SIGNATURES, "ctypes signature missing for " + n
    assert set(_lib.SIGNATURES) == set(names)
    assert L.lidf_version() == _lib.ABI == header_abi_version()
    assert b"workspace" in L.lidf_strerror(-3)
    assert L.lidf_query_workspace_bytes(76800, 729, 0) > 76800 * 512 * 4
    assert (L.lidf_query_workspace_bytes(76800, 729, 32 * 240 * 320)
            >= L.lidf_query_workspace_bytes(76800, 729, 0) + 32 * 240 * 320 * 4)
    assert L.lidf_decoders_workspace_bytes(10, 385) > 0


def test_struct_layouts_match_header(tmp_path):
    """Every ctypes mirror in _lib.py against the C compiler's view of include/lidf_hip.h: size of
    the struct and offset of every field (gcc compiles a probe that prints them)."""
    import ctypes as C
    import subprocess
    from implicit_depth_amd import _lib
    structs = [n for n in dir(_lib) if n.startswith("Lidf") and isinstance(getattr(_lib, n), type)
               and issubclass(getattr(_lib, n), C.Structure)]
    assert {"LidfDecoder", "LidfQueryArgs", "LidfRefineArgs", "LidfPointNet", "LidfDecoderGrads",
            "LidfQueryTrainArgs"} <= set(structs)
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "lidf_hip.h"', 'int main(void){']
    for n in structs:
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (n, n))
        for f, _ in getattr(_lib, n)._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (n, f, n, f))
    lines.append("return 0;}")
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    seen = dict(ln.split() for ln in out.strip().splitlines())
    for n in structs:
        cls = getattr(_lib, n)
        assert int(seen[n]) == C.sizeof(cls), n
        for f, _ in cls._fields_:
            assert int(seen["%s.%s" % (n, f)]) == getattr(cls, f).offset, (n, f)


def test_torch_extension_shim_loads():
    """The pybind11 shim over the C ABI is built in-tree and imports without a GPU."""
    from implicit_depth_amd import torch_ext
    m = torch_ext.ext()
    from implicit_depth_amd import _lib
    assert m.abi_version() == _lib.ABI
    for fn in ("ray_aabb", "pcl_aabb", "compute_ray_aabb", "forward_decoders", "forward_query"):
        assert callable(getattr(m, fn))
    with pytest.raises(RuntimeError, match="CUDA"):   # CHECK_INPUT of the reference bindings: CUDA tensors only
        m.pcl_aabb(torch.zeros(1, 3), torch.zeros(1, 6), torch.zeros(1, dtype=torch.int32),
                   torch.zeros(1, dtype=torch.int32))


def test_modules_keep_reference_interface():
    from implicit_depth_amd import IEF, IMNet, get_embedder
    fn, dim = get_embedder(8)
    assert dim == 51 and callable(fn)
    ident, d3 = get_embedder(8, i=-1)
    assert d3 == 3 and isinstance(ident, torch.nn.Identity)
    assert get_embedder(4)[1] == 27
    m = IMNet(385, 1, gf_dim=64, use_sigmoid=False)
    assert sorted(m.state_dict()) == sorted(
        ["linear_%d.%s" % (i, k) for i in range(1, 5) for k in ("weight", "bias")])
    assert m.linear_1.weight.shape == (256, 385) and m.linear_4.weight.shape == (1, 64)
    e = IEF(torch.device("cpu"), 385, 1, gf_dim=64, n_iter=2)
    assert "offset_enc.weight" in e.state_dict() and e.linear_1.weight.shape == (256, 401)
    assert "init_offset" not in e.state_dict()  # plain attribute, as in the reference (:104)
    assert float(e.init_offset) == pytest.approx(0.001)
    assert float(m.linear_4.weight.mean()) != 0.0


def test_cpu_tensors_are_refused():
    from implicit_depth_amd import IEF, IMNet, get_embedder
    with torch.no_grad():
        with pytest.raises(RuntimeError):
            IMNet(385, 1)(torch.zeros(4, 385))
        with pytest.raises(RuntimeError):
            IEF(torch.device("cpu"), 385, 1, n_iter=2)(torch.zeros(4, 385))
        with pytest.raises(RuntimeError):
            get_embedder(8)[0](torch.zeros(4, 3))
    from implicit_depth_amd.extensions import pcl_aabb, ray_aabb
    with pytest.raises(RuntimeError):
        ray_aabb.forward(torch.zeros(4, 3), torch.zeros(2, 6), torch.zeros(4, dtype=torch.int32),
                         torch.zeros(2, dtype=torch.int32))
    with pytest.raises(RuntimeError):
        pcl_aabb.forward(torch.zeros(4, 3), torch.zeros(2, 6), torch.zeros(4, dtype=torch.int32),
                         torch.zeros(2, dtype=torch.int32))


def test_inference_entry_points_refuse_autograd():
    """lidf_query / lidf_refine / the pipeline functions detach their outputs: with autograd recording
    and anything that requires grad they raise instead of silently returning graph-less tensors
    (the check precedes every device check, so it runs without a GPU)."""
    from implicit_depth_amd import IEF, IMNet, PointNet2Stage, pipeline as pl
    from implicit_depth_amd.query import lidf_query, lidf_refine
    prob, off = IMNet(385, 1), IEF(torch.device("cpu"), 385, 1, n_iter=2)
    z = torch.zeros(1)
    with pytest.raises(RuntimeError, match="inference path.*prob_dec.linear_1.weight.*lidf_query_train"):
        lidf_query(z, z, z, z, z, z, z, z, z, prob, off)
    for m in (prob, off):
        m.requires_grad_(False)
    fg = torch.zeros(1, 32, 4, 4, requires_grad=True)
    with pytest.raises(RuntimeError, match="feat_grid requires grad"):
        lidf_query(z, z, z, z, z, z, z, fg, z, prob, off)
    zi = torch.zeros(1, dtype=torch.int32)
    with torch.no_grad(), pytest.raises(RuntimeError, match="CUDA"):   # past the check: the usual refusal
        lidf_query(torch.zeros(1, 3), zi, zi, zi, zi, zi, z, fg, z, prob, off)
    offr, pn = IEF(torch.device("cpu"), 334, 1, n_iter=2), PointNet2Stage(6, 128, 32)
    with pytest.raises(RuntimeError, match="inference path.*pnet_model"):
        lidf_refine(z, z, z, z, z, z, z, z, z, z, z, z, z, pn, offr)
    with pytest.raises(RuntimeError, match="inference path"):
        pl.lidf_forward({}, z, pn, IMNet(385, 1), off)
    with pytest.raises(RuntimeError, match="inference path"):
        pl.refine_forward({}, pn, offr)


def test_shard_frames():
    from implicit_depth_amd.dist import shard_frames
    for n in (0, 1, 7, 8, 32, 33):
        for world in (1, 2, 8):
            spans = [shard_frames(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_frames(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gloo_worker(rank, world, port, n_frames, q):
    import torch.distributed as dist
    from implicit_depth_amd.dist import all_gather_depth, all_gather_depth_ragged, shard_frames
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank,
                            world_size=world)
    try:
        h, w = 6, 8
        lo, hi = shard_frames(n_frames, world, rank)
        # "depth" of global frame f is the constant f + 1 (stands in for the per-rank query result)
        local = torch.stack([torch.full((h, w), float(f + 1)) for f in range(lo, hi)]) if hi > lo \
            else torch.zeros((0, h, w))
        if n_frames % world == 0:
            full = all_gather_depth(local)
        else:
            full = all_gather_depth_ragged(local, n_frames)
        ok = full.shape == (n_frames, h, w) and all(
            bool((full[f] == f + 1).all()) for f in range(n_frames))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def _gloo_rows_worker(rank, world, port, h, q):
    """Row shards of ONE depth map (dist.shard_rays / slice_rays / all_gather_depth_rows): every rank
    cuts its rays out of the same ragged scene, 'queries' them (depth = a function of the ray's own
    candidates, standing in for lidf_query) and the gathered map must equal the unsharded one."""
    import torch.distributed as dist
    from implicit_depth_amd.dist import all_gather_depth_rows, shard_rays, slice_rays
    from implicit_depth_amd.synthetic import synthetic_scene
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        w = 12
        scene = synthetic_scene(1, h, w, 5, seed=11, ragged=True)      # identical on every rank

        def fake_query(s):   # per ray: sum of t_leave over its own candidates + its flat pixel index
            d = torch.zeros(s["R"]).index_add_(0, s["pair_ray"].long(), s["pair_t"][:, 1])
            return d + s["ray_flat"].float()
        lo, hi = shard_rays(h, world, rank)
        mine = slice_rays(scene, lo * w, hi * w)
        ok = int(mine["pair_off"][-1]) == mine["P"] and (mine["pair_ray"] >= 0).all() and \
            (mine["pair_ray"] < mine["R"]).all()
        full = all_gather_depth_rows(fake_query(mine).reshape(hi - lo, w), h)
        ok = ok and full.shape == (h, w) and torch.equal(full, fake_query(scene).reshape(h, w))
        # dist.crop_rows: the feature map cut to the shard's rows + halo. A stand-in for the RoIAlign box
        # (pixel +- 2 rows, clamped on the image it is given) reads the same values from the cut map
        from implicit_depth_amd.dist import crop_rows

        def fake_roi(s, fg):   # per ray: sum of channel 0 over rows y-2 .. y+2 (clamped) of its column
            hh = fg.shape[2]
            x, y = s["ray_pix"][:, 0].long(), s["ray_pix"][:, 1].long()
            return sum(fg[0, 0, (y + d).clamp(0, hh - 1), x] for d in range(-2, 3))
        cut, fg, r0 = crop_rows(mine, scene["feat_grid"], lo, hi, 2)
        ok = ok and fg.shape[2] == min(h, hi + 2) - max(0, lo - 2) and r0 == max(0, lo - 2)
        ok = ok and torch.equal(fake_roi(cut, fg), fake_roi(mine, scene["feat_grid"]))
        ok = ok and torch.equal(cut["ray_flat"] + r0 * w, mine["ray_flat"])
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("h", [8, 9])
def test_ray_row_shards_gloo_world2(h):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_rows_worker, args=(r, 2, port, h, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def test_slice_rays_covers_the_list():
    from implicit_depth_amd.dist import shard_rays, slice_rays
    from implicit_depth_amd.synthetic import synthetic_scene
    scene = synthetic_scene(1, 10, 7, 6, seed=3, ragged=True)
    for world in (1, 3, 8):
        parts = [slice_rays(scene, *[v * 7 for v in shard_rays(10, world, r)]) for r in range(world)]
        assert sum(p["P"] for p in parts) == scene["P"] and sum(p["R"] for p in parts) == scene["R"]
        assert torch.equal(torch.cat([p["pair_vox"] for p in parts]), scene["pair_vox"])
        base = 0
        for p in parts:
            assert torch.equal(p["pair_ray"] + base, scene["pair_ray"][int(scene["pair_off"][base]):
                                                                      int(scene["pair_off"][base + p["R"]])])
            base += p["R"]


@pytest.mark.parametrize("n_frames", [4, 5])
def test_depth_all_gather_gloo_world2(n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def test_c_abi_argument_errors_without_gpu():
    """Status codes of the C ABI for malformed calls (checked before any HIP call is made, so this
    runs without a GPU): negative = lidf_status, lidf_strerror gives the text."""
    import ctypes as C
    from implicit_depth_amd import _lib
    L = _lib.lib()
    BAD, UNSUP, WS = -1, -2, -3
    assert L.lidf_query_f32(None, None) == BAD
    assert L.lidf_refine_f32(None, None) == BAD
    assert L.lidf_embed_f32(None, -1, 8, None, None) == BAD
    assert L.lidf_embed_f32(None, 0, 8, None, None) == 0            # empty input: nothing to do
    assert L.lidf_miss_ray_count(None, 0, -5, None, None, 0, None) == BAD
    assert L.lidf_miss_ray_count(None, 9, 10, None, None, 0, None) == BAD   # unknown mask dtype
    assert L.lidf_query_pack_f32(None, None, 8, 4, 0, None, 0, None) == BAD
    d = _lib.LidfDecoder()                                           # all-NULL weights
    assert L.lidf_query_pack_f32(C.byref(d), C.byref(d), 8, 4, 0, None, 0, None) == BAD
    assert L.lidf_pointnet_pack_f32(None, None, 0, None) == BAD
    assert L.lidf_pack_guard_bytes() >= 32
    assert L.lidf_query_pack_guarded_f32(None, None, 8, 4, 0, None, 0, None, None) == BAD
    assert L.lidf_pointnet_pack_guarded_f32(None, None, 0, None, None) == BAD
    assert L.lidf_refine_pack_guarded_f32(None, 8, 4, None, 0, None, None) == BAD
    q = _lib.LidfQueryArgs()
    q.n_rays = -1
    assert L.lidf_query_f32(C.byref(q), None) == BAD
    q.n_rays, q.multires = 4, 99
    assert L.lidf_query_f32(C.byref(q), None) in (BAD, UNSUP)
    assert L.lidf_exclusive_scan_i32(None, 10, None, None, 0, None) == BAD
    for code, word in ((0, "ok"), (BAD, "bad argument"), (UNSUP, "unsupported"), (WS, "workspace")):
        assert word in L.lidf_strerror(code).decode()
    assert L.lidf_query_pack_bytes() > 0 and L.lidf_pointnet_pack_bytes() > 0
    assert L.lidf_miss_ray_workspace_bytes(76800) > 0
    assert L.lidf_ray_features_workspace_bytes(1, 240, 320, 76800) >= 32 * 240 * 320 * 4


def test_generic_width_paths_have_no_cpu_route():
    """Widths other than the shipped ones run layer by layer on the device (generic.py): CPU tensors
    are refused there as everywhere, and the any-width entries check their arguments without a GPU."""
    import ctypes as C
    from implicit_depth_amd import IMNet, _lib
    from implicit_depth_amd.generic import linear_hip
    with pytest.raises(RuntimeError, match="CUDA"):
        linear_hip(torch.zeros(4, 8), torch.zeros(3, 8))
    m = IMNet(20, 2, 24)
    with pytest.raises(RuntimeError, match="CUDA"):
        m(torch.zeros(5, 20))
    L = _lib.lib()
    assert L.lidf_linear_workspace_bytes(0) == 0 and L.lidf_linear_workspace_bytes(385) > 0
    null = C.c_void_p(None)
    bad = L.lidf_linear_f32(null, 8, 4, 0, null, 8, null, 3, 0, 0.0, null, null, 0, null, 0, null, null, 0, null, 0, null)
    assert bad == -1                                            # k <= 0: LIDF_ERR_BAD_ARG
    assert L.lidf_linear_f32(null, 8, 0, 8, null, 8, null, 3, 0, 0.0, null, null, 0, null, 0, null, null, 0,
                             null, 0, null) == 0              # n == 0: nothing to do
    assert L.lidf_roi_align_f32(null, 1, 4, 8, 8, null, null, 0, 8, 3, null, 0, null) == 0
    assert L.lidf_roi_align_f32(null, 1, 4, 8, 8, null, null, 5, 8, 0, null, 0, null) == -1
    # the chain launch of gf_dim 32 / 64 / 128 (ABI 12) and the decoder pair's backward
    assert L.lidf_decoder_chain_workspace_bytes(32, 102) > 0 and L.lidf_decoder_chain_workspace_bytes(48, 102) == 0
    assert L.lidf_decoder_chain_workspace_bytes(128, 385) > L.lidf_decoder_chain_workspace_bytes(32, 385)
    d = _lib.LidfDecoder()
    args = (8, 4, 0, 0, null, null, null, null, null, 0, null, 0, null)   # ldx, k, w1_col0, n, tables, out, prepacked, workspace
    assert L.lidf_decoder_chain_f32(None, 32, 20, null, *args) == -1            # no decoder
    assert L.lidf_decoder_chain_f32(C.byref(d), 48, 20, null, *args) == -2      # a width without a chain: unsupported
    assert L.lidf_decoder_chain_f32(C.byref(d), 32, 20, null, *args) == -1      # NULL weights
    assert L.lidf_decoder_chain_f32(C.byref(d), 32, 3, null, *args) == -1       # k columns do not fit inp_dim
    one, two = L.lidf_decoder_train_workspace_bytes(1000, 385), L.lidf_decoder_pair_workspace_bytes(1000, 385)
    assert two > 2 * one and L.lidf_decoder_pair_workspace_offset(1000, 385, 0) == 0
    assert L.lidf_decoder_pair_workspace_offset(1000, 385, 1) >= one
    assert L.lidf_decoder_pair_backward_f32(null, 0, 385, 385, None, None, null, null, null, null, null, 385, None, None,
                                            null, 0, null) == -1


def test_cpu_tensors_refused_unless_composite_is_allowed(monkeypatch):
    """SURVEY 8b (CPU behaviour of the drop-in modules): no silent CPU path — a CPU tensor raises, as the
    reference's CHECK_CUDA does — but LIDF_ALLOW_CPU_COMPOSITE=1 routes CPU tensors (only) through the
    modules' own torch-op definition, for checkpoint conversion and callers' unit tests."""
    from implicit_depth_amd import IEF, IMNet, PointNet2Stage
    torch.manual_seed(0)
    x = torch.randn(5, 385)
    mods = (IMNet(385, 1, 64), IEF(torch.device("cpu"), 385, 1, 64, n_iter=2))
    pn = PointNet2Stage(6, 128, 32)
    pts, idx = torch.randn(9, 6), torch.tensor([0, 0, 1, 2, 2, 2, 1, 0, 2])
    monkeypatch.delenv("LIDF_ALLOW_CPU_COMPOSITE", raising=False)
    for m in mods:
        with pytest.raises(RuntimeError, match="LIDF_ALLOW_CPU_COMPOSITE"):
            m(x)
    with pytest.raises(RuntimeError, match="LIDF_ALLOW_CPU_COMPOSITE"):
        pn(pts, idx)
    monkeypatch.setenv("LIDF_ALLOW_CPU_COMPOSITE", "1")
    for m in mods:
        y = m(x)
        assert y.shape == (5, 1) and torch.equal(y, m.forward_composite(x))
    out = pn(pts, idx)
    assert out.shape == (3, 128) and torch.equal(out, pn.forward_composite(pts, idx, 3))
    # gradients flow through the composite definition (a caller's CPU unit test of a training step)
    mods[0](x).sum().backward()
    assert mods[0].linear_1.weight.grad is not None
