"""Development aid: per-phase s_memtime counters of the training forward (LIDF_MODE_TRAIN of the rows
kernel, wavefront 0 of workgroup 0). Builds scripts/liblidf_prof.so with -DLIDF_PROFILE
(`python scripts/prof_phases_train.py --build` here, then run on the GPU box)."""
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
                    "-pragma-unroll-threshold=8000000", "-mllvm", "-amdgpu-mfma-vgpr-form", "-I",
                    os.path.join(ROOT, "include"), "-I", B.HERE, "-o", LIB] + src, check=True)
    if "--build" in sys.argv:
        sys.exit(0)
os.environ["LIDF_HIP_LIB"] = LIB
import ctypes as C  # noqa: E402
import torch  # noqa: E402
from util import make_module, orc, to_dev  # noqa: E402
from implicit_depth_amd import _lib  # noqa: E402
from implicit_depth_amd import decoders as _dec  # noqa: E402
from implicit_depth_amd.query import ray_features  # noqa: E402

dev = torch.device("cuda:0")
scene = orc.synthetic_scene(1, 240, 320, 8, seed=1235)
s = to_dev(scene, dev)
P, R, V = scene["P"], scene["R"], s["vox_feat"].shape[0]
L = _lib.lib()
rf = ray_features(s["feat_grid"], s["ray_dir"], s["ray_pix"], s["ray_bid"], 8, 4)
pe = torch.empty((P, 102), device=dev)
_lib.check(L.lidf_pe_rows_f32(_lib.ptr(s["pair_ray"]), _lib.ptr(s["pair_vox"]), _lib.ptr(s["pair_t"]),
                              _lib.ptr(s["ray_dir"]), None, 0, 8, P, _lib.ptr(pe), _lib.current_stream(dev)))
names = ["tile top: index loads, voxpart/raypart requests", "operand burst issued + waited", "raypart added",
         "layer 1 matrix instructions", "pass: u + first H1", "outputs", "pass: layer 2", "pass: layer 3 + tail"]
for kind in ("IMNET", "IEF"):
    mod = make_module(kind, scene["prob_p" if kind == "IMNET" else "off_p"], 385, dev).train()
    keep = []
    dec = _dec._decoder_struct(mod, keep)
    a = _lib.LidfQueryTrainArgs()
    a.n_pairs, a.n_rays, a.n_vox = P, R, V
    a.pair_off, a.pair_ray, a.pair_vox = s["pair_off"].data_ptr(), s["pair_ray"].data_ptr(), s["pair_vox"].data_ptr()
    a.pe, a.multires, a.multires_views = pe.data_ptr(), 8, 4
    a.vox_feat, a.rayfeat, a.dec = s["vox_feat"].data_ptr(), rf.data_ptr(), C.pointer(dec)
    npass = 2 if kind == "IEF" else 1
    act = torch.empty((L.lidf_query_decoder_act_floats(P, R, V, npass),), device=dev)
    wsb = L.lidf_query_decoder_workspace_bytes(P, R, V)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    out = torch.empty((P, 1), device=dev)
    for _ in range(2):
        _lib.check(L.lidf_query_decoder_forward_train_f32(C.byref(a), _lib.ptr(out), _lib.ptr(act), _lib.ptr(ws),
                                                          wsb, _lib.current_stream(dev)))
    torch.cuda.synchronize()
    pre = act[(V + R) * 256 + npass * P * 449:]
    t = pre[:16].view(torch.int64).cpu().tolist()
    per = ((P + 127) // 128 + 255) // 256
    tot = sum(t)
    print(kind, "tiles per wavefront", per)
    for n, v in zip(names, t):
        print("  %-50s %11d cycles %5.1f%%  per wave-tile %8.0f" % (n, v, 100.0 * v / max(tot, 1), v / per))
    mf = 416 + 654 * npass
    print("  total per wave-tile %.0f (matrix-instruction floor %d x 64 = %d)" % (tot / per, mf, mf * 64))
