#!/usr/bin/env python3
"""bench.py — LIDF per-point implicit-depth query throughput on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is
launched under torch.distributed.run (one rank per GPU, RCCL). A "step" is one pass of the fused
query (lidf_query_f32) over one synthetic 240x320 frame with 64 candidates per ray resident in HBM
on every rank (BASELINE.json configs[1]; weak scaling: each rank owns one frame), followed — for
N>1 — by the RCCL all-gather of the per-rank depth maps (the only collective on the path).
Rank 0 prints ONE JSON line with the metric, the MFMA roofline of the dominant kernel (HIP events
recorded by the library around lidf_points_fused_kernel on the launch stream) and, at N=1, the CPU
baseline (the oracle port on the host cores over the same frame in 614,400-point slabs, ~20 s) and
the kernel's average duration by rocprofv3 --kernel-trace of this same command, run as a child
process after the timed region (roofline.kernel_ms_rocprof_live; --no-rocprof to skip).
`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

F_ALG = 853952.0       # FLOP per point, reference formulation (SURVEY.md §8d / BASELINE.md §2)
# MFMA FLOP actually issued per point by lidf_points_kernel: per 32-point wave-tile
# 2 nets x (6*8+3) layer-1 k-steps x 8 tiles + 3 passes x (8 u + 129*4 layer-2 + 65*2 layer-3)
# v_mfma_f32_32x32x2 instructions of 4096 FLOP each.
F_EXEC = (2 * 51 * 8 + 3 * (8 + 129 * 4 + 65 * 2)) * 4096 / 32.0  # = 355,584
# per net (one launch of the per-point kernel with ONE net, lidf_query(offsets="selected")): layer 1 + n passes
F_EXEC_PROB = (51 * 8 + 1 * (8 + 129 * 4 + 65 * 2)) * 4096 / 32.0   # = 135,936 per pair   (prob_dec, IMNet)
F_EXEC_OFF = (51 * 8 + 2 * (8 + 129 * 4 + 65 * 2)) * 4096 / 32.0    # = 219,648 per pair   (offset_dec, IEF n_iter 2)
PEAK_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32 MFMA = f32 vector peak
# split-f16 kernel (lidf_points_h.hip): per wave-tile 3 passes x 406 v_mfma_f32_32x32x16_f16
# (8 tiles x (18 embedding + 2 mixed xyz/ray/IEF + 24 layer-2) + 6 bias + 48 layer-3) of 32,768 FLOP
F_EXEC_H = 3 * 406 * 32768 / 32.0  # = 1,247,232
PEAK_F16_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak (sustained under the power
#                           cap with random operands: ~1.4-1.6 PFLOP/s, scripts/mfma_power_ubench.hip)
# stage-2 decoder, issued FLOP per ray: on 16 x 16 x 4 sub-tiles (lidf_ief16.hip) / on 32 x 32 x 2 tiles (rounds 2-3)
F_IEF16 = (4 * 16 * 4 + 2 * (12 + 16 + 512 + 128)) * 2048 / 16.0   # = 203,776
F_IEF32 = (7 * 8 * 4 + 2 * 654) * 4096 / 32.0                      # = 196,096
DTYPE_F16X3 = "f16x3 (f32 operands split into two f16 pieces, 3 products per term, f32 accumulate)"


# The contract is ONE JSON line on stdout. Libraries write there too (RCCL prints a version banner
# when its communicator comes up on some boxes), so file descriptor 1 is pointed at stderr for the
# whole run and the record goes out through a duplicate of the original stdout.
_REAL_STDOUT = None


def claim_stdout():
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        try:
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            _REAL_STDOUT = saved
        except OSError:   # no usable stderr: leave stdout as it is
            _REAL_STDOUT = None


def emit(record):
    line = (json.dumps(record) + "\n").encode()
    sys.stdout.flush()
    if _REAL_STDOUT is None:
        os.write(1, line)
    else:
        os.write(_REAL_STDOUT, line)


class HipEvents:
    """Minimal hipEvent wrapper over libamdhip64 (torch.cuda.Event hides its handle until it has
    been recorded; the library needs raw hipEvent_t values to record on the launch stream)."""

    def __init__(self):
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        self.hip.hipEventSynchronize.argtypes = [C.c_void_p]
        self.hip.hipEventDestroy.argtypes = [C.c_void_p]

    def create(self):
        e = C.c_void_p()
        assert self.hip.hipEventCreate(C.byref(e)) == 0
        return e

    def elapsed_ms(self, a, b):
        assert self.hip.hipEventSynchronize(b) == 0
        ms = C.c_float()
        assert self.hip.hipEventElapsedTime(C.byref(ms), a, b) == 0
        return ms.value


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota
    (the GPU box exposes 256 hardware threads but the container quota is what bounds torch)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


SLAB_POINTS = 614400   # the CPU baseline processes P in 614,400-point slabs


def cpu_baseline(scene, hip_out=None, reps=3):
    """The oracle port on the host cores, as BASELINE.md §3 specifies: the same synthetic frame,
    P processed in 614,400-point slabs (whole image rows, so the per-ray reduction stays inside a
    slab), two numbers — full query (ROI + PE + gathers + decoders + per-ray softmax / arg-max /
    select) and decoder only ([slab,385] rows already materialised).
      full query:   1 warm-up slab, then the WHOLE first frame once, slab by slab (its outputs are
                    kept: they are the parity reference of the bench line), then slab 0 twice more
                    -> whole-frame rate and best-of-3 slab rate
      decoder only: 1 warm-up + best of `reps` on slab 0
    Returns (cpu_baseline dict, oracle outputs of the first frame)."""
    from oracle import lidf_oracle as orc  # the checker, timed as the CPU baseline (port)
    cores = usable_cores()
    torch.set_num_threads(cores)
    h, w, N = scene["h"], scene["w"], scene["N"]
    rows_per_slab = max(1, SLAB_POINTS // (w * N))
    R1, P1 = h * w, h * w * N                      # first frame only

    def run(r0, r1):
        p0, p1 = r0 * N, r1 * N
        t0 = time.perf_counter()
        o = orc.query(scene["ray_dir"][r0:r1], scene["ray_pix"][r0:r1], scene["ray_bid"][r0:r1],
                      scene["pair_ray"][p0:p1].long() - r0, scene["pair_vox"][p0:p1].long(),
                      scene["pair_t"][p0:p1], None, scene["feat_grid"], scene["vox_feat"],
                      scene["prob_p"], scene["off_p"], fast_roi=True, chunk=SLAB_POINTS)
        return o, time.perf_counter() - t0

    slab_r = rows_per_slab * w
    run(0, min(slab_r, R1))                                   # warm-up (thread pool, allocator)
    keys = ("pred_offset", "pred_prob_end", "pair_pred_pos", "pred_pos", "max_pair_id")
    parts, t_full, t_slab0 = [], 0.0, []
    for r0 in range(0, R1, slab_r):
        o, t = run(r0, min(r0 + slab_r, R1))
        o["max_pair_id"] = o["max_pair_id"] + r0 * N           # back to frame-wide pair indices
        parts.append({k: o[k] for k in keys})
        t_full += t
        if r0 == 0:
            t_slab0.append(t)
    for _ in range(reps - 1):
        t_slab0.append(run(0, min(slab_r, R1))[1])
    ref = {k: torch.cat([p[k] for p in parts], 0) for k in keys}
    n_slab = min(slab_r, R1) * N
    # decoder only on a materialised slab (the reference's own boundary, pipeline.py:434-435)
    inp, _, _ = orc.build_inp_embed(scene["ray_dir"][:slab_r], scene["ray_pix"][:slab_r],
                                    scene["ray_bid"][:slab_r], scene["pair_ray"][:n_slab].long(),
                                    scene["pair_vox"][:n_slab].long(), scene["pair_t"][:n_slab],
                                    scene["feat_grid"], scene["vox_feat"], 8, 4)
    t_dec = []
    for i in range(reps + 1):
        t0 = time.perf_counter()
        orc.ief_forward(scene["off_p"], inp, 2)
        orc.imnet_forward(scene["prob_p"], inp)
        if i:
            t_dec.append(time.perf_counter() - t0)
    del inp
    res = {
        "value": round(P1 / t_full / 1e6, 4), "unit": "Mpoints/s", "cores": cores, "kind": "port",
        "full_query": {"whole_frame": round(P1 / t_full / 1e6, 4),
                       "best_slab_of_%d" % reps: round(n_slab / min(t_slab0) / 1e6, 4)},
        "decoder_only": {"best_slab_of_%d" % reps: round(n_slab / min(t_dec) / 1e6, 4)},
        "slab_points": n_slab,
        "sample": "oracle/lidf_oracle.py (torch CPU ops, %d threads): full query over the whole "
                  "%dx%dx%d frame (%d points) in %d-point slabs, %.1f s; decoder only "
                  "(IMNet + IEF n_iter=2 on [%d,385] rows) best of %d, %.1f s each"
                  % (cores, h, w, N, P1, n_slab, t_full, n_slab, reps, min(t_dec)),
    }
    return res, ref


def parity_record(got, ref, scene, depth):
    """BASELINE metric, second half: HIP outputs vs the oracle's on the identical first frame."""
    N, hw = scene["N"], scene["h"] * scene["w"]
    P1 = hw * N
    mx = {}
    for k in ("pred_offset", "pred_prob_end", "pair_pred_pos"):
        mx[k] = float((got[k][:P1].cpu() - ref[k]).abs().max())
    mx["pred_pos"] = float((got["pred_pos"][:hw].cpu() - ref["pred_pos"]).abs().max())
    depth_ref = torch.zeros(hw)
    depth_ref[scene["ray_flat"][:hw].long()] = ref["pred_pos"][:, 2]
    d = (depth[0].reshape(-1).cpu() - depth_ref).abs()
    return {"reference": "oracle/lidf_oracle.py on the same frame (whole frame, %d points)" % P1,
            "depth_l1": float(d.mean()), "depth_max_abs": float(d.max()), "max_abs": mx,
            "argmax_mismatch_rays": int((got["max_pair_id"][:hw].cpu() != ref["max_pair_id"]).sum()),
            "tolerance": 1e-4, "ok": bool(max(mx.values()) <= 1e-4 and float(d.mean()) <= 1e-4)}


def _short(name):
    """Kernel name without its argument list (rocprofv3 reports the demangled signature)."""
    return name.split("(")[0].replace("void ", "").strip()


def _rocprof_child(opts, keep, steps, warmup, timeout):
    """This script as a child process under `rocprofv3 <opts>` (same workload flags, `steps` steps, no CPU
    baseline, no nested profiling); returns (results db path, temp dir) or (None, temp dir)."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    d = tempfile.mkdtemp(prefix="lidf_bench_prof_", dir="/tmp")
    if exe is None:
        return None, d
    env = dict(os.environ, TMPDIR="/tmp", LIDF_BENCH_NO_ROCPROF="1")
    cmd = [exe] + opts + ["-d", d, "-o", "r", "--", sys.executable, os.path.abspath(__file__),
                          "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline", "--no-rocprof",
                          "--no-split-f16"] + keep
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           timeout=timeout)
    except Exception:
        return None, d
    db = None
    for root, _, files in os.walk(d):
        for f in files:
            if f.endswith("_results.db"):
                db = os.path.join(root, f)
    return (db if r.returncode == 0 else None), d


def live_profile(argv, dominant, steps=6, warmup=2, pmc=False, timeout=300, per_step=1):
    """rocprofv3 legs of THIS command, run as child processes once the timed region is over — every number
    of the record is a number of this run on this box, none is read from a committed file:
      kernel trace (--kernel-trace --stats): per kernel the calls and the average duration (all launches,
        and excluding each kernel's first launch), launches per step (a step = one launch of `dominant`);
      pmc=True: two more passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; the TCC block cannot hold both) -> HBM
        bytes per launch, corrected as MI355X_MICROARCH.md §HBM prescribes for gfx950 (FETCH_SIZE tallies 64 B
        per 128-B request of a wide coalesced read: 2 x FETCH_SIZE + WRITE_SIZE, an upper bound for mixed
        access widths; the counters are KiB per dispatch).
    None when rocprofv3 is absent, disabled (LIDF_BENCH_NO_ROCPROF=1) or a child fails."""
    import shutil
    import sqlite3
    if os.environ.get("LIDF_BENCH_NO_ROCPROF") == "1":
        return None
    keep, skip = [], False
    for a in argv:            # the workload flags of this run without its step counts
        if skip:
            skip = False
            continue
        if a in ("--steps", "--warmup", "--gpus"):
            skip = True
            continue
        if a.startswith(("--steps=", "--warmup=", "--gpus=")) or a in ("--no-cpu-baseline", "--no-rocprof", "--pmc",
                                                                        "--no-split-f16"):
            continue
        keep.append(a)
    out = None
    dirs = []
    try:
        db, d = _rocprof_child(["--kernel-trace", "--stats"], keep, steps, warmup, timeout)
        dirs.append(d)
        if db is None:
            return None
        cur = sqlite3.connect(db).cursor()
        per, order = {}, []
        for name, dur in cur.execute("select name, duration from kernels order by start"):
            per.setdefault(_short(name), []).append(dur)
            order.append(_short(name))
        dom = [k for k in per if dominant in k]
        if not dom or len(per[dom[0]]) < 2:
            return None
        nstep = max(len(per[dom[0]]) // per_step, 1)   # (per_step launches of `dominant` make one step)
        # launches of a step in the steady state: what lies between the first and the last launch of `dominant`
        # (the process's set-up — module .to(device), synthetic inputs — issues copy / fill kernels of its own that
        # a division of all launches by the step count books on the steps: 74 copyBuffer launches of an
        # evaluation-path run are 74 whatever its number of frames)
        at = [i for i, k in enumerate(order) if k == dom[0]]
        steady = {}
        for k in order[at[0]:at[-1]]:
            steady[k] = steady.get(k, 0) + 1
        nsteady = max((len(at) - 1) // per_step, 1)
        kern = {}
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            kern[k] = {"calls_per_step": round(len(v) / nstep, 2), "avg_ms": round(sum(v) / len(v) / 1e6, 5),
                       "avg_ms_after_first": round(sum(v[1:]) / max(len(v) - 1, 1) / 1e6, 5),
                       "calls_per_step_steady": round(steady.get(k, 0) / nsteady, 2)}
        out = {"command": "rocprofv3 --kernel-trace --stats -- python bench.py --steps %d --warmup %d %s"
                          % (steps, warmup, " ".join(keep)),
               "steps_seen": nstep,
               "launches_per_step": round(sum(len(v) for v in per.values()) / nstep, 1),
               "launches_per_step_steady": round(sum(steady.values()) / nsteady, 1),
               "busy_ms_per_step": round(sum(sum(v) for v in per.values()) / nstep / 1e6, 4),
               # (the same sum without every kernel's first launch — code load, cold caches, clock ramp)
               "busy_ms_per_step_after_first": round(sum(sum(v[1:]) / max(len(v) - 1, 1) * len(v) for v in per.values())
                                                     / nstep / 1e6, 4),
               "kernels": dict(list(kern.items())[:24])}
        if pmc:
            vals = {}
            for cname in ("FETCH_SIZE", "WRITE_SIZE"):
                dbc, d = _rocprof_child(["--pmc", cname], keep, 3, 1, timeout)
                dirs.append(d)
                if dbc is None:
                    vals = None
                    break
                c2 = sqlite3.connect(dbc).cursor()
                for kn, n, avg in c2.execute("select kernel_name, count(*), avg(value) from counters_collection "
                                             "where counter_name=? group by kernel_name", (cname,)):
                    vals.setdefault(_short(kn), {})[cname] = (n, avg)
            if vals:
                hb, step_bytes = {}, 0.0
                ndom = [v for k, v in vals.items() if dominant in k]
                nst = max(ndom[0].get("FETCH_SIZE", (1, 0))[0] // per_step, 1) if ndom else 1
                for k, v in vals.items():
                    f, w = v.get("FETCH_SIZE", (0, 0.0)), v.get("WRITE_SIZE", (0, 0.0))
                    cor = (2.0 * f[1] + w[1]) * 1024.0
                    hb[k] = {"fetch_kib": round(f[1], 1), "write_kib": round(w[1], 1),
                             "bytes_corrected": round(cor), "calls_per_step": round(f[0] / max(nst, 1), 2)}
                    # (the children run the timed step's precision only — --no-split-f16 —, so every library
                    # kernel of the pass belongs to the step)
                    if "lidf_" in k:
                        step_bytes += cor * f[0] / max(nst, 1)
                out["hbm"] = {"command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE -- python bench.py --steps 3 "
                                         "--warmup 1 %s (two passes)" % " ".join(keep),
                              "correction": "2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md, gfx950), KiB per dispatch",
                              "bytes_per_step": round(step_bytes),
                              "kernels": dict(sorted(hb.items(), key=lambda kv: -kv[1]["bytes_corrected"])[:12])}
            # matrix-pipe counters (their own pass): MFMA FLOP issued, the shader clock the launch ran at
            # (SQ_BUSY_CYCLES / 32 shader engines / duration) and the share of those cycles the matrix pipe
            # was busy (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles) — meaningful for launches that fill
            # all 32 shader engines
            dbm, d = _rocprof_child(["--kernel-trace", "--pmc", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_BUSY_CYCLES",
                                     "SQ_VALU_MFMA_BUSY_CYCLES"], keep, 3, 1, timeout)
            dirs.append(d)
            if dbm is not None:
                c3 = sqlite3.connect(dbm).cursor()
                dur = {_short(r[0]): r[1] for r in c3.execute("select name, avg(duration) from kernels group by name")}
                by = {}
                for kn, cn, avg in c3.execute("select kernel_name, counter_name, avg(value) from counters_collection "
                                              "group by kernel_name, counter_name"):
                    by.setdefault(_short(kn), {})[cn] = avg
                mf = {}
                for k, v in by.items():
                    t = dur.get(k, 0.0) * 1e-9
                    if t <= 0 or not v.get("SQ_BUSY_CYCLES"):
                        continue
                    clk = v["SQ_BUSY_CYCLES"] / 32.0 / t
                    mf[k] = {"us": round(t * 1e6, 1), "ghz": round(clk / 1e9, 3),
                             "mfma_busy": round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / (clk * t), 3),
                             # (the counter ticks once per 512 f32 MFMA FLOP)
                             "mfma_flop": round(v.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512.0)}
                out["mfma"] = {"command": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES "
                                          "SQ_VALU_MFMA_BUSY_CYCLES -- python bench.py --steps 3 --warmup 1 %s"
                                          % " ".join(keep),
                               "kernels": dict(sorted(mf.items(), key=lambda kv: -kv[1]["us"])[:10])}
        return out
    except Exception:
        return out
    finally:
        for d in dirs:
            shutil.rmtree(d, ignore_errors=True)


def chain_launches_per_step(gf, P):
    """Launches of lidf_chain16_kernel in one query at another decoder width: two decoders x the slabs of generic.query."""
    from implicit_depth_amd import generic as g
    slab = g.QUERY_SLAB * (g.CHAIN_SLAB_FACTOR if gf >= 128 else 1)
    return 2 * ((P + slab - 1) // slab)


def live_kernel(live, name):
    """(avg ms after the first launch, avg ms of all launches, calls per step) of the kernel whose short name
    contains `name` in a live_profile record, or None."""
    if not live:
        return None
    for k, v in live["kernels"].items():
        if name in k:
            return v
    return None


# launches per step of the f32 query (for the per-step HBM counter sum)
PER_STEP_LAUNCHES = {"lidf_pack_kernel": 0, "void lidf_points_kernel<2>": 0,
                     "rowsh::lidf_rows_h_kernel": 0, "_ZN5rowsh18lidf_pack_r_kernelE12StreamLayout4NetWS1_5L1MapPDF16_Pf": 0}


def refine_setup(scene, s, dev, precision="f32"):
    """Synthetic stage-2 inputs for configs[3]: 10,000 valid points (valid_sample_num), a refine
    PointNet2Stage and IEF(D=334) with seeded weights, the occupied-voxel boxes of the 9^3 grid."""
    from implicit_depth_amd import IEF, PointNet2Stage
    from implicit_depth_amd.query import lidf_refine
    from implicit_depth_amd.synthetic import init_decoder_params
    g = torch.Generator().manual_seed(4321)
    B, h, w, V = scene["B"], scene["h"], scene["w"], scene["V"]
    vb = torch.cat((scene["vox_center"] - 0.125, scene["vox_center"] + 0.125), 1).to(dev)
    vbid = torch.arange(B).repeat_interleave(V // B).int().to(dev)
    rgb = torch.randn(B, 3, h, w, generator=g).to(dev)
    nv = 10000 * B
    valid_inp = (torch.randn(nv, 6, generator=g) * 0.2).to(dev)
    valid_vox = torch.randint(0, V, (nv,), generator=g).int().to(dev)
    torch.manual_seed(99)
    pnet = PointNet2Stage(6, 128, 32).to(dev).eval()
    offr = IEF(dev, 334, 1, 64, n_iter=2).to(dev).eval()
    offr.load_state_dict(init_decoder_params("IEF", 334, 9, 5.0))
    # the voxel list as cells of its grid (what LIDF.get_occ_vox_bound keeps, models/pipeline.py:167-201): the end
    # voxel of a ray (pcl_aabb + scatter max, :939-944) is looked up in a cell table instead of testing every ray
    # against every voxel (round 6: the record no longer carries lidf_refine_endvox_kernel)
    from implicit_depth_amd.synthetic import GRID_RES, GRID_XMIN, PART_SIZE
    ci = torch.arange(GRID_RES, dtype=torch.int32)
    coord = torch.stack(torch.meshgrid(ci, ci, ci, indexing="ij"), -1).reshape(-1, 3).repeat(B, 1).contiguous()
    grid = {"voxel_coord": coord.to(dev), "grid_dims": (GRID_RES,) * 3, "xmin": list(GRID_XMIN),
            "part_size": PART_SIZE}

    def run(out, events=None):
        return lidf_refine(s["ray_dir"], s["ray_pix"], s["ray_bid"], s["ray_flat"], out["pred_pos"],
                           out["max_pair_id"], s["pair_vox"], vb, vbid, rgb, s["feat_grid"], valid_inp,
                           valid_vox, pnet, offr, forward_times=2, rayfeat=out["rayfeat"],
                           precision=precision, profile_events=events, grid=grid)[0]
    run.n_valid = nv
    return run


def secondary(args):
    """Secondary single-GPU measurements with the same JSON schema (not the headline metric)."""
    from implicit_depth_amd import IEF, IMNet, decoders_forward, get_embedder
    from implicit_depth_amd.synthetic import init_decoder_params
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    P = 240 * 320 * args.samples
    g = torch.Generator(device=dev).manual_seed(5)
    if args.workload == "decoders":
        x = torch.randn(P, 385, generator=g, device=dev)
        prob = IMNet(385, 1, 64).to(dev).eval()
        prob.load_state_dict(init_decoder_params("IMNET", 385, 7, 5.0))
        off = IEF(dev, 385, 1, 64, n_iter=2).to(dev).eval()
        off.load_state_dict(init_decoder_params("IEF", 385, 8, 5.0))

        def step():
            with torch.no_grad():
                return decoders_forward(x, prob, off, precision=args.precision)
        flop_alg, bytes_alg, name = F_ALG, 1548.0, ("lidf_rows_h_kernel" if args.precision == "f16x3"
                                                   else "lidf_points_kernel<ROWS>")
        what = "prob_dec (IMNet) + offset_dec (IEF n_iter=2) on a materialised [P,385] f32 input"
    elif args.workload == "train-query":
        # one training step of the whole query: forward + backward to feat_grid, vox_feat and the
        # decoders' parameters (lidf_query_train), on one 240x320 frame with samples/8 candidates
        from implicit_depth_amd.query import lidf_query_train
        from implicit_depth_amd.synthetic import synthetic_scene
        scene = synthetic_scene(1, 240, 320, max(args.samples // 8, 1), seed=1235)
        P = scene["P"]
        sd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in scene.items()}
        prob = IMNet(scene["D"], 1, 64).to(dev).train()
        prob.load_state_dict(scene["prob_p"])
        off = IEF(dev, scene["D"], 1, 64, n_iter=2).to(dev).train()
        off.load_state_dict(scene["off_p"])
        fg = sd["feat_grid"].clone().requires_grad_(True)
        vf = sd["vox_feat"].clone().requires_grad_(True)
        Npr = max(args.samples // 8, 1)   # (the synthetic list is dense: Npr candidates on every ray)
        gl = torch.Generator().manual_seed(99)
        gt_pos = (torch.rand(scene["R"], 3, generator=gl) * 2 - 1).to(dev)
        label = torch.randint(0, Npr, (scene["R"], 1), generator=gl).to(dev)

        def step():
            for m in (prob, off):
                for p in m.parameters():
                    p.grad = None
            fg.grad = None
            vf.grad = None
            o = lidf_query_train(sd["ray_dir"], sd["ray_pix"], sd["ray_bid"], sd["pair_off"], sd["pair_ray"],
                                 sd["pair_vox"], sd["pair_t"], fg, vf, prob, off, offsets=args.offsets)
            # the reference's loss structure (models/pipeline.py:468-490): position loss on pred_pos (the selected
            # pair of every ray), ray-wise cross-entropy of the logits at a label pair; --dense-offset-grad adds a
            # term on pred_offset itself, which no loss of the reference has (every pair then carries a gradient)
            pos_loss = (o["pred_pos"] - gt_pos).abs().mean()
            lsm = torch.log_softmax(o["pred_prob_end"].view(-1, Npr), dim=1)
            prob_loss = -lsm.gather(1, label).mean()
            loss = pos_loss + prob_loss
            if args.dense_offset_grad:
                loss = loss + o["pred_offset"].sum()
            loss.backward()
        flop_alg, bytes_alg, name = 3.0 * F_ALG, 0.0, "lidf_points_fused_train_kernel + lidf_dgrad_chain_kernel + lidf_wgrad2_kernel"
        what = ("training step of the query (lidf_query_train): ROI pooling, decoder input rows, both "
                "decoders forward + backward, gradients to feat_grid / vox_feat / parameters; loss = L1(pred_pos, gt) + "
                "ray-wise cross-entropy of the logits (pipeline.py:468-490)%s; 240x320 rays x %d candidates"
                % (" + a term on pred_offset of every pair (--dense-offset-grad)" if args.dense_offset_grad else
                   ": offset_dec's backward runs over the selected pair of every ray", max(args.samples // 8, 1)))
    elif args.workload == "train-refine":
        # one training step of stage 2 at the reference's per-GPU shape (trainers/train_refine.py:393-399,
        # train_lidf.yaml: batch 8 per GPU, 20,000 miss rays + 10,000 valid points per image): RefineNet.forward
        # (2 x get_pred_refine) + backward to every PointNet2Stage / IEF parameter; stage 1 is frozen there
        # (its pred_pos / max_pair_id arrive detached)
        from implicit_depth_amd import PointNet2Stage
        from implicit_depth_amd.query import lidf_refine_train
        from implicit_depth_amd.synthetic import synthetic_scene
        Bt, n_miss, n_val = 8, 20000, 10000
        scene = synthetic_scene(Bt, 240, 320, 1, seed=1236)
        gc = torch.Generator().manual_seed(77)
        sel = torch.cat([torch.randperm(240 * 320, generator=gc)[:n_miss].sort().values + b * 240 * 320
                         for b in range(Bt)])
        rays = {k: scene[k][sel].contiguous().to(dev) for k in ("ray_dir", "ray_pix", "ray_bid", "ray_flat")}
        # occupied voxels = the cells the rays' candidates fall into (a real frame has 50-150 of the 729)
        occ, inv = torch.unique(scene["pair_vox"][sel].long(), return_inverse=True)
        V = int(occ.numel())
        pair_vox = inv.int().contiguous().to(dev)                        # one candidate per ray: pair r = ray r
        t_mid = scene["pair_t"][sel].mean(1, keepdim=True)
        pred_pos = (scene["ray_dir"][sel] * t_mid).contiguous().to(dev)  # stage 1's prediction, detached
        max_pair_id = torch.arange(sel.numel(), dtype=torch.int64, device=dev)
        vc = scene["vox_center"][occ]
        vb = torch.cat((vc - 0.125, vc + 0.125), 1).to(dev)
        vbid_h = (occ // 729).int()
        vbid = vbid_h.to(dev)
        rgb = torch.randn(Bt, 3, 240, 320, generator=gc).to(dev)
        feat = scene["feat_grid"].to(dev)
        valid_inp = (torch.randn(Bt * n_val, 6, generator=gc) * 0.2).to(dev)
        parts = []
        for b in range(Bt):   # the valid points of a frame fall into that frame's voxels
            own = torch.nonzero(vbid_h == b)[:, 0]
            parts.append(own[torch.randint(0, own.numel(), (n_val,), generator=gc)])
        valid_vox = torch.cat(parts).int().to(dev)
        torch.manual_seed(99)
        pnet = PointNet2Stage(6, 128, 32).to(dev).train()
        offr = IEF(dev, 334, 1, 64, n_iter=2).to(dev).train()
        offr.load_state_dict(init_decoder_params("IEF", 334, 9, 5.0))
        offr = offr.to(dev)
        P = sel.numel()                                                  # "points" of this record = rays
        # the voxel list as cells of its grid (LIDF.get_occ_vox_bound keeps them, models/pipeline.py:167-201): the end
        # voxels of both iterations through the cell table instead of the every-voxel test (round 6)
        from implicit_depth_amd.synthetic import GRID_RES, GRID_XMIN, PART_SIZE
        cell = occ % (GRID_RES ** 3)
        coord = torch.stack((cell // (GRID_RES * GRID_RES), (cell // GRID_RES) % GRID_RES, cell % GRID_RES), 1)
        grid = {"voxel_coord": coord.int().contiguous().to(dev), "grid_dims": (GRID_RES,) * 3,
                "xmin": list(GRID_XMIN), "part_size": PART_SIZE}

        def step():
            for m in (pnet, offr):
                for p in m.parameters():
                    p.grad = None
            pos, _ = lidf_refine_train(rays["ray_dir"], rays["ray_pix"], rays["ray_bid"], rays["ray_flat"], pred_pos,
                                       max_pair_id, pair_vox, vb, vbid, rgb, feat, valid_inp, valid_vox, pnet, offr,
                                       forward_times=2, grid=grid)
            pos.sum().backward()
        flop_alg, bytes_alg, name = 0.0, 0.0, ("lidf_refine_train_forward_f32 / _backward_f32: lidf_points_kernel<TRAIN> + "
                                              "lidf_pnet_train_fwd_kernel<1|2> + lidf_pnet_bwd_a|b_kernel + "
                                              "lidf_dgrad_chain_kernel + lidf_wgrad2_kernel")
        what = ("training step of stage 2 (lidf_refine_train = RefineNet.forward 'train', trainers/train_refine.py:"
                "393-399): 2 x get_pred_refine over %d frames x %d miss rays + %d valid points, %d voxels, forward "
                "+ backward to every PointNet2Stage / IEF(D=334) parameter; rows = rays" % (Bt, n_miss, n_val, V))
        # forward FLOP per ray and iteration: IEF(334 + 16) layer 1 once + per pass the encoding columns and
        # layers 2-4; PointNet2Stage 35,008 MAC per point over (valid + predicted) points
        ief = 2.0 * 256 * 334 + 2 * (2.0 * 256 * 16 + 2.0 * (256 * 128 + 128 * 64 + 64))
        pn = 2.0 * (6 * 32 + 32 * 64 + 128 * 128 + 128 * 128) * (Bt * n_val + P) / P
        train_refine_fwd = 2 * (ief + pn)
    elif args.workload == "train":
        # one training step of both decoders at the decoder boundary: forward that keeps the
        # activations + backward (input and parameter gradients), liblidf_hip on both sides
        P = 240 * 320 * args.samples // 8      # 614,400 rows: the activations of 3 passes stay modest
        x = torch.randn(P, 385, generator=g, device=dev).requires_grad_(True)
        prob = IMNet(385, 1, 64).to(dev).train()
        prob.load_state_dict(init_decoder_params("IMNET", 385, 7, 5.0))
        off = IEF(dev, 385, 1, 64, n_iter=2).to(dev).train()
        off.load_state_dict(init_decoder_params("IEF", 385, 8, 5.0))

        def step():
            for m in (prob, off):
                for p in m.parameters():
                    p.grad = None
            x.grad = None
            if args.decoder_pair:   # both decoders as one autograd node (decoders_forward_train): one product for d rows
                from implicit_depth_amd import decoders_forward_train
                yp, yo = decoders_forward_train(x, prob, off)
                (yp.sum() + yo.sum()).backward()
            else:
                (prob(x).sum() + off(x).sum()).backward()
        flop_alg, bytes_alg, name = 3.0 * F_ALG, 0.0, "lidf_points_kernel<TRAIN> + lidf_dgrad_chain_kernel + lidf_wgrad2_kernel"
        what = ("training step of prob_dec (IMNet) + offset_dec (IEF n_iter=2) on [P,385] rows: "
                "forward with kept activations + backward (d input, d parameters); FLOP = 3 x forward"
                + ("; both decoders as ONE autograd node (decoders_forward_train: the rows' gradient is one K = 512 "
                   "product over both decoders, stored once)" if args.decoder_pair else
                   "; the two modules as two autograd nodes (the drop-in of pipeline.py:434-435)"))
    else:
        x = (torch.rand(P, 3, generator=g, device=dev) - 0.5) * 4.6
        fn, dim = get_embedder(8)

        def step():
            return fn(x)
        flop_alg, bytes_alg, name = 0.0, 12.0 + 4.0 * dim, "lidf_embed_kernel"
        what = "get_embedder(8) on [P,3] f32 -> [P,51]"
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms = e0.elapsed_time(e1) / args.steps  # one dominant kernel per step on torch's stream
    # the host's share of a step (enqueue only: the device is idle at every start, nothing waits inside step()):
    # a step whose host time approaches its device time reads the box's CPU, not the kernels
    host = 0.0
    nh = min(args.steps, 10)
    for _ in range(nh):
        torch.cuda.synchronize()
        th = time.perf_counter()
        step()
        host += time.perf_counter() - th
    torch.cuda.synchronize()
    host_ms = host / max(nh, 1) * 1e3
    hbm = args.workload == "embed"
    split_rows = args.workload == "decoders" and args.precision == "f16x3"
    # frac = issued MFMA FLOP / time / peak where the instruction count is known (decoders); the
    # training steps in the FLOP of the executed formulation; achieved_alg = model FLOPs (F_alg)
    flop_exec, basis = flop_alg, "model FLOPs (F_alg), not a hardware utilisation"
    if split_rows:
        # executed: 2 nets x 25 layer-1 k-steps x 24 + 3 passes x 254 v_mfma_f32_32x32x16_f16 per 32 rows
        flop_exec, basis = (2 * 25 * 24 + 3 * 254) * 32768 / 32.0, "issued MFMA FLOP"
    elif args.workload == "decoders":
        # rows mode: 2 nets x 49 k-quads x 8 tiles x 4 + 3 passes x 654 v_mfma_f32_32x32x2_f32 per 32 rows
        flop_exec, basis = (2 * 49 * 8 * 4 + 3 * 654) * 4096 / 32.0, "issued MFMA FLOP"
    elif args.workload == "train-refine":
        flop_exec, basis = 3.0 * train_refine_fwd, "FLOP of the executed formulation (3 x forward), not an instruction count"
        flop_alg = flop_exec
    elif args.workload in ("train", "train-query"):
        # 3 x the forward FLOP of the formulation that runs (forward, input gradient, weight
        # gradient), without tile padding: layers 2-4 per pass 2 (256*128 + 128*64 + 64); layer 1
        # per net on the per-pair operand (train-query: the 102 position-embedding columns, the
        # voxel / ray parts are per-voxel / per-ray products; train: all 385 columns, once per net
        # for the IEF), + the IEF rank-1 term per pass
        chain, k1 = 2.0 * (256 * 128 + 128 * 64 + 64), (102 if args.workload == "train-query" else 385)
        fwd_prob, fwd_off = 2.0 * 256 * k1 + chain, 2.0 * 256 * k1 + 2 * chain + 2 * (2.0 * 256)
        fwd = fwd_prob + fwd_off
        flop_exec, basis = 3.0 * fwd, "FLOP of the executed formulation (3 x forward), not an instruction count"
        if args.workload == "train-query" and not args.dense_offset_grad:
            # offset_dec's backward (2 x its forward) over one pair per ray instead of every pair (--offsets selected:
            # its forward as well)
            npr = max(args.samples // 8, 1)
            flop_exec = fwd_prob + fwd_off / (npr if args.offsets == "selected" else 1) + 2.0 * fwd_prob + 2.0 * fwd_off / npr
            basis = ("FLOP of the executed formulation: forward of both decoders on every pair, backward (2 x forward) of "
                     "prob_dec on every pair and of offset_dec on the selected pair of every ray; not an instruction count")
    ach = (bytes_alg * P / (kern_ms * 1e-3) / 1e9) if hbm else (flop_exec * P / (kern_ms * 1e-3) / 1e12)
    peak = 8000.0 if hbm else (PEAK_F16_TFLOPS if split_rows else PEAK_F32_TFLOPS)
    # the training steps carry their own rocprofv3 leg (a child process of this same command after the timed
    # region): launches per step, busy time per step and the kernels of a step, largest first. A step is counted by
    # a kernel it launches a known number of times.
    live = None
    marker = {"train-refine": ("lidf_pnet_bwd_b_kernel", 2), "train-query": ("lidf_points_fused_train_kernel", 1),
              "train": ("lidf_points_kernel<5>", 2)}.get(args.workload)
    profiled = any(k.startswith(("ROCP_", "ROCPROF")) for k in os.environ)
    if marker and not args.no_rocprof and not profiled:
        live = live_profile(sys.argv[1:], marker[0], steps=6, warmup=2, per_step=marker[1])
    emit({
        "metric": "Mpoints/sec, %s" % args.workload, "value": round(P * args.steps / elapsed / 1e6, 2),
        "unit": "Mpoints/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "host_enqueue_ms_per_step": round(host_ms, 4),
        "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPE_F16X3 if (args.workload == "decoders" and args.precision == "f16x3") else "f32",
        "data": "synthetic",
        "config": {"workload": "secondary: %s, P = %d rows" % (what, P),
                   # (train-query: offset_dec's backward over the selected rows and the forward's embedding rows run on
                   # a side stream beside the large launches — kernels overlap, busy time per step exceeds the step)
                   "concurrent_streams": (2 if args.workload == "train-query" and
                                          os.environ.get("LIDF_TRAIN_STREAMS", "2") != "1" else 1)},
        "roofline": {"bound": "hbm" if hbm else "mfma", "achieved": round(ach, 2), "peak": peak,
                     "unit": "GB/s" if hbm else "TFLOP/s", "frac": round(ach / peak, 4),
                     "traffic": None, "kernel": name, "kernel_ms": round(kern_ms, 4),
                     "basis": "algorithmic bytes" if hbm else basis,
                     "achieved_alg": None if hbm else round(flop_alg * P / (kern_ms * 1e-3) / 1e12, 2)},
        "profile": ({k: live[k] for k in ("command", "steps_seen", "launches_per_step", "launches_per_step_steady",
                                          "busy_ms_per_step", "busy_ms_per_step_after_first", "kernels") if k in live} if live else None)})


def e2e(args):
    """--workload e2e: the whole evaluation path of a batch of frames on a ragged geometry-derived
    scene (synthetic_batch): valid points -> occupied voxels -> PointNet2Stage -> miss rays -> compact
    ray/voxel pairs -> fused query -> 2 x get_pred_refine -> eval metrics. Secondary record, same schema.
      --e2e-mode frame (default)  pipeline.FrameRunner: ONE library call per batch (lidf_frame_f32),
                                  list lengths stay on the device, no host round trip inside the step
      --e2e-mode graph            the same call captured in a HIP graph and replayed
      --e2e-mode stepwise         pipeline.lidf_forward + refine_forward: one call per reference method,
                                  every list sized on the host (four syncs per frame, as the reference's
                                  nonzero()/unique() have)"""
    from implicit_depth_amd import IEF, IMNet, PointNet2Stage, pipeline as pl
    from implicit_depth_amd.synthetic import init_decoder_params, synthetic_batch
    # --gpus N (the reference refuses multi-GPU evaluation, trainers/train_lidf.py:693-694, and
    # shards its training batch over the ranks, :162-164): the frames of an evaluation stream are sharded over
    # the ranks (dist.shard_frames: rank g owns frames [g B, (g+1) B) of the world * B frames of a step), one
    # FrameRunner per rank, weights replicated, no data-path collective; the refined depth maps of all ranks
    # are all-gathered inside the timed region. Weak scaling: every rank runs --frames frames per step.
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    share_gpu = os.environ.get("LIDF_TEST_SHARE_GPU") == "1"   # (tests only: all ranks on cuda:0 over gloo)
    if share_gpu:
        local = 0
    dev = torch.device("cuda", local)
    torch.cuda.set_device(local)
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        if args.e2e_mode == "stepwise" or max(1, args.streams) != 1:
            raise SystemExit("--workload e2e --gpus N shards frames over FrameRunners (frame / graph mode, one stream)")
    B, h, w = args.frames, 240, 320
    from implicit_depth_amd.dist import shard_frames
    f0, _f1 = shard_frames(world * B, world, rank)
    # (frame f of the stream is synthetic frame seed 77 + f // B: rank g's batch is the one a single GPU
    # would have met as its g-th batch)
    batch, feat = synthetic_batch(B, h, w, seed=77 + f0 // B)
    batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
    feat = feat.to(dev)
    torch.manual_seed(3)
    pnet = PointNet2Stage(6, 128, 32).to(dev).eval()
    pnet_r = PointNet2Stage(6, 128, 32).to(dev).eval()
    prob = IMNet(385, 1, 64).to(dev).eval()
    prob.load_state_dict(init_decoder_params("IMNET", 385, 7, 5.0))
    off = IEF(dev, 385, 1, 64, n_iter=2).to(dev).eval()
    off.load_state_dict(init_decoder_params("IEF", 385, 8, 5.0))
    offr = IEF(dev, 334, 1, 64, n_iter=2).to(dev).eval()
    offr.load_state_dict(init_decoder_params("IEF", 334, 9, 5.0))
    # valid_sample_num = 10000 of the shipped configs, as a deterministic stride over the valid pixels
    n_valid = int((batch["depth_corrupt"] != 0).sum().item())
    opt = pl.LidfOptions(valid_stride=max(1, n_valid // (10000 * B)))
    state = {"ws": None}
    mode = args.e2e_mode
    stages = {}
    if mode == "stepwise":
        def step(marks=None):
            with torch.no_grad():
                ok, dd = pl.lidf_forward(batch, feat, pnet, prob, off, opt, precision=args.precision,
                                         workspace=state["ws"], marks=marks, offsets=args.offsets)
                assert ok
                state["ws"] = dd["workspace"]
                pl.refine_forward(dd, pnet_r, offr, opt, precision=args.precision)
                pl._mark(marks, "refine_x2")
                m = pl.eval_metrics(dd, "pred_depth_refine")
                pl._mark(marks, "metrics")
            return dd, m
    else:
        # --streams S: S runners (each with its own buffers, packed-weight entries and stream) take the
        # batches in turn, so that the low-occupancy stretches of one frame (PointNet, per-voxel layers,
        # scans, the partial last round of the matrix kernels) are filled by its neighbour's kernels
        S = max(1, args.streams)
        # FrameRunner's default: a side stream inside eager single-stream frames only (as FramePipeline does)
        side_mode = True if args.side_stream else (False if (args.no_side_stream or S > 1) else None)
        runners = [pl.FrameRunner(B, h, w, dev, pnet, prob, off, opt, pnet_r, offr, precision=args.precision,
                                  guard_every=args.guard_every, offsets=args.offsets,
                                  side_stream=side_mode, lds_voxels=args.lds_voxels or None)
                   for _ in range(S)]
        lanes = [torch.cuda.Stream(dev) for _ in range(S)] if S > 1 else [None]
        for r in runners:
            with torch.no_grad():
                r.load(batch, feat)
                if mode == "graph":
                    r.capture()
        runner = runners[0]
        turn = {"i": 0, "timed": False}
        # HIP events of the library around the per-point kernel and the two stage-2 decoder launches
        # (LidfFrameArgs.profile_events): the HIP-event clock of the fractions below, beside the profiler's.
        # Only where the launches run alone — one stream, no side stream, eager.
        hev, ev_sets = None, []
        if S == 1 and mode == "frame" and side_mode is False and args.precision == "f32" and args.offsets == "all":
            hev = HipEvents()
            ev_sets = [[hev.create() for _ in range(6)] for _ in range(min(args.steps, 64))]

        gathered = torch.empty((world * B, h, w), device=dev) if use_dist else None

        def step(marks=None):
            k = turn["i"] % S
            turn["i"] += 1
            with torch.no_grad():
                if S == 1:
                    pl._mark(marks, "start")
                    if ev_sets and marks is not None and (turn["i"] - 1) % args.steps < len(ev_sets) and turn["timed"]:
                        runner.profile_events = ev_sets[(turn["i"] - 1) % args.steps]
                    else:
                        runner.profile_events = None
                    runner.run(batch, feat)          # input copies + the frame (+ graph replay)
                    pl._mark(marks, "frame")
                    m = runner.metrics(batch)
                    pl._mark(marks, "metrics")
                    if use_dist:                     # the one collective: every rank's refined depth maps
                        from implicit_depth_amd.dist import all_gather_depth
                        all_gather_depth(runner.buf["pred_depth_refine"], gathered)
                        pl._mark(marks, "all_gather")
                else:
                    with torch.cuda.stream(lanes[k]):
                        runners[k].run(batch, feat)
                        m = runners[k].metrics(batch)
            return None, m

    def barrier():
        if use_dist:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    all_marks = []
    if mode != "stepwise":
        turn["i"], turn["timed"] = 0, True
    for _ in range(args.steps):
        mk = []
        dd, m = step(mk)
        all_marks.append(mk)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    hip_ms = None
    if mode != "stepwise" and ev_sets:
        n_ev = min(args.steps, len(ev_sets))
        hip_ms = [sum(hev.elapsed_ms(es[2 * j], es[2 * j + 1]) for es in ev_sets[:n_ev]) / n_ev for j in range(3)]
    for mk in all_marks:
        for (n0, e0), (n1, e1) in zip(mk[:-1], mk[1:]):
            stages[n1] = stages.get(n1, 0.0) + e0.elapsed_time(e1) / args.steps
    if mode == "stepwise":
        P, R = int(dd["pair_ray"].shape[0]), int(dd["miss_ray_dir"].shape[0])
        V, NV = int(dd["voxel_bound"].shape[0]), int(dd["valid_xyz"].shape[0])
        syncs = "4 host size reads per frame"
    else:
        c = runner.counts()                      # the one size read, after the timed region
        assert not c["OVERFLOW"]
        P, R, V, NV = c["P"], c["R"], c["V"], c["NVS"]
        syncs = "none inside the step (list lengths stay on the device)"
    coll = None
    P_all, R_all = P, R
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        every = torch.empty((world,), device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(every, t)               # every rank's own clock
        rank_ms = [round(float(v) / args.steps * 1e3, 4) for v in every.tolist()]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())                           # the step time of the job = the slowest rank's
        cnt = torch.tensor([1, P, R], device=dev, dtype=torch.int64)
        dist.all_reduce(cnt)                                # ranks that joined; pairs / rays of all ranks
        ranks_seen, P_all, R_all = (int(v) for v in cnt.tolist())
        mine = bool((gathered[rank * B:(rank + 1) * B] == runner.buf["pred_depth_refine"]).all()) and \
            bool(torch.isfinite(gathered).all())
        okt = torch.tensor([1 if mine else 0], device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        coll = {"op": "all_gather_into_tensor (RCCL) of [%d,%d,%d] f32 refined depth maps per rank" % (B, h, w),
                "inside_timed_region": True, "gathered_equals_local": bool(okt.item()), "ranks_seen": ranks_seen,
                "ms_per_step_by_rank": rank_ms, "pairs_all_ranks": P_all, "rays_all_ranks": R_all,
                "frames_all_ranks": world * B}
    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return
    # rocprofv3 legs of this same command (child processes after the timed region): launches per step, busy
    # time, the kernels of a step, and the issued-FLOP fractions of its three matrix kernels
    live = None
    profiled = any(k.startswith(("ROCP_", "ROCPROF")) for k in os.environ)
    if not args.no_rocprof and not profiled and max(1, args.streams) == 1 and not use_dist:
        live = live_profile(sys.argv[1:], "lidf_points_fused_kernel" if args.precision == "f32" else "lidf_points_h_kernel",
                            pmc=args.pmc)
    roof = None
    side_on = mode != "stepwise" and bool(side_mode or (side_mode is None and mode == "frame"))
    if live and args.precision == "f32" and mode != "stepwise":
        def frac(name, flop, hip=None):
            """Issued FLOP of one launch / its duration / peak, on BOTH clocks where both exist: `frac_rocprof` from the
            profiler's average duration (rocprofv3 --kernel-trace child of this command; short launches read up to
            12 % longer there), `frac` from HIP events of the timed run itself (LidfFrameArgs.profile_events)."""
            k = live_kernel(live, name)
            if not k:
                return None
            t = k["avg_ms_after_first"] * 1e-3
            out = {"kernel_ms_rocprof": k["avg_ms_after_first"], "calls_per_step": k["calls_per_step"],
                   "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                   "achieved_rocprof": round(flop / t / 1e12, 2),
                   "frac_rocprof": round(flop / t / 1e12 / PEAK_F32_TFLOPS, 4),
                   "clock": "frac_rocprof: rocprofv3 --kernel-trace (child run); frac: HIP events (timed run)"}
            if hip:
                out.update({"kernel_ms": round(hip, 5), "achieved": round(flop / (hip * 1e-3) / 1e12, 2),
                            "frac": round(flop / (hip * 1e-3) / 1e12 / PEAK_F32_TFLOPS, 4)})
            else:   # (no HIP-event leg in this mode: side stream / several streams / graph)
                out.update({"kernel_ms": None, "achieved": None, "frac": None})
            return out
        npn = c["NPN"] if mode != "stepwise" else 0
        roof = {
            # (offsets='selected': two launches of one net each — P points, then R — : no single-launch fraction)
            "points": (dict(frac("lidf_points_fused_kernel", F_EXEC * P, hip_ms[0] if hip_ms else None) or {},
                            flop_per_point_exec=F_EXEC)
                       if args.offsets == "all" else None),
            # stage-2 decoder (lidf_ief16_kernel): per 16 rays 4 x 16 x 4 layer-1 + 2 passes x (12 bias + 16 rank-1 +
            # 512 + 128) v_mfma_f32_16x16x4_f32 of 2048 FLOP
            "ief": dict(frac("lidf_ief16_kernel", F_IEF16 * R, (hip_ms[1] + hip_ms[2]) / 2 if hip_ms else None)
                        or frac("lidf_points_kernel<6>", F_IEF32 * R) or {},
                        sub_tiles=(R + 15) // 16, flop_per_ray_exec=F_IEF16),
            # PointNet2Stage chains of one refine pass: 44 (stage 1) and 444 (stage 2) matrix instructions per 32 points
            "pointnet_chain2": frac("lidf_pointnet_chain_kernel<2, true>", 444 * 4096 / 32.0 * (NV + 2 * npn) / 3.0),
            # (with the side stream the stage-2 table is a second launch of this kernel that runs beside — and is
            # stretched over — the per-point kernel: no per-launch fraction; the one-stream record carries it)
            "l1part": (frac("lidf_l1part_pair_kernel", 2.0 * 155 * (512 + 256) * R + 2.0 * 129 * 512 * V)
                       if not side_on else None),
        }
    emit({
        "metric": "Mpoints/sec, e2e evaluation path", "value": round(P_all * args.steps / elapsed / 1e6, 3),
        "unit": "Mpoints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPE_F16X3 if args.precision == "f16x3" else "f32",
        "data": "synthetic",
        "config": {"workload": "secondary: whole evaluation path of %d 240x320 frame(s): valid points, "
                               "occupied voxels, PointNet2Stage, miss rays, compact ray/voxel pairs, fused "
                               "query, 2 x get_pred_refine, eval metrics; geometry-derived ragged scene" % B,
                   "mode": mode, "host_syncs": syncs,
                   "offsets": ("selected pairs only (opt-in: pred_offset / pair_pred_pos undefined elsewhere; pred_pos, "
                               "depth, stage 2 and statistics bit-identical)" if args.offsets == "selected"
                               else "every pair (the reference's data flow)"),
                   "side_stream": (None if mode == "stepwise" else
                                   bool(side_mode or (side_mode is None and mode == "frame"))),
                   "guard_every": args.guard_every if mode != "stepwise" else None, "streams": max(1, args.streams) if mode != "stepwise" else 1,
                   "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "default (4)"),
                   "rays": R, "pairs": P, "pairs_per_ray": round(P / R, 3),
                   "voxels": V, "valid_points": NV,
                   "parallelism": "frames of the evaluation stream sharded over %d GPU(s), one FrameRunner per rank, "
                                  "%d frame(s) per rank per step%s" %
                                  (world, B, "; RCCL all-gather of the refined depth maps" if use_dist else "")},
        "ms_per_frame": round(elapsed / args.steps * 1e3 / (B * world), 4),
        "frames_per_s": round(B * world * args.steps / elapsed, 2),
        "rays_per_s": round(R_all * args.steps / elapsed, 1),
        "collective": coll,
        "stage_ms": {k: round(v, 4) for k, v in stages.items()},
        "roofline_kernels": roof,
        "roofline_kernels_note": ("side stream: the stage-2 layer-1 table runs beside (in the tail of) the per-point "
                                  "kernel, whose duration then includes the shared rounds; per-kernel fractions of "
                                  "launches that never overlap: --no-side-stream" if side_on and roof else None),
        "profile": ({k: live[k] for k in ("command", "launches_per_step", "launches_per_step_steady", "busy_ms_per_step",
                                          "busy_ms_per_step_after_first", "kernels", "hbm", "mfma")
                     if k in live} if live else None),
        "metrics_frame0": {k: round(float(v), 6) for k, v in m.items()}})
    if use_dist:
        dist.destroy_process_group()


def config_name(args, refine):
    """Which BASELINE.json configuration the per-GPU workload is (named in config.workload)."""
    if refine:
        return "configs[3]"
    if args.pairs != "dense":
        return "configs[1]"     # replaced by the 'secondary' wording below
    shape = (args.frames, args.samples)
    return {(1, 64): "configs[1]", (4, 64): "configs[2] per-GPU shard (32 frames over 8 GPUs)",
            (4, 256): "configs[4] per-GPU shard (32 frames x 256 candidates over 8 GPUs)",
            (1, 256): "configs[4], one frame of the shard"}.get(shape, "secondary shape")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--samples", type=int, default=64, help="candidates per ray (N)")
    ap.add_argument("--frames", type=int, default=1, help="frames per rank per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rocprof", action="store_true",
                    help="skip the rocprofv3 legs of an N = 1 run (roofline.kernel_ms_rocprof_live and the "
                         "\"profile\" block: per-kernel durations and launches per step by rocprofv3 --kernel-trace "
                         "of this same command, run as a child process after the timed region)")
    ap.add_argument("--no-split-f16", action="store_true",
                    help="query workload: skip the secondary legs of the default f32 run (the same workload through "
                         "the split-f16 kernel -> \"split_f16\", and with offsets for the selected pairs only -> "
                         "\"offsets_selected\") and the extra step that keeps outputs for the parity record. The "
                         "rocprofv3 children of a run pass it, so that profile.* and hbm.* are numbers of the timed "
                         "f32 step alone")
    ap.add_argument("--pmc", action="store_true",
                    help="also run the two HBM counter passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE) of "
                         "this command -> roofline.traffic, hbm.bytes_counter; the default headline run does so "
                         "by itself")
    ap.add_argument("--precision", default="f32", choices=["f32", "f16x3"],
                    help="arithmetic of the decoders' matrix products: f32 (default, the headline) or "
                         "f16x3 = three f16-piece products per term with f32 accumulation (f32-level "
                         "accuracy, LidfQueryArgs.precision); the default f32 run also reports the "
                         "f16x3 rate and its deviation from the f32 outputs as \"split_f16\"")
    ap.add_argument("--pairs", default="dense", choices=["dense", "ragged", "n1", "scene"],
                    help="candidate list of the query workload: dense = N per ray (the headline shape); "
                         "ragged = U{0..N} per ray; n1 = one per ray; scene = the pairs "
                         "compute_ray_aabb finds on a geometry-derived frame (0..8 per ray)")
    ap.add_argument("--config", type=int, default=None, choices=[1, 2, 3, 4],
                    help="BASELINE.json configs[k] at its per-GPU shape: 1 = one 240x320 frame x 64 "
                         "candidates (the headline, default); 2 = the per-GPU shard of the 32-frame batch "
                         "over 8 GPUs (4 frames x 64); 3 = stage 1 + 2 x get_pred_refine; 4 = the dense "
                         "resample stress shard (4 frames x 256 candidates). Sets --frames/--samples/"
                         "--workload; weak scaling: every rank runs this shard")
    ap.add_argument("--shard", default="frames", choices=["frames", "rays"],
                    help="multi-GPU partition: frames = every rank owns whole frames (weak scaling, the "
                         "default and what configs[2]/[4] describe); rays = ONE frame, image rows split "
                         "over the ranks (strong scaling, SURVEY 8e for fewer frames than GPUs), depth rows "
                         "all-gathered")
    ap.add_argument("--decoder-pair", action="store_true",
                    help="--workload train: both decoders as ONE autograd node (decoders_forward_train) instead of the "
                         "two modules' own nodes")
    ap.add_argument("--dense-offset-grad", action="store_true",
                    help="--workload train-query: add a loss term on pred_offset of every pair (not in the reference's "
                         "loss): offset_dec's backward then runs over all pairs, as in rounds 1-4")
    ap.add_argument("--streams", type=int, default=1,
                    help="query workloads: consecutive steps (independent frames) alternate over this many HIP "
                         "streams, so the light kernels at the head and tail of a step (per-ray features, "
                         "layer-1 partial products, per-ray reduce) run beside the previous step's per-point "
                         "kernel instead of in front of it; 1 (default) = one stream, steps strictly one after "
                         "another. Measured on MI355X (DESIGN.md 7): 388.4 / 388.2 / 391.1 Mpoints/s at 1 / 2 / 3 "
                         "streams — the per-point kernel already fills the matrix pipes, what is overlapped "
                         "comes back as a longer kernel")
    ap.add_argument("--imnet-gf", type=int, default=64,
                    help="query workload: gf_dim of both decoders (model.imnet_gf; 64 in every shipped config = "
                         "the fused kernels). Another value runs the layer-by-layer path of implicit_depth_amd/"
                         "generic.py (rows materialised in 614,400-pair slabs): a secondary record, no roofline")
    ap.add_argument("--offsets", default="all", choices=["all", "selected"],
                    help="--workload e2e (frame / graph / stepwise): 'selected' = the offset decoder on the arg-max pair "
                         "of every ray only (opt-in, FrameRunner(offsets=...)): pred_pos / depth / stage 2 / statistics "
                         "bit-identical, pred_offset / pair_pred_pos defined at the selected pairs only. A secondary "
                         "record that says so; the default and every other record compute the reference's full data flow")
    ap.add_argument("--side-stream", action="store_true",
                    help="--workload e2e (frame / graph): FrameRunner(side_stream=True) — the weight-stream guard, box "
                         "sums and per-ray features of a frame on a second stream beside its head, pairs and PointNet. "
                         "Default: FrameRunner's own (on for eager single-stream frames, off under a graph / --streams)")
    ap.add_argument("--no-side-stream", action="store_true", help="--workload e2e: FrameRunner(side_stream=False)")
    ap.add_argument("--lds-voxels", type=int, default=0,
                    help="--workload e2e (frame / graph): FrameRunner(lds_voxels=N) (0 = the runner's default)")
    ap.add_argument("--guard-every", type=int, default=1,
                    help="--workload e2e (frame / graph): FrameRunner(guard_every=N) — the packed weight streams are "
                         "re-validated (one fingerprint launch over every module + two early-exit pack launches) on "
                         "every N-th frame only; 1 (default) = every frame")
    ap.add_argument("--e2e-mode", default="frame", choices=["frame", "graph", "stepwise"],
                    help="--workload e2e: frame = one sync-free library call per batch (default), graph = "
                         "that call replayed from a HIP graph, stepwise = one call per reference method")
    ap.add_argument("--workload", default="query",
                    choices=["query", "query+refine", "decoders", "embed", "train", "train-query", "train-refine",
                             "e2e"],
                    help="query = BASELINE configs[1] (default, the headline metric); query+refine = "
                         "configs[3] (stage 1 + 2 x get_pred_refine); decoders = IMNet+IEF on a "
                         "materialised [P,385] input (the reference's decoder boundary); embed = "
                         "stand-alone positional encoding (the one HBM-bound kernel of the path)")
    args = ap.parse_args()
    if args.config is not None:
        preset = {1: (1, 64, "query"), 2: (4, 64, "query"), 3: (1, 64, "query+refine"), 4: (4, 256, "query")}
        args.frames, args.samples, args.workload = preset[args.config]
    if args.workload == "e2e" and (args.gpus > 1 or "RANK" in os.environ):
        # A frame forks onto a side stream; RCCL brings streams of its own, and with HIP's default of 4 hardware
        # queues the two streams of a frame can land on one queue (the frame then runs 6 % slower than without a
        # process group: profiles/r06_ab_pg.txt). Must be in the environment before the first HIP call.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # called as `python bench.py --gpus N`: become the launcher the driver would have used (one
        # process per GPU, RCCL rendezvous on the loopback address)
        import socket
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                                  "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
                                  "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    claim_stdout()
    if args.workload in ("decoders", "embed", "train", "train-query", "train-refine"):
        return secondary(args)
    if args.workload == "e2e":
        return e2e(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # LIDF_TEST_SHARE_GPU=1 (tests/test_rccl_gpu.py only): every rank on cuda:0 with the gloo backend, so
    # that the N > 1 logic (shards, gather slots, max-over-ranks clock) runs on a 1-GPU box. Never a
    # measurement: the ranks share one GPU and the collective goes through host memory.
    share_gpu = os.environ.get("LIDF_TEST_SHARE_GPU") == "1"
    if share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or "RANK" in os.environ  # under torchrun even N=1 goes through RCCL
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from implicit_depth_amd import IEF, IMNet
    from implicit_depth_amd.dist import all_gather_depth, all_gather_depth_rows
    from implicit_depth_amd.query import lidf_query
    from implicit_depth_amd.synthetic import synthetic_scene

    h, w, N, B = 240, 320, args.samples, args.frames
    if args.pairs == "n1":
        N = 1
    by_rays = args.shard == "rays"
    if by_rays and (B != 1 or args.pairs == "scene" or args.workload != "query"):
        raise SystemExit("--shard rays splits the rows of ONE frame of the query workload")
    scene = synthetic_scene(B, h, w, N, seed=1235 + (0 if by_rays else rank), ragged=args.pairs == "ragged")
    rows, row0 = (0, h), 0
    if by_rays:   # image rows [lo, hi) of the one frame; maps, voxel features and weights replicated
        from implicit_depth_amd.dist import crop_rows, shard_rays, slice_rays
        rows = shard_rays(h, world, rank)
        scene = slice_rays(scene, rows[0] * w, rows[1] * w)
        # (the feature map cut to the rank's rows + the 4-pixel RoIAlign halo: box sums and depth map of a
        # rank shrink with its shard; results bit-equal to the uncut map, tests/rccl_worker.py)
        scene, fg_cut, row0 = crop_rows(scene, scene["feat_grid"], rows[0], rows[1], 4)
        scene["feat_grid"] = fg_cut
    s = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in scene.items()}
    if args.pairs == "scene":
        # rays, voxels and pairs as the candidate generator produces them on real geometry
        from implicit_depth_amd import PointNet2Stage, pipeline as pl
        from implicit_depth_amd.synthetic import synthetic_batch
        batch, feat = synthetic_batch(B, h, w, seed=77 + rank)
        torch.manual_seed(3)
        pn = PointNet2Stage(6, 128, 32).to(dev).eval()
        pm = IMNet(scene["D"], 1, 64).to(dev).eval()
        om = IEF(dev, scene["D"], 1, 64, n_iter=2).to(dev).eval()
        pm.load_state_dict(scene["prob_p"]), om.load_state_dict(scene["off_p"])
        with torch.no_grad():
            ok, dd = pl.lidf_forward({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()},
                                     feat.to(dev), pn, pm, om)
        assert ok
        s.update({"ray_dir": dd["miss_ray_dir"], "ray_pix": dd["ray_pix"], "ray_bid": dd["ray_bid"],
                  "ray_flat": dd["ray_flat"], "pair_off": dd["pair_off"], "pair_ray": dd["pair_ray"],
                  "pair_vox": dd["pair_vox"], "pair_t": dd["pair_t"], "feat_grid": dd["full_rgb_feat"],
                  "vox_feat": dd["occ_voxel_feat"]})
        scene = dict(scene, P=int(dd["pair_ray"].shape[0]), V=int(dd["occ_voxel_feat"].shape[0]))
    P = scene["P"]
    dense = args.pairs == "dense"
    gf = args.imnet_gf
    prob = IMNet(scene["D"], 1, gf).to(dev).eval()
    off = IEF(dev, scene["D"], 1, gf, n_iter=2).to(dev).eval()
    if gf == 64:
        prob.load_state_dict(scene["prob_p"]), off.load_state_dict(scene["off_p"])
    else:   # the scene's seeds and scale at another width
        from implicit_depth_amd.synthetic import init_decoder_params
        prob.load_state_dict(init_decoder_params("IMNET", scene["D"], 7, 5.0, gf=gf))
        off.load_state_dict(init_decoder_params("IEF", scene["D"], 8, 5.0, gf=gf))
        prob, off = prob.to(dev), off.to(dev)
    S = max(1, args.streams)
    streams = [torch.cuda.Stream(dev) for _ in range(S)] if S > 1 else [None]
    hl = s["feat_grid"].shape[2]    # rows of this rank's maps (the whole image, or its row shard + halo)
    depths = [torch.zeros((B, hl, w), device=dev) for _ in range(S)]     # per stream: steps in flight
    gathers = [torch.empty((world * B, h, w), device=dev) if use_dist and not by_rays else None
               for _ in range(S)]
    depth, gathered = depths[0], gathers[0]
    refine = None
    if args.workload == "query+refine":
        refine = refine_setup(scene, s, dev, args.precision)
    ev = HipEvents()
    pairs = [(ev.create(), ev.create()) for _ in range(args.steps)]
    # stage 2: per step and iteration (PointNet begin, end, IEF rows kernel begin, end)
    rpairs = [[[ev.create() for _ in range(4)] for _ in range(2)] for _ in range(args.steps)] \
        if refine is not None and args.precision == "f32" else None
    state = {"ws": [None] * S, "step": None}

    def step(events=None, precision=args.precision, lane=0, offsets=args.offsets):
        depth, gathered = depths[lane], gathers[lane]
        with torch.no_grad():
            out = lidf_query(s["ray_dir"], s["ray_pix"], s["ray_bid"], s["pair_off"], s["pair_ray"],
                             s["pair_vox"], s["pair_t"], s["feat_grid"], s["vox_feat"], prob, off,
                             ray_flat=s["ray_flat"], depth=depth, workspace=state["ws"][lane],
                             profile_events=events, want_rayfeat=refine is not None,
                             precision=precision, offsets=offsets)
            if refine is not None:
                out["pred_pos_refine"] = refine(out, rpairs[state["step"]] if (rpairs and events is not None
                                                                                 and state["step"] is not None) else None)
                depth.view(-1)[s["ray_bid"].long() * (hl * w) + s["ray_flat"].long()] = \
                    out["pred_pos_refine"][:, 2]
        state["ws"][lane] = out["workspace"]
        if use_dist and by_rays:
            state["full"] = all_gather_depth_rows(depth[0, rows[0] - row0:rows[1] - row0], h)
        elif use_dist:
            all_gather_depth(depth, gathered)
        return out

    def lane_step(i, events=None):
        """Step i on stream i mod S (its own workspace, depth map and gather buffer)."""
        if S == 1:
            return step(events)
        with torch.cuda.stream(streams[i % S]):
            return step(events, lane=i % S)

    for i in range(max(args.warmup, S)):
        lane_step(i)

    def barrier():
        if use_dist:
            import torch.distributed as dist
            dist.barrier()

    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        state["step"] = i
        lane_step(i, None if gf != 64 else pairs[i])
    state["step"] = None
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    rank_ms, ranks_seen, points_all = [elapsed / args.steps * 1e3], 1, scene["P"]
    if use_dist:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        every = torch.empty((world,), device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(every, t)               # every rank's own clock
        rank_ms = [round(float(v) / args.steps * 1e3, 4) for v in every.tolist()]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        one = torch.ones((1,), device=dev, dtype=torch.int64)
        dist.all_reduce(one)                                # ranks that actually joined the RCCL group
        ranks_seen = int(one.item())
        pts = torch.tensor([scene["P"]], device=dev, dtype=torch.int64)
        dist.all_reduce(pts)                                # points of the whole job (shards may differ)
        points_all = int(pts.item())

    # (widths other than the shipped ones run layer by layer: no single dominant kernel to time)
    kern_ms = sum(ev.elapsed_ms(a, b) for a, b in pairs) / args.steps if gf == 64 else float("nan")
    gather_ok = None
    if use_dist:   # the collective's result: every rank's slot holds that rank's depth maps
        import torch.distributed as dist
        if by_rays:
            full = state["full"]
            mine = bool((full[rows[0]:rows[1]] == depth[0, rows[0] - row0:rows[1] - row0]).all()) and \
                bool(torch.isfinite(full).all()) and tuple(full.shape) == (h, w)
        else:
            mine = bool((gathered[rank * B:(rank + 1) * B] == depth).all()) and bool(torch.isfinite(gathered).all())
        t = torch.tensor([1 if mine else 0], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        gather_ok = bool(t.item())
    split = None
    hip_f32 = None
    legs = not args.no_split_f16
    sel_leg = None
    if world == 1 and refine is None and dense and gf == 64 and legs and not args.no_cpu_baseline:
        hip_f32 = step()   # outputs of the measured configuration, kept for the parity record
        hip_f32 = {k: hip_f32[k].clone() for k in ("pred_offset", "pred_prob_end", "pair_pred_pos",
                                                   "pred_pos", "max_pair_id")}
        hip_f32["depth"] = depth.clone()
    hip_h = None
    if world == 1 and args.precision == "f32" and refine is None and gf == 64 and legs and args.offsets == "all":
        # the same workload with the offset decoder on the arg-max pair of every ray only (opt-in,
        # lidf_query(offsets="selected")): a labelled secondary rate beside the headline, with the FLOP its two
        # launches of the per-point kernel issue (prob net over the P pairs, offset net over the R selected ones)
        for _ in range(args.warmup):
            step(offsets="selected")
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for i in range(args.steps):
            got = step(pairs[i], offsets="selected")
        torch.cuda.synchronize()
        el = time.perf_counter() - ts
        kms = sum(ev.elapsed_ms(a, b) for a, b in pairs) / args.steps
        fl = F_EXEC_PROB * P + F_EXEC_OFF * scene["R"]
        sel_leg = {"what": "opt-in lidf_query(offsets='selected'): offset_dec on the arg-max pair of every ray only; "
                           "pred_prob_end, softmax, max_pair_id, pred_pos and depth bit-identical to the headline "
                           "step, pred_offset / pair_pred_pos defined at the selected pairs only (NaN elsewhere) — "
                           "NOT the reference's data flow, never the headline",
                   "value": round(P * args.steps / el / 1e6, 2), "unit": "Mpoints/s",
                   "ms_per_step": round(el / args.steps * 1e3, 4),
                   "kernel": "lidf_points_fused_kernel x 2 (prob net over P pairs | offset net over R selected pairs) "
                             "+ lidf_ray_reduce_kernel + lidf_selected_finish_kernel between the events",
                   "kernel_ms": round(kms, 4), "flop_issued": fl,
                   "achieved": round(fl / (kms * 1e-3) / 1e12, 2), "peak": PEAK_F32_TFLOPS, "unit_roofline": "TFLOP/s",
                   "frac": round(fl / (kms * 1e-3) / 1e12 / PEAK_F32_TFLOPS, 4)}
        if hip_f32 is not None and "pred_pos" in hip_f32:
            sel_leg["bit_identical_to_headline"] = {
                k: bool(torch.equal(got[k].reshape(-1), hip_f32[k].reshape(-1)))
                for k in ("pred_prob_end", "pred_pos", "max_pair_id")}
            sel_leg["bit_identical_to_headline"]["depth"] = bool(torch.equal(depth, hip_f32["depth"]))
            m_ = hip_f32["max_pair_id"].clamp(max=P - 1)
            sel_leg["bit_identical_to_headline"]["pred_offset_at_selected_pairs"] = bool(torch.equal(
                got["pred_offset"].reshape(-1)[m_], hip_f32["pred_offset"].reshape(-1)[m_]))
    if world == 1 and args.precision == "f32" and refine is None and gf == 64 and legs and args.offsets == "all":
        if hip_f32 is None:
            hip_f32 = step()
            hip_f32 = {k: hip_f32[k].clone() for k in ("pred_offset", "pred_prob_end", "pair_pred_pos")}
        # the same workload through the split-f16 kernel: rate, and deviation from the f32 outputs
        for _ in range(args.warmup):
            step(precision="f16x3")
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for i in range(args.steps):
            got = step(pairs[i], precision="f16x3")
        torch.cuda.synchronize()
        el = time.perf_counter() - ts
        kms = sum(ev.elapsed_ms(a, b) for a, b in pairs) / args.steps
        hip_h = {k: got[k].clone() for k in ("pred_offset", "pred_prob_end", "pair_pred_pos",
                                             "pred_pos", "max_pair_id")}
        hip_h["depth"] = depth.clone()
        split = {"value": round(P * args.steps / el / 1e6, 2), "unit": "Mpoints/s",
                 "ms_per_step": round(el / args.steps * 1e3, 4), "dtype": DTYPE_F16X3,
                 "kernel": "lidf_points_h_kernel", "kernel_ms": round(kms, 4),
                 "achieved": round(F_EXEC_H * P / (kms * 1e-3) / 1e12, 2),
                 "peak": PEAK_F16_TFLOPS, "unit_roofline": "TFLOP/s",
                 "frac": round(F_EXEC_H * P / (kms * 1e-3) / 1e12 / PEAK_F16_TFLOPS, 4),
                 "max_abs_diff_vs_f32": {k: float((hip_h[k] - hip_f32[k]).abs().max())
                                         for k in ("pred_offset", "pred_prob_end", "pair_pred_pos")}}
    if rank == 0:
        value = points_all * args.steps / elapsed / 1e6
        h16 = args.precision == "f16x3"
        peak = PEAK_F16_TFLOPS if h16 else PEAK_F32_TFLOPS
        f_exec = F_EXEC_H if h16 else F_EXEC
        kname = "lidf_points_h_kernel" if h16 else ("lidf_points_fused_kernel" if gf == 64 else "lidf_chain16_kernel")
        # rocprofv3 legs of this very command (child processes after the timed region). Every N = 1 query run
        # gets the kernel trace; the HBM counter passes run for the default headline command (the line the
        # driver records) and wherever --pmc asks for them. Nothing is read from committed files.
        live = None
        profiled = any(k.startswith(("ROCP_", "ROCPROF")) for k in os.environ)   # already under a profiler
        headline = (dense and B == 1 and N == 64 and refine is None and not h16 and S == 1 and gf == 64
                    and args.offsets == "all")
        if world == 1 and not use_dist and not args.no_rocprof and not profiled:
            live = live_profile(sys.argv[1:], kname, pmc=args.pmc or (headline and not args.no_cpu_baseline),
                                per_step=(2 if (args.offsets == "selected" and gf == 64 and not h16) else
                                          chain_launches_per_step(gf, P) if gf != 64 else 1))
        lk = live_kernel(live, kname)
        lh = (live or {}).get("hbm")
        lhk = None
        if lh:
            lhk = next((v for k, v in lh["kernels"].items() if kname in k), None)
        lmk = next((v for k, v in ((live or {}).get("mfma") or {"kernels": {}})["kernels"].items() if kname in k), None)
        ach = f_exec * P / (kern_ms * 1e-3) / 1e12          # MFMA FLOP issued / time
        ach_alg = F_ALG * P / (kern_ms * 1e-3) / 1e12       # reference-formulation FLOP / time
        # algorithmic HBM bytes of the fused query: 20 B/point in+out, 32 B/ray,
        # per frame the feature map + voxel features + weights
        bytes_alg = 20.0 * P + 32.0 * scene["R"] + B * (32 * hl * w * 4 + 729 * 128 * 4) + 1136776
        line = {
            "metric": "Mpoints/sec implicit-MLP query, 240x320x%d samples; depth L1 vs ref" % N,
            "value": round(value, 2), "unit": "Mpoints/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if by_rays else "weak", "vs_baseline": None,
            "dtype": DTYPE_F16X3 if h16 else "f32", "data": "synthetic",
            "config": {"workload": "%s: %d x 240x320 frame(s) per GPU, %d candidates/ray, LIDF "
                                   "stage-1 fused query (ROI + PE + prob_dec IMNet + offset_dec IEF n_iter=2 "
                                   "+ per-ray softmax/argmax + depth)%s%s" %
                                   (config_name(args, refine is not None), B, N,
                                    " + stage 2 (2 x get_pred_refine: PointNet2Stage over 10,000 valid + "
                                    "76,800 predicted points, IEF D=334)" if refine is not None else "",
                                    "; RCCL all-gather of depth maps" if use_dist else ""),
                       "rays_per_gpu": scene["R"], "points_per_gpu": P, "voxels": scene["V"],
                       "streams": S,
                       "parallelism": ("image rows of one frame sharded over %d GPU(s), depth rows all-gathered"
                                       if by_rays else "frames sharded over %d GPU(s)") % world},
            # frac = MFMA FLOP the kernel ISSUES (counted from its instruction stream, >99 % useful
            # MACs) / its HIP-event time / dense peak: a hardware utilisation. The reference
            # formulation needs 2.4x more FLOP per point (no layer-1 factorisation, no IEF hoist):
            # that model-FLOPs rate is reported separately as achieved_alg / frac_alg.
            "roofline": {"bound": "mfma", "achieved": round(ach, 2), "peak": peak,
                         "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                         # HBM bytes of one launch of the kernel by the PMC counters of this run (null: no pass)
                         "traffic": lhk["bytes_corrected"] if lhk else None,
                         "kernel": kname, "kernel_ms": round(kern_ms, 4),
                         # (rocprofv3 --kernel-trace of this same command, run after the timed region: average
                         # duration of the kernel's launches after its first / of all of them)
                         "kernel_ms_rocprof_live": ({"kernel_ms": lk["avg_ms_after_first"], "kernel_ms_all": lk["avg_ms"],
                                                     "calls": live["steps_seen"], "command": live["command"]}
                                                    if lk else None),
                         # the same issued FLOP over the profiler's average duration (after the first launch / all)
                         "frac_rocprof": (round(f_exec * P / (lk["avg_ms_after_first"] * 1e-3) / 1e12 / peak, 4)
                                          if lk and args.offsets == "all" else None),
                         "frac_rocprof_all": (round(f_exec * P / (lk["avg_ms"] * 1e-3) / 1e12 / peak, 4)
                                              if lk and args.offsets == "all" else None),
                         "flop_per_point_exec": f_exec, "flop_per_point_alg": F_ALG,
                         "flop_per_point_counter": (round(lmk["mfma_flop"] / P, 1) if lmk and not h16 else None),
                         # which clock each fraction is on
                         "clocks": {"frac": "HIP events recorded by the library on the launch stream around the kernel, "
                                            "timed run",
                                    "frac_rocprof": "rocprofv3 --kernel-trace child run of this command, launches "
                                                    "after the kernel's first",
                                    "pipe.ghz": "the rocprofv3 --pmc child run's shader clock (SQ_BUSY_CYCLES / 32 / "
                                                "duration there), NOT the timed run's"},
                         # (ghz_timed_implied: the clock at which the timed run's HIP-event duration holds the same
                         # busy cycles the counter run saw — ghz x counter-run duration / HIP-event duration)
                         "pipe": ({"ghz": lmk["ghz"], "mfma_busy": lmk["mfma_busy"], "kernel_us_pmc_run": lmk["us"],
                                   "ghz_timed_implied": round(lmk["ghz"] * lmk["us"] / (kern_ms * 1e3), 3)}
                                  if lmk else None),
                         "achieved_alg": round(ach_alg, 2), "frac_alg": round(ach_alg / peak, 4)},
            "hbm": {"bytes_alg": round(bytes_alg), "bytes_counter": lh["bytes_per_step"] if lh else None,
                    "gbps_alg": round(bytes_alg / (elapsed / args.steps) / 1e9, 2),
                    "gbps_counter": (round(lh["bytes_per_step"] / (elapsed / args.steps) / 1e9, 2) if lh else None),
                    "counter_source": lh["command"] if lh else None,
                    "peak_gbps": 8000.0,
                    "frac_of_8TBs": round(bytes_alg / (elapsed / args.steps) / 8e12, 5),
                    "note": "whole step; the path is MFMA-bound (>= 3,000 FLOP per HBM byte)"},
        }
        if rpairs:
            # stage 2 (2 x get_pred_refine) on its own: the IEF rows kernel (lidf_points_kernel<ROWS_GATHER>) and
            # the PointNet2Stage pass (two register chains: 44 + 444 matrix instructions per 32 points +
            # the per-voxel layers), HIP events recorded by lidf_refine_profile_f32 on the launch stream
            t_pn = sum(ev.elapsed_ms(it[0], it[1]) for st_ in rpairs for it in st_) / args.steps / 2
            t_ief = sum(ev.elapsed_ms(it[2], it[3]) for st_ in rpairs for it in st_) / args.steps / 2
            R = scene["R"]
            # issued per 32 rays and iteration: 7 layer-1 k-quads (embed(pos); the ROI / direction columns are
            # a per-ray product formed once per call) x 8 tiles x 4 + 2 passes x 654 v_mfma_f32_32x32x2
            ief16 = os.environ.get("LIDF_IEF16", "1") != "0"
            f_ief = F_IEF16 if ief16 else F_IEF32
            n_pn = refine.n_valid + R
            f_pn = (44 + 444) * 4096 / 32.0                          # issued FLOP per PointNet point
            a_ief, a_pn = f_ief * R / (t_ief * 1e-3) / 1e12, f_pn * n_pn / (t_pn * 1e-3) / 1e12
            wt = (R + 31) // 32
            line["roofline_stage2"] = {
                "refine_ms_per_step": round(elapsed / args.steps * 1e3 - 0.0, 4),
                "ief_rows": {"kernel": "lidf_ief16_kernel" if ief16 else "lidf_points_kernel<LIDF_MODE_ROWS_GATHER>", "bound": "mfma",
                             "kernel_ms": round(t_ief, 4), "launches_per_step": 2,
                             "flop_per_ray_exec": f_ief, "achieved": round(a_ief, 2), "peak": PEAK_F32_TFLOPS,
                             "unit": "TFLOP/s", "frac": round(a_ief / PEAK_F32_TFLOPS, 4),
                             "wave_tiles": wt,
                             "rounds": ("%d 16-ray sub-tiles over 1024 wavefronts = %.2f -> %d half-rounds"
                                        % (2 * wt, 2 * wt / 1024.0, -(-2 * wt // 1024)) if ief16 else
                                        "%d wave-tiles over 1024 SIMDs = %.2f -> %d rounds"
                                        % (wt, wt / 1024.0, -(-wt // 1024)))},
                "pointnet": {"kernel": "lidf_pointnet_chain_kernel<1|2> + lidf_vox2_kernel (per-voxel layers)",
                             "bound": "mfma / atomic max-pool", "ms": round(t_pn, 4), "launches_per_step": 2,
                             "points": n_pn, "flop_per_point_exec": f_pn, "achieved": round(a_pn, 2),
                             "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": round(a_pn / PEAK_F32_TFLOPS, 4)},
            }
            del line["roofline_stage2"]["refine_ms_per_step"]
        if live:
            line["profile"] = {k: live[k] for k in ("command", "launches_per_step", "launches_per_step_steady",
                                                    "busy_ms_per_step", "busy_ms_per_step_after_first", "kernels")}
            if lh:
                line["profile"]["hbm"] = lh
            if live.get("mfma"):
                line["profile"]["mfma"] = live["mfma"]
        if gather_ok is not None:
            line["collective"] = {"op": ("all_gather_into_tensor (RCCL) of the depth rows of one [%d,%d] map" % (h, w)
                                         if by_rays else
                                         "all_gather_into_tensor (RCCL) of [%d,%d,%d] f32 depth maps per rank" % (B, h, w)),
                                  "inside_timed_region": True, "gathered_equals_local": gather_ok,
                                  "ranks_seen": ranks_seen, "ms_per_step_by_rank": rank_ms,
                                  "points_all_ranks": points_all}
        if split is not None:
            line["split_f16"] = split
        if sel_leg is not None:
            line["offsets_selected"] = sel_leg
        if args.offsets == "selected":   # `--offsets selected`: the opt-in mode as the measured step (secondary record)
            fl = F_EXEC_PROB * P + F_EXEC_OFF * scene["R"]
            a_ = fl / (kern_ms * 1e-3) / 1e12
            line["metric"] = "Mpoints/sec implicit-MLP query, offsets for the selected pairs only (opt-in)"
            line["config"]["workload"] = "secondary (opt-in lidf_query(offsets='selected'), not the reference's data " \
                "flow: pred_offset / pair_pred_pos defined at the arg-max pairs only): " + line["config"]["workload"]
            rl = line["roofline"]
            rl.update({"achieved": round(a_, 2), "frac": round(a_ / peak, 4), "flop_issued_per_step": fl,
                       "kernel": "lidf_points_fused_kernel x 2 (prob net over P pairs | offset net over R selected "
                                 "pairs) + lidf_ray_reduce_kernel + lidf_selected_finish_kernel between the events",
                       "flop_per_point_exec": None, "achieved_alg": None, "frac_alg": None,
                       "frac_rocprof": (round(fl / (lk["avg_ms_after_first"] * lk["calls_per_step"] * 1e-3) / 1e12 / peak, 4)
                                        if lk else None),
                       "frac_rocprof_note": "issued FLOP of both launches / (their profiler average x launches per step)"})
        if not dense:
            line["config"]["pairs"] = {"kind": args.pairs, "points": P, "rays": scene["R"],
                                       "pairs_per_ray": round(P / scene["R"], 3)}
            line["config"]["workload"] = line["config"]["workload"].replace(
                "configs[1]", "secondary (not the headline shape): %s candidate list" % args.pairs)
            line["metric"] = "Mpoints/sec implicit-MLP query, %s candidate list" % args.pairs
        if gf != 64:
            # FLOP of the layer-by-layer formulation per pair (generic.query): layer 1 over the 2E pair columns, the IEF's
            # 16 offset-encoding columns per pass, layers 2-4 of every pass; the per-voxel / per-ray tables are noise
            h1, h2, h3, e2 = 4 * gf, 2 * gf, gf, 2 * (3 + 6 * 8)
            tail = 2 * (h1 * h2 + h2 * h3 + h3)
            fl_pt = 2 * (2 * e2 * h1) + tail + 2 * (2 * 16 * h1 + tail)
            a_ = fl_pt * P / (elapsed / args.steps) / 1e12
            from implicit_depth_amd import generic as _gen
            chained = _gen.chain_ok(prob) and _gen.chain_ok(off)
            line["roofline"] = {"bound": "mfma", "achieved": round(a_, 2), "peak": peak, "unit": "TFLOP/s",
                                "frac": round(a_ / peak, 4), "traffic": None,
                                "kernel": ("lidf_chain16_kernel<gf / 16, ..> once per decoder and slab" if chained else
                                           "lidf_linear_kernel<NT> per layer") + " (whole step, between the events)",
                                "flop_per_point_exec": fl_pt,
                                "clocks": "frac = the formulation's FLOP over the WHOLE step (wall clock between syncs), "
                                          "not a kernel's duration",
                                "basis": "FLOP of the executed formulation (factorised layer 1, 3 decoder passes), "
                                         "not an instruction count; the shipped width issues %d per point" % F_EXEC}
            if chained and lk:
                # the chain launch itself, on the profiler's clock: issued 16 x 16 x 4 matrix instructions per 16-row
                # sub-tile (layer 1 over ceil(2E / 16) k-quads, per pass the bias / u steps and layers 2-3), both decoders
                G = gf // 16
                t1, t2, t3 = 4 * G, 2 * G, G
                l1m = ((e2 + 15) // 16) * t1 * 4
                pm = 4 * ((t2 + 3) // 4) + t1 + t1 * t2 * 4 + min(4, t3) * ((t3 + 3) // 4) + t2 * t3 * 4
                mf = (l1m + pm) + (l1m + 2 * pm)          # IMNet + IEF n_iter 2
                per_step_ms = lk["avg_ms_after_first"] * lk["calls_per_step"]
                line["roofline"]["chain_kernel"] = {
                    "mfma_16x16x4_per_16_rows_both_decoders": mf, "calls_per_step": lk["calls_per_step"],
                    "ms_per_step_rocprof": round(per_step_ms, 4),
                    "frac_rocprof": round(mf * 2048.0 * (P / 16.0) / (per_step_ms * 1e-3) / 1e12 / peak, 4)}
            line["config"]["workload"] = ("secondary: imnet_gf %d (not a shipped width): get_embedding + get_pred with layer 1 "
                                          "factorised into per-voxel / per-ray tables + the pair's position embedding, %s, "
                                          "pairs in slabs (implicit_depth_amd/generic.py), %d x 240x320 frame(s), %d "
                                          "candidates/ray" % (gf, "each decoder ONE register-chained launch per slab "
                                                              "(lidf_decoder_chain_f32)" if chained else
                                                              "the decoders layer by layer (LIDF_CHAIN16=0)", B, N))
            line["metric"] = "Mpoints/sec implicit-MLP query at imnet_gf %d (%s)" % (gf, "chain launch" if chained
                                                                                    else "layer by layer")
        if world == 1 and not args.no_cpu_baseline and dense and gf == 64:
            cb, ref = cpu_baseline(scene)
            line["cpu_baseline"] = cb
            if hip_f32 is not None:
                line["parity"] = parity_record(hip_f32, ref, scene, hip_f32["depth"])
                if hip_h is not None:
                    ph = parity_record(hip_h, ref, scene, hip_h["depth"])
                    line["split_f16"]["parity"] = {k: ph[k] for k in ("depth_l1", "max_abs", "ok")}
        emit(line)
    if use_dist:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
