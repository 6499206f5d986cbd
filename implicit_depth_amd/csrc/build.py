"""Build liblidf_hip.so (gfx950) in-tree with hipcc. No torch involved: the library is a plain
C-ABI shared object (include/lidf_hip.h) loaded through ctypes by implicit_depth_amd._lib."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SOURCES = ["lidf_points.hip", "lidf_points_h.hip", "lidf_rows_h.hip", "lidf_linear.hip", "lidf_linear_s.hip", "lidf_linear_x.hip", "lidf_linear_sx.hip", "lidf_aux.hip", "lidf_frame.hip", "lidf_refine.hip", "lidf_train.hip", "lidf_dgrad.hip", "lidf_pointnet.hip", "lidf_pointnet_train.hip", "lidf_ief16.hip", "lidf_chain16.hip", "lidf_api.hip"]
HEADERS = ["lidf_device.h", "lidf_linear_kernel.inc", "lidf_api_refine_train.inc", "lidf_api_pointnet_train.inc", os.path.join(ROOT, "include", "lidf_hip.h")]
LIB = os.path.join(HERE, "liblidf_hip.so")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(HERE, s) for s in SOURCES] + [
        h if os.path.isabs(h) else os.path.join(HERE, h) for h in HEADERS
    ]
    return any(os.path.getmtime(d) > t for d in deps)


_PERF_FLAGS = [
    # MFMA results in architectural VGPRs wherever they fit: a layer's outputs are the next
    # layer's B operands, which must be VGPRs — with accumulators in AGPRs every activation
    # costs a v_accvgpr_read, a VALU slot that f32 MFMA does not hide on gfx950. The decoder
    # pass is ordered k-major so that the live set fits (lidf_points.hip); AGPRs stay as spill space
    ["-mllvm", "-amdgpu-mfma-vgpr-form"],
]


def _probe_flags(hipcc, objdir):
    """Performance-only LLVM options: kept when this compiler knows them (an older LLVM answers
    'Unknown command line argument' and would otherwise block the whole build)."""
    src = os.path.join(objdir, "probe.hip")
    with open(src, "w") as f:
        f.write("#include <hip/hip_runtime.h>\n__global__ void k(float* p) { p[0] = 1.f; }\n")
    keep = []
    base = [hipcc, "--offload-arch=gfx950", "-c", src, "-o", os.path.join(objdir, "probe.o")]
    subprocess.run(base, check=True)    # the compiler itself must work
    for fl in _PERF_FLAGS:
        r = subprocess.run(base + fl, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        if r.returncode == 0:
            keep += fl
        else:
            print("build.py: compiler does not accept %s; building without it (slower kernels)" % " ".join(fl),
                  file=sys.stderr)
    return keep


def build(force=False, verbose=False):
    """Compile every source to its own object (in parallel, only the stale ones) and link."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = [
        "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
        "-ffp-contract=off", "-fvisibility=hidden",
        # the decoder pass is one 168-step fully unrolled software pipeline; clang's default
        # 16k-instruction cap on `#pragma unroll` would silently leave it rolled (arrays in scratch)
        "-mllvm", "-pragma-unroll-threshold=8000000",
    ] + _probe_flags(hipcc, objdir) + ["-I", os.path.join(ROOT, "include"), "-I", HERE]
    flags += os.environ.get("LIDF_EXTRA_HIPCC_FLAGS", "").split()   # (development builds: -DLIDF_PROFILE, A/B macros)
    hdr_t = max(os.path.getmtime(h if os.path.isabs(h) else os.path.join(HERE, h)) for h in HEADERS)
    hdr_t = max(hdr_t, os.path.getmtime(os.path.abspath(__file__)))
    jobs, objs = [], []
    for src in SOURCES:
        sp, op = os.path.join(HERE, src), os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(op)
        if force or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), hdr_t):
            cmd = [hipcc] + flags + ["-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd))
            jobs.append((src, subprocess.Popen(cmd)))
    failed = [src for src, p in jobs if p.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, "hipcc -c " + " ".join(failed))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


EXT_SRC = os.path.join(HERE, "lidf_torch_ext.cpp")
EXT = os.path.join(HERE, "lidf_torch_ext.so")


def build_torch_ext(force=False, verbose=False):
    """The pybind11 torch-extension shim over the C ABI (lidf_torch_ext.cpp): plain g++ against
    torch's headers, linked to liblidf_hip.so beside it ($ORIGIN rpath)."""
    build(force=False)
    deps = [EXT_SRC, LIB, os.path.join(ROOT, "include", "lidf_hip.h")]
    if not force and os.path.exists(EXT) and all(os.path.getmtime(d) <= os.path.getmtime(EXT) for d in deps):
        return EXT
    import sysconfig
    import torch
    ti = os.path.dirname(torch.__file__)
    cmd = [
        os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared",
        "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=lidf_torch_ext",
        "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
        "-I", os.path.join(ti, "include"), "-I", os.path.join(ti, "include", "torch", "csrc", "api", "include"),
        "-I", "/opt/rocm/include", "-I", sysconfig.get_paths()["include"], "-I", os.path.join(ROOT, "include"),
        EXT_SRC, "-o", EXT, "-L", HERE, "-l:liblidf_hip.so", "-L", os.path.join(ti, "lib"),
        "-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-ltorch_python",
        "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(ti, "lib"),
    ]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return EXT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_torch_ext(force="--force" in sys.argv, verbose=True))
