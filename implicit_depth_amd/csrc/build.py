"""Build liblidf_hip.so (gfx950) in-tree with hipcc. No torch involved: the library is a plain
C-ABI shared object (include/lidf_hip.h) loaded through ctypes by implicit_depth_amd._lib."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SOURCES = ["lidf_points.hip", "lidf_points_h.hip", "lidf_rows_h.hip", "lidf_linear.hip", "lidf_aux.hip", "lidf_refine.hip", "lidf_train.hip", "lidf_api.hip"]
HEADERS = ["lidf_device.h", os.path.join(ROOT, "include", "lidf_hip.h")]
LIB = os.path.join(HERE, "liblidf_hip.so")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(HERE, s) for s in SOURCES] + [
        h if os.path.isabs(h) else os.path.join(HERE, h) for h in HEADERS
    ]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [
        hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
        "-ffp-contract=off", "-fvisibility=hidden",
        # the decoder pass is one 168-step fully unrolled software pipeline; clang's default
        # 16k-instruction cap on `#pragma unroll` would silently leave it rolled (arrays in scratch)
        "-mllvm", "-pragma-unroll-threshold=8000000",
        "-I", os.path.join(ROOT, "include"),
        "-I", HERE, "-o", LIB,
    ] + [os.path.join(HERE, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
