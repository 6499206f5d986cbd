#!/bin/bash
# Development aid: build ablated variants of liblidf_hip.so (gpurun_out/abl_<mask>.so) — run on
# the authoring box; then `scripts/ablate_run.sh` on the GPU box times each.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out_local
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fvisibility=hidden \
    -DLIDF_ABLATE=$m -I include -I implicit_depth_amd/csrc -o implicit_depth_amd/csrc/abl_$m.so \
    implicit_depth_amd/csrc/lidf_points.hip implicit_depth_amd/csrc/lidf_aux.hip implicit_depth_amd/csrc/lidf_api.hip &
done
wait
ls -la implicit_depth_amd/csrc/abl_*.so
