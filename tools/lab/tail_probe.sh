#!/usr/bin/env bash
# lab: rebuild librt355.so with -DRT355_TAIL_PROBE ON THE BOX'S COPY (nothing is merged back but gpurun_out/) and print, for every
# persistent launch of one bench frame, when the queue ran dry and when the waves exited (k_trace_persist only).
cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function \
  -DRT355_TAIL_PROBE magr_ray_tracer_amd/csrc/rt355.hip -o magr_ray_tracer_amd/librt355.so || exit 1
for env in "RT355_FIXED_CHUNKS=1,1" "RT355_FIXED_CHUNKS=0,0" "RT355_COHERENT=0"; do
  echo "#### $env"
  env $env python tools/lab/tail_probe.py
done
