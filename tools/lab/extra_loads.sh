#!/usr/bin/env bash
# lab: librt355.so rebuilt ON THE BOX'S COPY with -DRT355_EXTRA_LOADS=n: n extra 4-byte load requests (plain cached loads by inline asm) per node event of k_trace_persist
# (to the record the event fetches anyway: same cache line, no new data) - how much does a vector-memory REQUEST cost the traversal?
cd $GRAFT_REPO_ROOT
for n in 0 1 4 0; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function \
    -DRT355_EXTRA_LOADS=$n magr_ray_tracer_amd/csrc/rt355.hip -o magr_ray_tracer_amd/librt355.so || exit 1
  for l in 4 1; do
    echo -n "== extra loads $n lanes $l : "
    timeout -k 10 200 python bench.py --lanes $l --no-cpu-baseline --no-profile --no-single 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['accum_rgb_sum'])"
  done
done
