#!/usr/bin/env bash
# lab: connect through the wide quantised tree (RT355_WIDE_CONNECT=1) against the BVH2 any-hit traversal: stage times of one context
cd $GRAFT_REPO_ROOT
for w in 0 1; do
  RT355_WIDE_CONNECT=$w python bench.py --lanes 1 --steps 64 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/ab_wide_$w.json
  python3 - $w <<'PY'
import json, sys
d = json.load(open(f"/tmp/ab_wide_{sys.argv[1]}.json")); c = d["connect_roofline"]
print("wide", sys.argv[1], d["value"], d["stage_ms_per_step"], "connect: records/s", c["gather_records_per_s"], "rays", c["rays_per_launch"], "ms", c["avg_launch_ms"])
PY
done
