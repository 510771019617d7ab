#!/usr/bin/env bash
# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): lanes whose streams share one are serialised
cd $GRAFT_REPO_ROOT
for q in 4 8 16; do for l in 3 4 5 6; do
  echo -n "== GPU_MAX_HW_QUEUES=$q lanes $l : "
  env GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python bench.py --lanes $l --no-cpu-baseline --no-profile --no-single 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
