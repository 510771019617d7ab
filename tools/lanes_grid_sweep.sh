#!/usr/bin/env bash
# more than three sample streams per GPU, with smaller persistent grids (5th RT355_TUNE field = workgroups per CU): bench value
cd $GRAFT_REPO_ROOT
for l in 3 4 5 6; do
  for d in 4 3 2; do
    echo -n "== lanes $l persist blocks/CU $d : "
    env RT355_TUNE=112,24,6,16,$d timeout -k 10 200 python bench.py --lanes $l --no-cpu-baseline --no-profile --no-single 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
