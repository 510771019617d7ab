#!/usr/bin/env bash
cd $GRAFT_REPO_ROOT
run() { echo -n "== $* : "; env "${@:2}" timeout -k 10 200 python bench.py --lanes $1 --no-cpu-baseline --no-profile --no-single 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run 4 GPU_MAX_HW_QUEUES=8
run 4 GPU_MAX_HW_QUEUES=8 RT355_TUNE=112,24,6,16,3
run 4 GPU_MAX_HW_QUEUES=8 RT355_TUNE=112,24,6,16,5
run 4 GPU_MAX_HW_QUEUES=8 RT355_TUNE=112,24,6,16,7
run 4 GPU_MAX_HW_QUEUES=8 RT355_TUNE=112,24,6,8,4
run 4 GPU_MAX_HW_QUEUES=8 RT355_SHADE_PER_CU=2
run 7 GPU_MAX_HW_QUEUES=8
run 8 GPU_MAX_HW_QUEUES=16
run 8 GPU_MAX_HW_QUEUES=16 RT355_TUNE=112,24,6,16,3
run 12 GPU_MAX_HW_QUEUES=16
run 2 GPU_MAX_HW_QUEUES=8
run 3 GPU_MAX_HW_QUEUES=8
run 4 GPU_MAX_HW_QUEUES=8
