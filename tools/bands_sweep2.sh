#!/usr/bin/env bash
# experiment 30 again with eight hardware queues (bench.py sets GPU_MAX_HW_QUEUES=8): one frame as K interleaved row-band contexts
cd $GRAFT_REPO_ROOT
run() { echo -n "== $* : "; timeout -k 10 300 python bench.py --steps 64 --no-cpu-baseline --no-profile --no-single "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['contexts'], 'contexts', d['value'], d['ms_per_step'])"; }
run --lanes 1
run --lanes 4
for r in 540 360 270 216 180 136; do run --shard ibands --band-rows $r; run --shard ibands --band-rows $r --persist-blocks 3; done
