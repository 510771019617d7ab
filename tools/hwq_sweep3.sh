#!/usr/bin/env bash
cd $GRAFT_REPO_ROOT
run() { echo -n "== $* : "; timeout -k 10 200 python bench.py "$@" --no-cpu-baseline --no-profile --no-single 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run --lanes 4 --persist-blocks 3
run --lanes 4 --persist-blocks 2
run --lanes 4 --persist-blocks 4
run --lanes 5 --persist-blocks 3
run --lanes 6 --persist-blocks 3
run --lanes 6 --persist-blocks 2
run --lanes 8 --persist-blocks 3
run --lanes 8 --persist-blocks 2
run --lanes 3 --persist-blocks 4
run --lanes 3 --persist-blocks 3
run --lanes 4 --persist-blocks 3
