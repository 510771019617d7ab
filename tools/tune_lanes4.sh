#!/usr/bin/env bash
# four lanes on eight hardware queues: chunk / refill / leafK / workgroups per CU of the persistent kernels (RT355_TUNE sets extend, connect, BVH4 alike)
cd $GRAFT_REPO_ROOT
run() { echo -n "== $* : "; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-profile --no-single 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run A=default
run RT355_TUNE=112,24,6,16,2
run RT355_TUNE=64,24,6,16,2
run RT355_TUNE=160,24,6,16,2
run RT355_TUNE=112,16,6,16,2
run RT355_TUNE=112,32,6,16,2
run RT355_TUNE=112,24,6,8,2
run RT355_TUNE=112,24,6,24,2
run RT355_TUNE=112,24,4,16,2
run RT355_TUNE=112,24,8,16,2
run RT355_TUNE=112,24,6,16,1
run RT355_SHADE_TILE=512
run A=default
