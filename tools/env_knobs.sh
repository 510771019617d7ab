#!/usr/bin/env bash
# ROCm runtime knobs on the bench (one context / four lanes)
cd $GRAFT_REPO_ROOT
run() { for l in 1 4; do echo -n "== $* lanes $l : "; env "$@" timeout -k 10 200 python bench.py --lanes $l --no-cpu-baseline --no-profile --no-single 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; }
run A=default
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run HSA_ENABLE_INTERRUPT=0
run GPU_MAX_HW_QUEUES=16
run AMD_SERIALIZE_KERNEL=0 HIP_LAUNCH_BLOCKING=0
run A=default
