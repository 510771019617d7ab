#!/usr/bin/env bash
# session 2, call A: experiment 29 repeated (static round-robin chunks), then the band sweep
cd $GRAFT_REPO_ROOT
{
bash tools/tune_single.sh RT355_TAIL=0,0 RT355_TAIL=1,1 RT355_TAIL=0,0 RT355_TAIL=1,1 RT355_TAIL=1,0
for t in RT355_TAIL=0,0 RT355_TAIL=1,1 RT355_TAIL=0,0 RT355_TAIL=1,1; do
  echo -n "== lanes 3 $t : "
  env $t timeout -k 10 200 python bench.py --lanes 3 --no-cpu-baseline --no-profile --no-single 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
} > gpurun_out/r2_tail2.log 2>&1
cat gpurun_out/r2_tail2.log
bash tools/bands_sweep.sh 2>&1 | tee gpurun_out/r2_bands.log
