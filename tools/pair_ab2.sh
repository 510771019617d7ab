# pair fetch with grids that match the occupancy of the PF kernels (6 / 5 workgroups per CU)
set -e
O=gpurun_out/r2_pair; mkdir -p $O
run() { # label, env...
  env "${@:2}" python bench.py --steps 64 --lanes 1 --no-cpu-baseline > $O/t.json 2> $O/t.err || { tail -5 $O/t.err; exit 1; }
  python -c "
import json; d=json.load(open('$O/t.json')); print('$1', 'value', d['value'], 'ms/step', d['ms_per_step'], d.get('stage_ms_per_step'))"
}
run "plain  extend 6/CU" RT355_PAIRFETCH=0 RT355_TUNE=112,24,6,8,6
run "PF     extend 6/CU" RT355_PAIRFETCH=1 RT355_TUNE=112,24,6,8,6
run "PF     extend 5/CU" RT355_PAIRFETCH=1 RT355_TUNE=112,24,6,8,5
run "plain connect 5/CU" RT355_PAIRFETCH=0 RT355_TUNE=128,32,6,16,5
run "PF    connect 5/CU" RT355_PAIRFETCH=2 RT355_TUNE=128,32,6,16,5
run "PF    connect 4/CU" RT355_PAIRFETCH=2 RT355_TUNE=128,32,6,16,4
