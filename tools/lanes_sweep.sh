# bench at 1, 2, 3, 4 lanes (64 steps, no CPU baseline): value / single-context value / stage table
for l in 1 2 3 4; do
  python bench.py --steps 64 --lanes $l --no-cpu-baseline > gpurun_out/r2_lanes$l.json 2> gpurun_out/r2_lanes$l.err || { tail -5 gpurun_out/r2_lanes$l.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/r2_lanes$l.json')); print('lanes', $l, 'value', d['value'], 'single', d['value_single_context'], 'ms/step', d['ms_per_step'], d['stage_ms_per_step'])"
done
