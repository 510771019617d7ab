#!/usr/bin/env bash
# round-2 end state after experiment 29 (chunks dealt round-robin for a context alone): GPU tests, bench lines, rocprofv3 kernel stats of the
# default command, one-context per-bounce table, the five configurations.  Outputs under gpurun_out/r2_final3/.
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_final3; mkdir -p $O
python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --lanes 1 --no-cpu-baseline > $O/bench_lanes1.json 2> $O/bench_lanes1.err
python bench.py --accel bvh4 --no-cpu-baseline > $O/bench_bvh4.json 2> $O/bench_bvh4.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err
bash tools/trace_default.sh r2_final3/trace_default --no-single
bash tools/trace.sh r2_final3/trace_lanes1 --lanes 1
python tools/config_table.py > $O/config_table.log 2>&1
tail -8 $O/config_table.log
python -c "
import json
for f in ('bench','bench_lanes1','bench_bvh4','bench_driver_flags'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d.get('value_single_context'), d['ms_per_step'], d['repeats'], d['roofline']['frac'], d['roofline']['gather']['frac'], d['roofline'].get('dram_frac'), d['roofline']['single_stream'].get('avg_launch_ms'), d['stage_ms_per_step'])
"
