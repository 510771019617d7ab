#!/usr/bin/env bash
# evidence refresh after rt_share_scene: default bench line, the driver's flags, rocprofv3 kernel stats of the default command
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_final5; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err
python bench.py --accel bvh4 --no-cpu-baseline > $O/bench_bvh4.json 2> $O/bench_bvh4.err
bash tools/trace_default.sh r2_final5/trace_default --no-single
python -c "
import json
for f in ('bench','bench_driver_flags','bench_bvh4'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d.get('value_single_context'), d['ms_per_step'], d['repeats'], d['roofline']['frac'], d['roofline']['gather']['frac'], d['roofline'].get('dram_frac'), d['roofline']['avg_launch_ms'])
"
