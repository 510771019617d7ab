#!/usr/bin/env bash
# final check of the round: whole test suite on the GPU box (CPU + GPU marks), smoke, default bench line
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_final4; mkdir -p $O
python -m pytest tests -q -m "not gpu" > $O/cpu_tests.log 2>&1 || { tail -30 $O/cpu_tests.log; exit 1; }
tail -1 $O/cpu_tests.log
python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -1 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench_driver_flags.json')); print('bench', d['value'], d['value_single_context'], d['ms_per_step'], d['repeats'], d['roofline']['frac'], d['cpu_baseline']['value'])"
