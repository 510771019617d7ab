# round-2 evidence run on the GPU box: default bench line, rocprofv3 kernel stats of the default command, one-context bench,
# per-bounce table, the five configurations, the gather probe.  Outputs under gpurun_out/r2_final/.
set -e
O=gpurun_out/r2_final; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --lanes 1 --no-cpu-baseline > $O/bench_lanes1.json 2> $O/bench_lanes1.err
python bench.py --accel bvh4 --no-cpu-baseline > $O/bench_bvh4.json 2> $O/bench_bvh4.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err
bash tools/trace_default.sh r2_final/trace_default --no-single
bash tools/trace.sh r2_final/trace_lanes1 --lanes 1
python tools/config_table.py > $O/config_table.log 2>&1
hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o /tmp/gather_probe.bin && for n in 8 12 14 16 17 19 21; do /tmp/gather_probe.bin $n; done > $O/gather_probe.log 2>&1
python -c "
import json
for f in ('bench','bench_lanes1','bench_bvh4','bench_driver_flags'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d.get('value_single_context'), d['ms_per_step'], d['repeats'], d['roofline']['frac'], d['roofline']['gather']['frac'], d['roofline'].get('dram_frac'))
"
