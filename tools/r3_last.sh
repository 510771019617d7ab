O=gpurun_out/r3_last; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputests.log 2>&1 || { grep -E "^FAILED|^ERROR" $O/gputests.log | head -30; }
tail -2 $O/gputests.log
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/err.txt || tail -5 $O/err.txt
python -c "
import json; d=json.load(open('$O/bench_driver_flags.json')); r=d['roofline']; print('driver flags: value', d['value'], 'single', d['value_single_context'], 'streams', d['config']['streams_concurrent'], 'roofline', r['bound'], r['frac'], r['avg_launch_ms'], 'cpu', d['cpu_baseline']['value'])"
bash tools/trace_default.sh r3_last/trace_c4 --config 4 --no-single --steps 64 > $O/trace_c4.txt 2>&1; tail -8 $O/trace_c4.txt | cut -c1-150
find $O -name "*kernel_trace.csv" -delete
