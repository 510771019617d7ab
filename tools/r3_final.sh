# round-3 evidence run on the end state: GPU suite, bench lines (default, driver flags, one lane, configs 2 / 4 / 5), kernel stats of the default command, per-bounce table
O=gpurun_out/r3_final; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputests.log 2>&1 || { grep -E "^FAILED|^ERROR" $O/gputests.log | head -30; }
tail -2 $O/gputests.log
python bench.py > $O/bench.json 2> $O/bench.err || tail -5 $O/bench.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/err.txt || tail -5 $O/err.txt
python bench.py --lanes 1 --no-cpu-baseline > $O/bench_lanes1.json 2> $O/err.txt || tail -5 $O/err.txt
python bench.py --config 4 --no-cpu-baseline > $O/bench_config4.json 2> $O/err.txt || tail -5 $O/err.txt
python bench.py --config 2 --no-cpu-baseline > $O/bench_config2.json 2> $O/err.txt || tail -5 $O/err.txt
python bench.py --config 5 --steps 64 --no-cpu-baseline > $O/bench_config5.json 2> $O/err.txt || tail -5 $O/err.txt
python bench.py --config 5 --steps 64 --no-cpu-baseline --extend-variant 4 > $O/bench_config5_nested.json 2> $O/err.txt || tail -5 $O/err.txt
for f in bench bench_driver_flags bench_lanes1 bench_config4 bench_config2 bench_config5 bench_config5_nested; do python -c "
import json; d=json.load(open('$O/$f.json')); r=d['roofline']; print('%-22s value %8.1f single %s ms/step %.4f bound %s frac %s job %s util %s' % ('$f', d['value'], d['value_single_context'], d['ms_per_step'], r.get('bound'), r.get('frac'), (r.get('job') or {}).get('frac'), r.get('lane_utilisation')))"; done
bash tools/trace_default.sh r3_final/trace_default --no-single > $O/trace_default.txt 2>&1; tail -12 $O/trace_default.txt | cut -c1-180
bash tools/trace.sh r3_final/trace_lanes1 --lanes 1 --no-single --no-repeat > $O/trace_lanes1.txt 2>&1; tail -7 $O/trace_lanes1.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
