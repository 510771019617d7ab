O=gpurun_out/r3_b26; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputests.log 2>&1 || { grep -E "^FAILED|^ERROR" $O/gputests.log | head -30; }
tail -2 $O/gputests.log
python -c "import __graft_entry__ as g; g.smoke()"
python tools/deep_fuzz.py 180000 1500 > $O/fuzz_soups.txt 2>&1; tail -1 $O/fuzz_soups.txt
python tools/deep_fuzz.py 190000 600 mixed > $O/fuzz_mixed.txt 2>&1; tail -1 $O/fuzz_mixed.txt
python bench.py > $O/bench.json 2> $O/bench.err || tail -5 $O/bench.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/err.txt || tail -5 $O/err.txt
python bench.py --lanes 1 --no-cpu-baseline > $O/bench_lanes1.json 2> $O/err.txt || tail -5 $O/err.txt
python bench.py --config 2 --no-cpu-baseline > $O/bench_config2.json 2> $O/err.txt || tail -5 $O/err.txt
python bench.py --config 4 --no-cpu-baseline > $O/bench_config4.json 2> $O/err.txt || tail -5 $O/err.txt
for f in bench bench_driver_flags bench_lanes1 bench_config2 bench_config4; do python -c "
import json; d=json.load(open('$O/$f.json')); r=d['roofline']; print('%-22s value %8.1f single %s lanes %s ms/step %.4f frac %s valu %s job %s' % ('$f', d['value'], d['value_single_context'], d['config']['lanes'], d['ms_per_step'], r.get('frac'), (r.get('valu_issue') or {}).get('frac'), (r.get('job') or {}).get('frac')))"; done
bash tools/trace_default.sh r3_b26/trace_default --no-single > $O/trace_default.txt 2>&1; tail -3 $O/trace_default.txt | cut -c1-180
bash tools/trace.sh r3_b26/trace_lanes1 --lanes 1 --no-single --no-repeat > $O/trace_lanes1.txt 2>&1; tail -7 $O/trace_lanes1.txt
bash tools/trace.sh r3_b26/trace_c2 --config 2 --lanes 1 --no-single --no-repeat > $O/trace_c2.txt 2>&1; tail -7 $O/trace_c2.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
