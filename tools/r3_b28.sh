O=gpurun_out/r3_b28; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputests.log 2>&1 || { grep -E "^FAILED|^ERROR" $O/gputests.log | head -30; }
tail -2 $O/gputests.log
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py > $O/bench.json 2> $O/bench.err || tail -5 $O/bench.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/err.txt || tail -5 $O/err.txt
python bench.py --lanes 1 --no-cpu-baseline > $O/bench_lanes1.json 2> $O/err.txt || tail -5 $O/err.txt
python bench.py --config 2 --no-cpu-baseline > $O/bench_config2.json 2> $O/err.txt || tail -5 $O/err.txt
python bench.py --config 4 --no-cpu-baseline > $O/bench_config4.json 2> $O/err.txt || tail -5 $O/err.txt
for f in bench bench_driver_flags bench_lanes1 bench_config2 bench_config4; do python -c "
import json; d=json.load(open('$O/$f.json')); r=d['roofline']; print('%-22s value %8.1f single %s lanes %s ms/step %.4f frac %s valu %s job %s' % ('$f', d['value'], d['value_single_context'], d['config']['lanes'], d['ms_per_step'], r.get('frac'), (r.get('valu_issue') or {}).get('frac'), (r.get('job') or {}).get('frac')))"; done



find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
python bench.py --config 5 --steps 96 --no-cpu-baseline > $O/bench_config5.json 2> $O/err.txt || tail -5 $O/err.txt
