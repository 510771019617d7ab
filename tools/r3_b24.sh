O=gpurun_out/r3_b24; mkdir -p $O
python bench.py --config 2 --no-cpu-baseline > $O/bench_config2.json 2> $O/err.txt || tail -3 $O/err.txt
python bench.py --config 5 --steps 96 --no-cpu-baseline > $O/bench_config5.json 2> $O/err.txt || tail -3 $O/err.txt
for f in bench_config2 bench_config5; do python -c "
import json; d=json.load(open('$O/$f.json')); r=d['roofline']; print('%-22s value %8.1f single %s lanes %s ms/step %.4f bound %s frac %s valu %s job %s / %s' % ('$f', d['value'], d['value_single_context'], d['config']['lanes'], d['ms_per_step'], r.get('bound'), r.get('frac'), (r.get('valu_issue') or {}).get('frac'), (r.get('job') or {}).get('frac'), ((r.get('job') or {}).get('valu_issue') or {}).get('frac')))"; done
python -m pytest tests -m gpu -q > $O/gputests.log 2>&1 || { grep -E "^FAILED|^ERROR" $O/gputests.log | head -30; }
tail -2 $O/gputests.log
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/err.txt || tail -5 $O/err.txt
python -c "
import json; d=json.load(open('$O/bench_driver_flags.json')); print('driver flags: value', d['value'], 'single', d['value_single_context'], 'streams', d['config']['streams_concurrent'])"
