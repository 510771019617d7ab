O=gpurun_out/r3_b10; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputests.log 2>&1 || { grep -E "^FAILED|^ERROR" $O/gputests.log | head -30; }
tail -2 $O/gputests.log
python tools/deep_fuzz.py 120000 600 mixed > $O/fuzz_mixed.txt 2>&1; tail -1 $O/fuzz_mixed.txt
python bench.py --config 4 --no-cpu-baseline > $O/bench_config4.json 2> $O/err.txt || tail -5 $O/err.txt
python bench.py --config 4 --no-cpu-baseline > $O/bench_config4_b.json 2> $O/err.txt || tail -5 $O/err.txt
for f in bench_config4 bench_config4_b; do python -c "
import json; d=json.load(open('$O/$f.json')); print('config 4: value %8.1f single %s stages %s' % (d['value'], d['value_single_context'], d['stage_ms_per_step']))"; done
