# round-3 GPU run 6: PMC for config 4; bench lines for configs 2, 4, 5 (4 lanes + nested 4 lanes for comparison); default bench
O=gpurun_out/r3_run6; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err || tail -5 $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); r=d['roofline']; print('default', d['value'], d['value_single_context'], d['ms_per_step'], r['bound'], r['frac'], r['levels'], r['job'], r.get('lane_utilisation'))"
bash tools/r3_pmc.sh r3_run6/pmc_config4_lanes1 --config 4 --lanes 1
python bench.py --config 4 --no-cpu-baseline > $O/bench_config4.json 2> $O/err.txt || tail -5 $O/err.txt
python bench.py --config 2 --no-cpu-baseline > $O/bench_config2.json 2> $O/err.txt || tail -5 $O/err.txt
python bench.py --config 5 --steps 64 --no-cpu-baseline > $O/bench_config5.json 2> $O/err.txt || tail -5 $O/err.txt
python bench.py --config 5 --steps 64 --no-cpu-baseline --extend-variant 4 --no-single > $O/bench_config5_nested_l4.json 2> $O/err.txt || tail -5 $O/err.txt
for f in bench_config4 bench_config2 bench_config5 bench_config5_nested_l4; do python -c "
import json; d=json.load(open('$O/$f.json')); print('$f', d['value'], d['value_single_context'], d['ms_per_step'], d['stage_ms_per_step'], d['roofline']['kernel'][:60])"; done
python -m pytest tests -m gpu -q -k "through_a_tlas" > $O/gputests_tlas.log 2>&1 || { grep -E "^FAILED|^ERROR|Error" $O/gputests_tlas.log | head -20; }
tail -2 $O/gputests_tlas.log
