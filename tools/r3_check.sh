O=gpurun_out/r3_check; mkdir -p $O
python -m pytest tests -m gpu -q -k "through_a_tlas or fixed_image or config5 or group_of_lanes or renderer_mirror_with_lanes or bench_" > $O/t.log 2>&1 || { grep -E "^FAILED|^ERROR|Error" $O/t.log | head -20; }
tail -2 $O/t.log
python bench.py --config 5 --steps 64 --no-cpu-baseline > $O/bench_config5.json 2> $O/err.txt || tail -5 $O/err.txt
python bench.py --steps 64 --no-cpu-baseline > $O/bench.json 2> $O/err.txt || tail -5 $O/err.txt
for f in bench_config5 bench; do python -c "
import json; d=json.load(open('$O/$f.json')); r=d['roofline']; print('%-22s value %8.1f single %s ms/step %.4f bound %s frac %s util %s stages %s' % ('$f', d['value'], d['value_single_context'], d['ms_per_step'], r.get('bound'), r.get('frac'), r.get('lane_utilisation'), d['stage_ms_per_step']))"; done
