O=gpurun_out/r3_b7; mkdir -p $O
python -m pytest tests/test_gpu_reference.py -m gpu -q -k "extend_every_bounce" > $O/t.log 2>&1 || { grep -E "^FAILED|^ERROR|Error|assert" $O/t.log | head -30; }
tail -2 $O/t.log
python bench.py --config 5 --steps 96 --no-cpu-baseline > $O/bench_config5.json 2> $O/err.txt || tail -5 $O/err.txt
python bench.py --config 2 --no-cpu-baseline > $O/bench_config2.json 2> $O/err.txt || tail -5 $O/err.txt
for f in bench_config5 bench_config2; do python -c "
import json; d=json.load(open('$O/$f.json')); r=d['roofline']; print('%-22s value %8.1f single %s lanes %s ms/step %.4f bound %s frac %s valu %s job %s' % ('$f', d['value'], d['value_single_context'], d['config']['lanes'], d['ms_per_step'], r.get('bound'), r.get('frac'), (r.get('valu_issue') or {}).get('frac'), (r.get('job') or {}).get('valu_issue')))"; done
bash tools/r3_pmc.sh r3_b7/pmc_config5_lanes6 --config 5
