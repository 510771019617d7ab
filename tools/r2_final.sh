set -e
O=gpurun_out/r2_final2; mkdir -p $O
python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
bash tools/trace_default.sh r2_final2/trace_default --no-single
python bench.py --shard ibands --steps 16 --no-cpu-baseline > $O/bench_ibands.json 2> $O/bench_ibands.err
python bench.py --shard bands --steps 16 --no-cpu-baseline > $O/bench_bands.json 2> $O/bench_bands.err
python bench.py > $O/bench.json 2> $O/bench.err
python -c "
import json
for f in ('bench','bench_ibands','bench_bands'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d.get('value_single_context'), d['ms_per_step'], d['repeats'], d['scaling'])
"
