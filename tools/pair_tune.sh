set -e
O=gpurun_out/r2_pair; mkdir -p $O
for t in 56,8,6,4 112,16,8,8 32,12,4,2 64,24,6,16 64,4,6,8; do
 for m in 1 2; do
  RT355_TUNE_PAIR=$t RT355_PAIR=$m python bench.py --steps 64 --lanes 1 --no-cpu-baseline > $O/t.json 2> $O/t.err || { tail -5 $O/t.err; exit 1; }
  python -c "
import json; d=json.load(open('$O/t.json')); print('tune $t mode', $m, 'value', d['value'], 'ms/step', d['ms_per_step'], d.get('stage_ms_per_step'))"
 done
done
