set -e
python -m pytest tests/test_gpu_parity.py -x -q -k "frames_bit_exact or stage_by_stage or lanes or odd_sizes" > gpurun_out/r2_acc.log 2>&1 || { tail -20 gpurun_out/r2_acc.log; exit 1; }
tail -1 gpurun_out/r2_acc.log
python -c "import __graft_entry__ as g; g.smoke()"
for cfg in "1 0" "1 1" "3 0" "3 1"; do
  set -- $cfg
  if [ "$2" = "1" ]; then export RT355_SHADE_PER_CU=1; else unset RT355_SHADE_PER_CU; fi
  python bench.py --steps 64 --lanes $1 --no-cpu-baseline > gpurun_out/r2_b_$1_$2.json 2> gpurun_out/r2_b_$1_$2.err
  python -c "
import json; d=json.load(open('gpurun_out/r2_b_$1_$2.json')); print('lanes $1 shade_per_cu_1=$2', 'value', d['value'], 'single', d['value_single_context'], d['stage_ms_per_step'])"
done
