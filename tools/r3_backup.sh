O=gpurun_out/r3_backup; mkdir -p $O
python -m pytest tests -m gpu -q -k "tlas or multi_blas or config5 or instance" > $O/t.log 2>&1 || { grep -E "^FAILED|^ERROR|Error|assert" $O/t.log | head -20; }
tail -2 $O/t.log
run() { env "$@" python bench.py --config 5 --steps 64 --no-cpu-baseline > $O/b.json 2>$O/err.txt || tail -3 $O/err.txt; python -c "
import json; d=json.load(open('$O/b.json')); print('%-50s 4 lanes %8.1f one context %8.1f stages %s' % ('$*', d['value'], d['value_single_context'], d['stage_ms_per_step']))"; }
run RT355_TLAS_BACKUP=0
run RT355_TLAS_BACKUP=1
run RT355_TLAS_BACKUP=1 RT355_SPILL_CAP=20
run RT355_TLAS_BACKUP=0 RT355_SPILL_CAP=12
run RT355_TLAS_BACKUP=1 RT355_SPILL_CAP=8
