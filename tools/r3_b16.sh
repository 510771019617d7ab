O=gpurun_out/r3_b16; mkdir -p $O
for v in "8 1024" "16 1024" "24 1024" "16 4096" "16 0"; do set -- $v
  for c in 2 5 3 4; do
    RT355_THIN=$1 RT355_XCD_RAYS=$2 python bench.py --config $c --steps 96 --no-cpu-baseline > $O/b.json 2>$O/err.txt || tail -3 $O/err.txt
    python -c "
import json; d=json.load(open('$O/b.json')); print('thin $1 xcd_rays $2 config $c lanes %d: %8.1f one context %8.1f extend %s connect %s' % (d['config']['lanes'], d['value'], d['value_single_context'] or 0, d['stage_ms_per_step']['extend'], d['stage_ms_per_step']['connect']))"
  done
done
