O=gpurun_out/r3_b15; mkdir -p $O
RT355_THIN=1 python -m pytest tests -m gpu -q -x -k "tlas or persist or config2 or config5 or frame or stage" > $O/t1.log 2>&1 || { grep -E "^FAILED|^ERROR" $O/t1.log | head; }; tail -1 $O/t1.log
for v in "0 16384" "1 16384" "1 4096" "1 1024" "1 0"; do set -- $v
  for c in 2 5 3; do
    RT355_THIN=$1 RT355_XCD_RAYS=$2 python bench.py --config $c --steps 96 --no-cpu-baseline > $O/b.json 2>$O/err.txt || tail -3 $O/err.txt
    python -c "
import json; d=json.load(open('$O/b.json')); print('thin $1 xcd_rays $2 config $c lanes %d: %8.1f one context %8.1f extend %s' % (d['config']['lanes'], d['value'], d['value_single_context'] or 0, d['stage_ms_per_step']['extend']))"
  done
done
RT355_THIN=1 bash tools/trace.sh r3_b15/trace_c2 --config 2 --lanes 1 --no-single --no-repeat > $O/pb_c2.txt 2>&1; tail -7 $O/pb_c2.txt | head -2
RT355_THIN=1 bash tools/trace.sh r3_b15/trace_c5 --config 5 --lanes 1 --no-single --no-repeat > $O/pb_c5.txt 2>&1; tail -7 $O/pb_c5.txt | head -2
RT355_THIN=1 bash tools/trace.sh r3_b15/trace_c3 --lanes 1 --no-single --no-repeat > $O/pb_c3.txt 2>&1; tail -7 $O/pb_c3.txt | head -2
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
