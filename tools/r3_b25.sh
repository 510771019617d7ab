O=gpurun_out/r3_b25; mkdir -p $O
python -m pytest tests -m gpu -q -x -k "frame or stage or persist or tlas or config" > $O/t.log 2>&1 || { grep -E "^FAILED|^ERROR" $O/t.log | head; }; tail -1 $O/t.log
for c in 2 3 5; do
  python bench.py --config $c --no-cpu-baseline $( [ $c = 5 ] && echo --steps 96 ) > $O/b.json 2>$O/err.txt || tail -3 $O/err.txt
  python -c "
import json; d=json.load(open('$O/b.json')); print('leaf-one config $c lanes %d: %8.1f one context %8.1f stages %s' % (d['config']['lanes'], d['value'], d['value_single_context'] or 0, d['stage_ms_per_step']))"
done
bash tools/trace.sh r3_b25/trace_c3 --lanes 1 --no-single --no-repeat > $O/pb_c3.txt 2>&1; tail -7 $O/pb_c3.txt | head -2
bash tools/trace.sh r3_b25/trace_c5 --config 5 --lanes 1 --no-single --no-repeat > $O/pb_c5.txt 2>&1; tail -7 $O/pb_c5.txt | head -2
bash tools/trace.sh r3_b25/trace_c2 --config 2 --lanes 1 --no-single --no-repeat > $O/pb_c2.txt 2>&1; tail -7 $O/pb_c2.txt | head -2
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
