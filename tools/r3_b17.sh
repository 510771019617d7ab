O=gpurun_out/r3_b17; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputests.log 2>&1 || { grep -E "^FAILED|^ERROR" $O/gputests.log | head -30; }
tail -2 $O/gputests.log
python tools/deep_fuzz.py 150000 1500 multi > $O/fuzz_multi.txt 2>&1; tail -1 $O/fuzz_multi.txt
python tools/deep_fuzz.py 160000 1500 > $O/fuzz_soups.txt 2>&1; tail -1 $O/fuzz_soups.txt
python tools/deep_fuzz.py 170000 600 mixed > $O/fuzz_mixed.txt 2>&1; tail -1 $O/fuzz_mixed.txt
for c in 2 5 3 4; do
  python bench.py --config $c --no-cpu-baseline $( [ $c = 5 ] && echo --steps 96 ) > $O/bench_config$c.json 2>$O/err.txt || tail -3 $O/err.txt
  python -c "
import json; d=json.load(open('$O/bench_config$c.json')); print('config $c lanes %d: %8.1f one context %8.1f stages %s' % (d['config']['lanes'], d['value'], d['value_single_context'] or 0, d['stage_ms_per_step']))"
done
bash tools/trace.sh r3_b17/trace_c2 --config 2 --lanes 1 --no-single --no-repeat > $O/pb_c2.txt 2>&1; tail -7 $O/pb_c2.txt
bash tools/trace.sh r3_b17/trace_c5 --config 5 --lanes 1 --no-single --no-repeat > $O/pb_c5.txt 2>&1; tail -7 $O/pb_c5.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
