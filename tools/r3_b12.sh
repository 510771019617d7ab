O=gpurun_out/r3_b12; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/gputests.log 2>&1 || { grep -E "^FAILED|^ERROR" $O/gputests.log | head -30; }
tail -2 $O/gputests.log
run() { cfg=$1; shift; python bench.py --config $cfg --no-cpu-baseline "$@" > $O/b$cfg.json 2>$O/err.txt || tail -3 $O/err.txt; python -c "
import json; d=json.load(open('$O/b$cfg.json')); print('config $cfg lanes %d: %8.1f one context %8.1f stages %s' % (d['config']['lanes'], d['value'], d['value_single_context'] or 0, d['stage_ms_per_step']))"; }
run 3
run 2
run 5 --steps 96
run 4
bash tools/trace.sh r3_b12/trace_c3_lanes1 --lanes 1 --no-single --no-repeat > $O/per_bounce_c3.txt 2>&1; tail -7 $O/per_bounce_c3.txt
bash tools/trace.sh r3_b12/trace_c5_lanes1 --config 5 --lanes 1 --no-single --no-repeat > $O/per_bounce_c5.txt 2>&1; tail -7 $O/per_bounce_c5.txt
bash tools/trace.sh r3_b12/trace_c2_lanes1 --config 2 --lanes 1 --no-single --no-repeat > $O/per_bounce_c2.txt 2>&1; tail -7 $O/per_bounce_c2.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
