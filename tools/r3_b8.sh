O=gpurun_out/r3_b8; mkdir -p $O
python -m pytest tests -m gpu -q -k "tlas or multi_blas or config5 or instance" > $O/t.log 2>&1 || { grep -E "^FAILED|^ERROR|Error|assert" $O/t.log | head -20; }
tail -2 $O/t.log
python tools/deep_fuzz.py 70000 800 multi > $O/fuzz_multi_small.txt 2>&1; tail -1 $O/fuzz_multi_small.txt
python bench.py --config 5 --steps 96 --no-cpu-baseline > $O/bench_config5.json 2> $O/err.txt || tail -5 $O/err.txt
python -c "
import json; d=json.load(open('$O/bench_config5.json')); print('config 5: value %8.1f single %s lanes %s stages %s' % (d['value'], d['value_single_context'], d['config']['lanes'], d['stage_ms_per_step']))"
bash tools/trace.sh r3_b8/trace_c5_lanes1 --config 5 --lanes 1 --no-single --no-repeat > $O/per_bounce_c5.txt 2>&1; tail -7 $O/per_bounce_c5.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
