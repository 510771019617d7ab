O=gpurun_out/r3_b9; mkdir -p $O
python -m pytest tests -m gpu -q -k "tlas or multi_blas or config5 or instance" > $O/t.log 2>&1 || { grep -E "^FAILED|^ERROR|Error|assert" $O/t.log | head -20; }
tail -2 $O/t.log
python tools/deep_fuzz.py 100000 1500 multi > $O/fuzz_multi_small.txt 2>&1; tail -1 $O/fuzz_multi_small.txt
python tools/deep_fuzz.py 110000 400 multi big > $O/fuzz_multi_big.txt 2>&1; tail -1 $O/fuzz_multi_big.txt
run() { env "$@" python bench.py --config 5 --steps 96 --no-cpu-baseline > $O/b.json 2>$O/err.txt || tail -3 $O/err.txt; python -c "
import json; d=json.load(open('$O/b.json')); print('%-22s lanes %d: %8.1f one context %8.1f stages %s' % ('$*', d['config']['lanes'], d['value'], d['value_single_context'], d['stage_ms_per_step']))"; }
run RT355_COHERENT=0
run RT355_COHERENT=1
run RT355_COHERENT=0
run RT355_COHERENT=1
cp $O/b.json $O/bench_config5.json
bash tools/trace.sh r3_b9/trace_c5_lanes1 --config 5 --lanes 1 --no-single --no-repeat > $O/per_bounce_c5.txt 2>&1; tail -7 $O/per_bounce_c5.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
