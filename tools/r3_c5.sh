# config 5 on the end state (world ray in LDS): TLAS / multi-BLAS tests, multi-BLAS fuzz, bench lines, kernel stats, per bounce, PMC passes
O=gpurun_out/r3_c5; mkdir -p $O
python -m pytest tests -m gpu -q -k "tlas or multi_blas or config5 or instance or group or lanes" > $O/t.log 2>&1 || { grep -E "^FAILED|^ERROR|Error|assert" $O/t.log | head -20; }
tail -2 $O/t.log
python tools/deep_fuzz.py 50000 1500 multi > $O/fuzz_multi_small.txt 2>&1; tail -2 $O/fuzz_multi_small.txt
python tools/deep_fuzz.py 60000 600 multi big > $O/fuzz_multi_big.txt 2>&1; tail -2 $O/fuzz_multi_big.txt
python bench.py --config 5 --steps 64 --no-cpu-baseline > $O/bench_config5.json 2> $O/err.txt || tail -5 $O/err.txt
python bench.py --config 5 --steps 64 --no-cpu-baseline --extend-variant 4 > $O/bench_config5_nested.json 2> $O/err.txt || tail -5 $O/err.txt
for f in bench_config5 bench_config5_nested; do python -c "
import json; d=json.load(open('$O/$f.json')); r=d['roofline']; print('%-22s value %8.1f single %s ms/step %.4f bound %s frac %s job %s util %s' % ('$f', d['value'], d['value_single_context'], d['ms_per_step'], r.get('bound'), r.get('frac'), (r.get('job') or {}).get('frac'), r.get('lane_utilisation')))"; done
bash tools/trace_default.sh r3_c5/trace_c5 --config 5 --no-single --steps 64 > $O/trace_c5.txt 2>&1; tail -8 $O/trace_c5.txt | cut -c1-150
bash tools/trace.sh r3_c5/trace_c5_lanes1 --config 5 --lanes 1 --no-single --no-repeat > $O/per_bounce_c5.txt 2>&1; tail -7 $O/per_bounce_c5.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
bash tools/r3_pmc.sh r3_c5/pmc_config5_lanes1 --config 5 --lanes 1
bash tools/r3_pmc.sh r3_c5/pmc_config5_lanes4 --config 5
