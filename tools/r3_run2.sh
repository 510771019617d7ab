# round-3 GPU run 2: the new kernels' tests (TLAS persistent + spill stack, REF_BUILTINS build), config-5 A/B, PMC passes for config 3.
O=gpurun_out/r3_run2; mkdir -p $O
python -m pytest tests -m gpu -q -k "tlas or config5 or ref_builtins or instance or two_blas or bench_default or headless" > $O/gputests_new.log 2>&1 || { grep -E "^FAILED|^ERROR|Error|assert" $O/gputests_new.log | head -30; }
tail -3 $O/gputests_new.log
for v in 0 4; do
  python bench.py --config 5 --lanes 1 --steps 32 --warmup 2 --no-cpu-baseline --extend-variant $v > $O/bench_config5_v$v.json 2> $O/bench_config5_v$v.err || tail -5 $O/bench_config5_v$v.err
  python -c "
import json; d=json.load(open('$O/bench_config5_v$v.json')); print('config5 variant $v', d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['roofline']['kernel'])"
done
RT355_NO_SPILL=1 python bench.py --config 5 --lanes 1 --steps 32 --warmup 2 --no-cpu-baseline > $O/bench_config5_nospill.json 2> $O/bench_config5_nospill.err || tail -5 $O/bench_config5_nospill.err
python -c "
import json; d=json.load(open('$O/bench_config5_nospill.json')); print('config5 no spill', d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['roofline']['kernel'])"
python bench.py --config 5 --steps 32 --warmup 2 --no-cpu-baseline > $O/bench_config5_lanes4.json 2> $O/bench_config5_lanes4.err || tail -5 $O/bench_config5_lanes4.err
python -c "
import json; d=json.load(open('$O/bench_config5_lanes4.json')); print('config5 lanes4', d['value'], d['ms_per_step'], d['value_single_context'])"
bash tools/r3_pmc.sh r3_run2/pmc_config3_lanes1 --lanes 1
bash tools/r3_pmc.sh r3_run2/pmc_config3_lanes4
