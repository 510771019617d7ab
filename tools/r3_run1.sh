# round-3 GPU run 1: GPU tests on the refactored library (group API, k_trace_persist_tlas), PMC calibration, bench lines, config-5 A/B.
O=gpurun_out/r3_run1; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputests.log 2>&1 || { grep -E "^FAILED|^ERROR|passed|failed" $O/gputests.log | tail -20; }
tail -3 $O/gputests.log
bash tools/r3_pmc.sh r3_run1/calib calib
./tools/pmc_calib.bin > $O/calib_bytes.txt; cat $O/calib_bytes.txt
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err || tail -20 $O/bench_driver_flags.err
cut -c1-1800 $O/bench_driver_flags.json
for v in 0 4; do
  python bench.py --config 5 --lanes 1 --steps 32 --warmup 2 --no-cpu-baseline --extend-variant $v > $O/bench_config5_v$v.json 2> $O/bench_config5_v$v.err || tail -5 $O/bench_config5_v$v.err
  python -c "
import json; d=json.load(open('$O/bench_config5_v$v.json')); print('config5 variant $v', d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['roofline']['kernel'])"
done
./examples/headless_tick --lanes 4 --spp 64 --out $O/tick4.png 2>&1 | tail -3
GPU_MAX_HW_QUEUES=2 ./examples/headless_tick --lanes 4 --spp 64 --out $O/tick4q2.png 2>&1 | tail -3
