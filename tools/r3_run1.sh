# round-3 GPU run 1: GPU tests on the refactored library (group API), PMC calibration, first PMC passes and bench lines.
set -e
O=gpurun_out/r3_run1; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1 || { tail -30 $O/gputests.log; exit 1; }
tail -3 $O/gputests.log
bash tools/r3_pmc.sh r3_run1/calib calib
./tools/pmc_calib.bin > $O/calib_bytes.txt; cat $O/calib_bytes.txt
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err || { tail -20 $O/bench_driver_flags.err; exit 1; }
cut -c1-1500 $O/bench_driver_flags.json
bash tools/r3_pmc.sh r3_run1/pmc_config3_lanes1 --lanes 1
bash tools/r3_pmc.sh r3_run1/pmc_config3_lanes4
./examples/headless_tick --lanes 4 --spp 64 --out $O/tick4.png 2>&1 | tail -3
GPU_MAX_HW_QUEUES=2 ./examples/headless_tick --lanes 4 --spp 64 --out $O/tick4q2.png 2>&1 | tail -3
