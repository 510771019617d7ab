# round-2 additions on the GPU box: new parity tests, then a short bench line
set -e
python -m pytest tests/test_gpu_parity.py -x -q -k "bench_starts or bench_scene_1080p or max_bounces or tlas_cycles" > gpurun_out/r2_tests.log 2>&1 || { tail -40 gpurun_out/r2_tests.log; exit 1; }
tail -3 gpurun_out/r2_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r2_bench_20.json 2> gpurun_out/r2_bench_20.err || { tail -20 gpurun_out/r2_bench_20.err; exit 1; }
cat gpurun_out/r2_bench_20.json
