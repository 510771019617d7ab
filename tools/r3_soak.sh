# end state of round 3: config 5 (k_trace_persist_tlas, six lanes, 4K) for ~30 s in one process, the default command for ~35 s, and the
# multi-BLAS fuzz on the kernels as they are now (world ray in LDS, tiled primary rays, spill policy)
O=gpurun_out/r3_soak; mkdir -p $O
timeout -k 10 400 python bench.py --config 5 --steps 16384 --no-cpu-baseline --no-single --no-repeat > $O/c5.json 2> $O/c5.err; echo "config 5 rc $?"
timeout -k 10 400 python bench.py --steps 16384 --no-cpu-baseline --no-single --no-repeat > $O/c3.json 2> $O/c3.err; echo "config 3 rc $?"
python - <<'PY'
import json
for n in ("c5", "c3"):
    d = json.load(open(f"gpurun_out/r3_soak/{n}.json"))
    print(n, d["value"], "M samples/s over", d["timed_s"], "s,", d["steps"], "frames, checksum", d["accum_rgb_sum"])
PY
python tools/deep_fuzz.py 80000 4000 multi > $O/fuzz_multi_small.txt 2>&1; tail -1 $O/fuzz_multi_small.txt
python tools/deep_fuzz.py 90000 1000 multi big > $O/fuzz_multi_big.txt 2>&1; tail -1 $O/fuzz_multi_big.txt
