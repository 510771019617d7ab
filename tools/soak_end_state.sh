#!/usr/bin/env bash
# soak of the end-state defaults (four lanes, one shared scene copy, eight hardware queues): one process for ~35 s, then two processes
# at once on the same GPU (8 contexts); exit codes and accumulator checksums (the two concurrent processes must agree)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_soak_end; mkdir -p $O
timeout -k 10 300 python bench.py --steps 16384 --no-cpu-baseline --no-single --no-repeat > $O/one.json 2> $O/one.err; echo "one process rc $?"
timeout -k 10 300 python bench.py --steps 4096 --no-cpu-baseline --no-single --no-repeat > $O/a.json 2> $O/a.err & pa=$!
timeout -k 10 300 python bench.py --steps 4096 --no-cpu-baseline --no-single --no-repeat > $O/b.json 2> $O/b.err & pb=$!
wait $pa; ra=$?; wait $pb; rb=$?
echo "two processes rc $ra $rb"
python - <<'PY'
import json
O = "gpurun_out/r2_soak_end"
one, a, b = (json.load(open(f"{O}/{n}.json")) for n in ("one", "a", "b"))
print("one process:", one["value"], "M samples/s over", one["timed_s"], "s, checksum", one["accum_rgb_sum"])
print("two at once:", a["value"], b["value"], "M samples/s each, checksums", a["accum_rgb_sum"], b["accum_rgb_sum"], "equal" if a["accum_rgb_sum"] == b["accum_rgb_sum"] else "DIFFERENT")
PY
