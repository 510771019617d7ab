#!/usr/bin/env bash
# Vector-L1 (TCP) counters for the bench workload, each set in its own rocprofv3 run.
# usage: tools/pmc_ta.sh <outdir-under-gpurun_out> [bench args...]
set -u
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
# (the TA_* and TD_* counter sets never returned on this pool - three passes ran into the 300 s limit - so only the TCP sets are collected)
for set in "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/pass$i" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 4 --warmup 1 --no-cpu-baseline --no-profile "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed: $(tail -2 $OUT/pass$i.log)"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "k_trace" not in k and "k_shade" not in k: continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(k, row["Counter_Name"])] += 1
with open(out + "/pmc_ta_summary.csv", "w") as fo:
    fo.write("kernel,counter,dispatches,sum,per_dispatch\n")
    for k in sorted(agg):
        for c in sorted(agg[k]):
            n = cnt[(k, c)]
            fo.write(f"\"{k}\",{c},{n},{agg[k][c]:.0f},{agg[k][c] / n:.1f}\n")
print(open(out + "/pmc_ta_summary.csv").read()[:8000])
PY
