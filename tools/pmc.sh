#!/usr/bin/env bash
# PMC passes for the bench workload (each counter set in its own rocprofv3 run, kernel-trace only).
# usage: tools/pmc.sh <outdir-under-gpurun_out> [bench args...]
set -u
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"; do
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
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(k, row["Counter_Name"])] += 1
with open(out + "/pmc_summary.csv", "w") as fo:
    fo.write("kernel,counter,dispatches,sum,per_dispatch\n")
    for k in sorted(agg):
        for c in sorted(agg[k]):
            n = cnt[(k, c)]
            fo.write(f"\"{k}\",{c},{n},{agg[k][c]:.0f},{agg[k][c] / n:.1f}\n")
print(open(out + "/pmc_summary.csv").read()[:6000])
PY
