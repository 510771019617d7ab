#!/usr/bin/env bash
# PMC passes for one bench workload: each counter set in its own rocprofv3 run (--kernel-trace only; MI355X_MICROARCH.md, rocprofv3 PMC slots),
# summed per kernel into <out>/pmc_summary.csv.  With "calib" as the workload the passes run tools/pmc_calib.bin (known byte counts) instead.
# usage: tools/r3_pmc.sh <outdir-under-gpurun_out> calib | <bench args...>
set -u
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE TCP_TCC_WRITE_REQ_sum" \
           "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_EA0_RDREQ_DRAM_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  if [ "${1:-}" = "calib" ]; then
    timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/pass$i" -- "$GRAFT_REPO_ROOT/tools/pmc_calib.bin" > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed: $(tail -2 $OUT/pass$i.log)"
  else
    timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/pass$i" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 8 --warmup 1 --no-repeat --no-single --no-cpu-baseline --no-profile "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed: $(tail -2 $OUT/pass$i.log)"
  fi
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
print(out, "kernels:", len(agg), "rows:", sum(len(v) for v in agg.values()))
PY
rm -rf "$OUT"/pass*/   # the per-dispatch CSVs are large; the summary and the logs stay
