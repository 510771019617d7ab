set -u
OUT="$GRAFT_REPO_ROOT/gpurun_out/r2_pmc_pf"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for m in 0 3; do
  export RT355_PAIRFETCH=$m RT355_TUNE=128,32,6,16,5
  timeout -k 10 240 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/m$m" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 4 --warmup 1 --no-cpu-baseline --no-profile --lanes 1 > "$OUT/m$m.log" 2>&1 || echo "mode $m failed: $(tail -2 $OUT/m$m.log)"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for m in (0, 3):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
    for f in glob.glob(out + "/m%d/**/*counter_collection.csv" % m, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            if "k_trace_persist" not in k: continue
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
    for k in sorted(agg):
        print("pairfetch", m, k, {c: round(agg[k][c] / cnt[(k, c)]) for c in sorted(agg[k])})
PY
