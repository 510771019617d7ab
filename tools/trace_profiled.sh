#!/usr/bin/env bash
# rocprofv3 kernel stats of the DEFAULT bench mode (HIP-event timing of the extend launches on) next to bench.py's own figure
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 64 --warmup 2 --no-cpu-baseline "$@" > "$OUT/bench.log" 2>&1
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
grep -E "k_trace_persist|k_shade" "$f" | cut -d, -f1-4 | cut -c1-160
grep '^{' "$OUT/bench.log" | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench value', d['value'], 'ms/step', d['ms_per_step'], 'extend avg_launch_ms', d['roofline']['avg_launch_ms'], 'connect', d['connect_roofline']['avg_launch_ms'], d['stage_ms_per_step'])"
