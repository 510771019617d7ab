#!/usr/bin/env bash
# kernel trace of a short bench run + per-bounce table; usage: tools/trace.sh <name> [bench args]
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 24 --warmup 2 --no-cpu-baseline --no-profile "$@" > "$OUT/bench.log" 2>&1
f=$(find "$OUT" -name "*kernel_trace.csv" | head -1)
python3 "$GRAFT_REPO_ROOT/tools/per_bounce.py" "$f" | tee "$OUT/per_bounce.txt"
grep '^{' "$OUT/bench.log" | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['rays_per_step'])"
