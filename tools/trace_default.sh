#!/usr/bin/env bash
# rocprofv3 --kernel-trace --stats of the DEFAULT bench command (python3 bench.py), JSON line kept beside the kernel stats
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$GRAFT_REPO_ROOT/bench.py" "$@" > "$OUT/bench.log" 2>&1
grep '^{' "$OUT/bench.log" | tail -1 > "$OUT/bench.json"
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
head -8 "$f" | cut -d, -f1-4 | cut -c1-150
python3 -c "import json; d=json.loads(open('$OUT/bench.json').read()); print('bench value', d['value'], 'ms_per_step', d['ms_per_step'])"
