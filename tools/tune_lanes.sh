#!/usr/bin/env bash
# bench value (default lanes) under different environment tunings: usage tune_lanes.sh "VAR=val VAR2=val" ...
cd $GRAFT_REPO_ROOT
for t in "$@"; do
  echo -n "== $t : "
  env $t timeout -k 10 200 python bench.py --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
