#!/usr/bin/env bash
# one-context bench value under different RT355_TUNE settings: usage tune_single.sh "VAR=val" ...
cd $GRAFT_REPO_ROOT
for t in "$@"; do
  echo -n "== $t : "
  env $t timeout -k 10 200 python bench.py --lanes 1 --no-cpu-baseline --no-profile --no-single 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
