#!/usr/bin/env bash
# three lanes with one shared device copy of the scene (default) against three copies (--no-share-scene)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for f in "" "--no-share-scene"; do
  echo -n "== lanes 3 ${f:-shared} : "
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-profile --no-single $f 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
for f in "" "--no-share-scene"; do
  echo -n "== lanes 6 ${f:-shared} : "
  timeout -k 10 200 python bench.py --lanes 6 --no-cpu-baseline --no-profile --no-single $f 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
