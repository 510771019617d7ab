#!/usr/bin/env bash
# one context with fixed chunks: chunk size x refill threshold (RT355_TUNE sets extend, connect and BVH4 alike; fixed chunks re-enabled after it)
cd $GRAFT_REPO_ROOT
for c in 64 96 112 128 160 224; do for r in 16 24 32; do
  echo -n "== chunk $c refill $r : "
  env RT355_TUNE=$c,$r,6,8 RT355_FIXED_CHUNKS=1,1 timeout -k 10 200 python bench.py --lanes 1 --no-cpu-baseline --no-profile --no-single 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
echo -n "== default : "; timeout -k 10 200 python bench.py --lanes 1 --no-cpu-baseline --no-profile --no-single 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
