#!/usr/bin/env bash
# sweep of the extend launches' persistent-wavefront parameters (connect stays at its own optimum)
cd $GRAFT_REPO_ROOT
for t in "$@"; do
  echo "== RT355_TUNE=$t"; RT355_TUNE=$t RT355_TUNE_CONNECT=128,32,6,8 timeout -k 10 120 python tools/ab_bench.py 8 0 2>&1 | tail -1 | cut -c1-170
done
