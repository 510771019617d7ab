#!/usr/bin/env bash
cd $GRAFT_REPO_ROOT
for t in "64,20,6" "64,32,6" "64,16,4" "64,8,4" "64,24,12" "128,20,6" "64,20,2" "64,40,6" "64,20,6,4" "64,20,6,5"; do
  echo "== RT355_TUNE=$t"; RT355_TUNE=$t timeout -k 10 120 python tools/ab_bench.py 6 0 2>&1 | tail -1
done
