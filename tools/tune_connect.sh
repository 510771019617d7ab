#!/usr/bin/env bash
cd $GRAFT_REPO_ROOT
for t in "$@"; do
  echo "== RT355_TUNE_CONNECT=$t"; RT355_TUNE_CONNECT=$t timeout -k 10 120 python tools/ab_bench.py 6 0 2>&1 | tail -1
done
