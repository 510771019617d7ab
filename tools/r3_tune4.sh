# refill threshold / inner steps of the extend event loop with FOUR lanes sharing the GPU (round 2 tuned them for a context alone)
O=gpurun_out/r3_tune4; mkdir -p $O
run() { env "$@" python bench.py --steps 96 --warmup 4 --no-cpu-baseline --no-single --no-profile > $O/b.json 2>$O/err.txt; python -c "
import json; d=json.load(open('$O/b.json')); print('%-60s 4 lanes %8.1f' % ('$*', d['value']))"; }
run RT355_X=0
for r in 8 16 32; do run RT355_TUNE=112,$r,6,16 RT355_TUNE_CONNECT=128,32,6,16; done
for i in 4 10; do run RT355_TUNE=112,24,$i,16 RT355_TUNE_CONNECT=128,32,6,16; done
for c in 64 192; do run RT355_TUNE=$c,24,6,16 RT355_TUNE_CONNECT=128,32,6,16; done
run RT355_TUNE=112,24,6,8 RT355_TUNE_CONNECT=128,32,6,16
run RT355_TUNE=112,24,6,16 RT355_TUNE_CONNECT=128,16,6,16
run RT355_TUNE=112,24,6,16 RT355_TUNE_CONNECT=128,32,6,8
run RT355_X=0
