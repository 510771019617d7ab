# A/B of the pair fetch of k_trace_persist (RT355_PAIRFETCH bit 0 extend, bit 1 connect) against the plain fetch on the bench frame: one context, 64 steps; accumulators must be identical
set -e
O=gpurun_out/r2_pair; mkdir -p $O
for m in 0 1 2 3; do
  RT355_PAIRFETCH=$m python bench.py --steps 64 --lanes 1 --no-cpu-baseline --no-single --dump-accum $O/acc$m.npy "$@" > $O/b$m.json 2> $O/b$m.err || { tail -5 $O/b$m.err; exit 1; }
  python -c "
import json; d=json.load(open('$O/b$m.json')); print('pair mode', $m, 'value', d['value'], 'ms/step', d['ms_per_step'], d['stage_ms_per_step'])"
done
python -c "
import numpy as np
a=np.load('$O/acc0.npy')
for m in (1,2,3):
    b=np.load('$O/acc%d.npy'%m); print('mode', m, 'accumulator identical:', bool(np.array_equal(a.view(np.uint32), b.view(np.uint32))))
"
