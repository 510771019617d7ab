#!/usr/bin/env bash
# ONE accumulation rendered as K interleaved row-band contexts on one GPU (bench.py --shard ibands --band-rows R) against
# one context (--lanes 1) and three sample streams (--lanes 3): value, and the accumulator compared bit for bit with --lanes 1.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_bands; mkdir -p $O
run() { name=$1; shift; timeout -k 10 300 python bench.py --steps 48 --no-cpu-baseline --no-single --no-repeat --dump-accum $O/$name.npy "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; exit 1; }; }
run lanes1 --lanes 1
run lanes3 --lanes 3
for r in 540 360 270 216 136 72; do run bands$r --shard ibands --band-rows $r; done
python - <<'PY'
import json, numpy as np, glob, os
O = "gpurun_out/r2_bands"
ref = np.load(f"{O}/lanes1.npy")
for f in ["lanes1", "lanes3"] + [f"bands{r}" for r in (540, 360, 270, 216, 136, 72)]:
    d = json.loads(open(f"{O}/{f}.json").read().strip().splitlines()[-1])
    a = np.load(f"{O}/{f}.npy")
    print(f, "contexts", d["config"]["contexts"], "value", d["value"], "ms/step", d["ms_per_step"],
          "bit-equal to one context:", bool(np.array_equal(a.view(np.uint32), ref.view(np.uint32))), flush=True)
for f in glob.glob(f"{O}/*.npy"):
    os.remove(f)
PY
