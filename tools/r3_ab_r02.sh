# same box, back to back: the round-2 end state (gpurun_ab/r02, built here) against the working tree: is the single-context figure a code or a box effect?
O=$GRAFT_REPO_ROOT/gpurun_out/r3_ab; mkdir -p $O
cd $GRAFT_REPO_ROOT/gpurun_ab/r02 && python -m magr_ray_tracer_amd.build > $O/build_r02.log 2>&1; tail -2 $O/build_r02.log
for rep in 1 2; do
cd $GRAFT_REPO_ROOT/gpurun_ab/r02 && python bench.py --steps 64 --warmup 4 --no-cpu-baseline > $O/r02_$rep.json 2>$O/err.txt; python -c "
import json; d=json.load(open('$O/r02_$rep.json')); print('r02  ', d['value'], d['value_single_context'], d['stage_ms_per_step'])"
cd $GRAFT_REPO_ROOT && python bench.py --steps 64 --warmup 4 --no-cpu-baseline > $O/new_$rep.json 2>$O/err.txt; python -c "
import json; d=json.load(open('$O/new_$rep.json')); print('new  ', d['value'], d['value_single_context'], d['stage_ms_per_step'])"
done
cd $GRAFT_REPO_ROOT/gpurun_ab/r02 && python bench.py --steps 64 --warmup 4 --no-cpu-baseline --lanes 1 > $O/r02_l1.json 2>$O/err.txt; python -c "
import json; d=json.load(open('$O/r02_l1.json')); print('r02 lanes1 ', d['value'], d['value_single_context'])"
cd $GRAFT_REPO_ROOT && python bench.py --steps 64 --warmup 4 --no-cpu-baseline --lanes 1 > $O/new_l1.json 2>$O/err.txt; python -c "
import json; d=json.load(open('$O/new_l1.json')); print('new lanes1 ', d['value'], d['value_single_context'])"
