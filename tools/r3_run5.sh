# round-3 GPU run 5: REF_BUILTINS build vs the reference kernels on all 64 bands; clocks traced vs untraced; PMC config 5
O=gpurun_out/r3_run5; mkdir -p $O
python -m pytest tests -m gpu -q -k "ref_builtins or through_a_tlas" > $O/gputests_refb.log 2>&1 || { grep -E "^FAILED|^ERROR|Error" $O/gputests_refb.log | head -20; }
tail -2 $O/gputests_refb.log
python tools/find_flipfree_s1.py --refb > $O/refb_s1_bands.log 2>&1; grep -c "counts True seeds True" $O/refb_s1_bands.log; grep "S1FREE\|counts False\|seeds False" $O/refb_s1_bands.log | head -20
# clocks: does the GPU clock differ between a traced and an untraced run of the same workload?
( python bench.py --config 5 --lanes 1 --steps 400 --warmup 2 --no-cpu-baseline --no-profile --no-single --no-repeat > $O/clk_untraced.json 2>/dev/null & ) ; sleep 12; for i in 1 2 3; do rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | head -3; sleep 0.4; done > $O/clk_untraced.txt; wait; sleep 3
cat $O/clk_untraced.txt | head -9; python -c "
import json; d=json.load(open('$O/clk_untraced.json')); print('untraced', d['value'], d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
( rocprofv3 --kernel-trace --output-format csv -d /tmp/clk_tr -- python3 $GRAFT_REPO_ROOT/bench.py --config 5 --lanes 1 --steps 400 --warmup 2 --no-cpu-baseline --no-profile --no-single --no-repeat > $GRAFT_REPO_ROOT/$O/clk_traced.log 2>/dev/null & ) ; sleep 14; for i in 1 2 3; do rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | head -3; sleep 0.4; done > $GRAFT_REPO_ROOT/$O/clk_traced.txt; wait; sleep 5
cd $GRAFT_REPO_ROOT; cat $O/clk_traced.txt | head -9; grep '^{' $O/clk_traced.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('traced', d['value'], d['ms_per_step'])"
bash tools/r3_pmc.sh r3_run5/pmc_config5_lanes1 --config 5 --lanes 1
