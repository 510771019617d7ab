# GPU clock under load: un-profiled (both HIP runtimes) vs under rocprofv3 - config 5, one context, long run; rocm-smi sampled all along
O=$GRAFT_REPO_ROOT/gpurun_out/r3_clocks; mkdir -p $O
sample() { for i in $(seq 1 60); do rocm-smi --showclocks 2>/dev/null | grep -E "sclk" | head -1 | sed 's/.*(\(.*\)Mhz)/\1/'; sleep 0.35; done | sort -n | uniq -c | sort -k2 -n | tr '\n' ';'; echo; }
for order in lib-first torch-first; do
  RT355_IMPORT_ORDER=$order python bench.py --config 5 --lanes 1 --steps 1500 --warmup 2 --no-cpu-baseline --no-profile --no-single --no-repeat > $O/u_$order.json 2>/dev/null &
  sleep 9; echo "sclk MHz histogram, un-profiled $order:"; sample; wait
  python -c "
import json; d=json.load(open('$O/u_$order.json')); print('un-profiled $order', d['value'], d['ms_per_step'])"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/clk_tr -- python3 $GRAFT_REPO_ROOT/bench.py --config 5 --lanes 1 --steps 1500 --warmup 2 --no-cpu-baseline --no-profile --no-single --no-repeat > $O/traced.log 2>/dev/null &
sleep 11; echo "sclk MHz histogram, under rocprofv3:"; sample; wait
grep '^{' $O/traced.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('traced', d['value'], d['ms_per_step'])"
