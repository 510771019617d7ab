# round-3: config-5 kernel stats; "while-while" = triangle path held until N lanes wait (N up to 64), with the lane utilisation split; tail probe on the end state
O=gpurun_out/r3_more; mkdir -p $O
bash tools/trace_default.sh r3_more/trace_c5 --config 5 --no-single --steps 64 > $O/trace_c5.txt 2>&1; tail -9 $O/trace_c5.txt | cut -c1-160
for L in 8 16 32 48 64; do
  RT355_TUNE=112,24,6,$L RT355_TUNE_CONNECT=128,32,6,16 python bench.py --lanes 1 --steps 64 --no-cpu-baseline > $O/b.json 2>$O/err.txt
  python -c "
import json; d=json.load(open('$O/b.json')); r=d['roofline']; print('leafK $L one context %7.1f extend ms/frame %.3f util %s' % (d['value'], d['stage_ms_per_step']['extend'], r['lane_utilisation']))"
  RT355_TUNE=112,24,6,$L RT355_TUNE_CONNECT=128,32,6,16 python bench.py --steps 64 --no-cpu-baseline --no-single > $O/b.json 2>$O/err.txt
  python -c "
import json; d=json.load(open('$O/b.json')); print('leafK $L four lanes  %7.1f' % d['value'])"
done
cp magr_ray_tracer_amd/librt355.so /tmp/librt355_keep.so
bash tools/lab/tail_probe.sh > $O/tail_probe.txt 2>&1; cp /tmp/librt355_keep.so magr_ray_tracer_amd/librt355.so
head -12 $O/tail_probe.txt | cut -c1-250
find $O -name "*kernel_trace.csv" -delete
