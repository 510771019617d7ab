# round-3 GPU run 3: what still separates the REF_BUILTINS build from the reference's shade kernel; config-5 per-bounce A/B.
O=gpurun_out/r3_run3; mkdir -p $O
python tools/diag_refb.py nee > $O/diag_refb_nee.txt 2>&1; head -60 $O/diag_refb_nee.txt
python tools/diag_refb.py fisheye > $O/diag_refb_fisheye.txt 2>&1; head -30 $O/diag_refb_fisheye.txt
bash tools/trace.sh r3_run3/c5_nested --config 5 --lanes 1 --no-single --no-repeat --extend-variant 4 > $O/c5_nested.txt 2>&1; cat $O/c5_nested.txt
bash tools/trace.sh r3_run3/c5_event --config 5 --lanes 1 --no-single --no-repeat > $O/c5_event.txt 2>&1; cat $O/c5_event.txt
RT355_TLAS_FLAT=1,1 bash tools/trace.sh r3_run3/c5_flat --config 5 --lanes 1 --no-single --no-repeat > $O/c5_flat.txt 2>&1; cat $O/c5_flat.txt
RT355_TLAS_FLAT=1,0 python bench.py --config 5 --lanes 1 --steps 32 --warmup 2 --no-cpu-baseline > $O/bench_c5_flat_event.json 2>$O/err.txt; python -c "
import json; d=json.load(open('$O/bench_c5_flat_event.json')); print('config5 flat extend + event connect', d['value'], d['value_single_context'], d['stage_ms_per_step'])"
RT355_TLAS_FLAT=1,0 python bench.py --config 5 --steps 32 --warmup 2 --no-cpu-baseline --no-single > $O/bench_c5_flat_event_l4.json 2>$O/err.txt; python -c "
import json; d=json.load(open('$O/bench_c5_flat_event_l4.json')); print('config5 4 lanes flat extend + event connect', d['value'])"
rm -rf $O/c5_*/*/  # traces are large
