# round-3 GPU run 4: refb diagnostics after the -Bsymbolic fix; traced vs untraced config 5; full GPU suite
O=gpurun_out/r3_run4; mkdir -p $O
python tools/diag_refb.py nee > $O/diag_refb_nee.txt 2>&1; head -40 $O/diag_refb_nee.txt
python tools/diag_refb.py fisheye > $O/diag_refb_fisheye.txt 2>&1; head -30 $O/diag_refb_fisheye.txt
for a in "--no-profile" ""; do
python bench.py --config 5 --lanes 1 --steps 32 --warmup 2 --no-cpu-baseline $a > $O/b.json 2>$O/err.txt; python -c "
import json; d=json.load(open('$O/b.json')); print('config5 default [$a]', d['value'], d['value_single_context'], d['ms_per_step'], d['repeats'], d['stage_ms_per_step'])"
done
python bench.py --config 5 --lanes 1 --steps 24 --warmup 2 --no-cpu-baseline --no-profile --no-single --no-repeat > $O/b.json 2>$O/err.txt; python -c "
import json; d=json.load(open('$O/b.json')); print('config5 as traced but untraced', d['value'], d['ms_per_step'], d['repeats'])"
python bench.py --config 5 --lanes 1 --steps 240 --warmup 2 --no-cpu-baseline --no-profile --no-single --no-repeat > $O/b.json 2>$O/err.txt; python -c "
import json; d=json.load(open('$O/b.json')); print('config5 240 steps untraced', d['value'], d['ms_per_step'], d['repeats'])"
python -m pytest tests -m gpu -q > $O/gputests.log 2>&1 || { grep -E "^FAILED|^ERROR" $O/gputests.log | head -30; }
tail -3 $O/gputests.log
