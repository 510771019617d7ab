O=gpurun_out/r3_b14; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputests.log 2>&1 || { grep -E "^FAILED|^ERROR" $O/gputests.log | head -30; }
tail -2 $O/gputests.log
python tools/deep_fuzz.py 130000 1000 multi > $O/fuzz_multi.txt 2>&1; tail -1 $O/fuzz_multi.txt
python tools/deep_fuzz.py 140000 1000 > $O/fuzz_soups.txt 2>&1; tail -1 $O/fuzz_soups.txt
for x in 0 16384 0 16384; do
  for c in 2 5; do
    RT355_XCD_RAYS=$x python bench.py --config $c --steps 96 --no-cpu-baseline > $O/b_${c}_$x.json 2>$O/err.txt || tail -3 $O/err.txt
    python -c "
import json; d=json.load(open('$O/b_${c}_$x.json')); print('xcd_rays $x config $c lanes %d: %8.1f one context %8.1f extend %s' % (d['config']['lanes'], d['value'], d['value_single_context'] or 0, d['stage_ms_per_step']['extend']))"
  done
done
python bench.py --no-cpu-baseline > $O/bench.json 2>$O/err.txt || tail -3 $O/err.txt
python -c "
import json; d=json.load(open('$O/bench.json')); print('config 3: %8.1f one context %8.1f' % (d['value'], d['value_single_context']))"
