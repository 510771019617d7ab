O=gpurun_out/r3_b22; mkdir -p $O
run() { python bench.py --no-cpu-baseline --no-single "$@" > $O/b.json 2> $O/err.txt || tail -3 $O/err.txt; python -c "
import json; d=json.load(open('$O/b.json')); print('%-32s lanes %d value %9.1f ms/step %.4f streams %s' % ('$*', d['config']['lanes'], d['value'], d['ms_per_step'], d['config']['streams_concurrent']))"; }
run --config 2
run --config 2
run --config 5 --steps 96 --lanes 6
run --config 5 --steps 96 --lanes 8
run --config 3
run --config 3 --lanes 8
run --config 4
run --config 4 --lanes 8
python -m pytest tests -m gpu -q -k "lanes or group or concurr or bench" > $O/t.log 2>&1 || { grep -E "^FAILED|^ERROR" $O/t.log | head; }; tail -1 $O/t.log
