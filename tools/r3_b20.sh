O=gpurun_out/r3_b20; mkdir -p $O
for rep in 1 2; do for l in 6 7 8; do
  python bench.py --config 2 --lanes $l --no-cpu-baseline --no-single > $O/b.json 2> $O/err.txt || tail -3 $O/err.txt
  python -c "
import json; d=json.load(open('$O/b.json')); print('config 2 lanes $l value %9.1f ms/step %.4f streams %s repeats %s' % (d['value'], d['ms_per_step'], d['config']['streams_concurrent'], d['repeats']))"
done; done
for l in 6 7; do
  python bench.py --config 3 --lanes $l --no-cpu-baseline --no-single > $O/b.json 2> $O/err.txt || tail -3 $O/err.txt
  python -c "
import json; d=json.load(open('$O/b.json')); print('config 3 lanes $l value %9.1f ms/step %.4f streams %s' % (d['value'], d['ms_per_step'], d['config']['streams_concurrent']))"
done
