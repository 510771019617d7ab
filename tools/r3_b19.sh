O=gpurun_out/r3_b19; mkdir -p $O
for c in 2 5; do for l in 3 4 5 6 7; do
  python bench.py --config $c --lanes $l --steps 96 --no-cpu-baseline --no-single > $O/b.json 2> $O/err.txt || tail -3 $O/err.txt
  python -c "
import json; d=json.load(open('$O/b.json')); print('config $c lanes $l value %9.1f ms/step %.4f streams %s' % (d['value'], d['ms_per_step'], d['config']['streams_concurrent']))"
done; done
