O=gpurun_out/r3_b21; mkdir -p $O
for q in 16 12; do for l in 7 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --config 2 --lanes $l --no-cpu-baseline --no-single > $O/b.json 2> $O/err.txt || tail -3 $O/err.txt
  python -c "
import json; d=json.load(open('$O/b.json')); print('hw queues $q config 2 lanes $l value %9.1f ms/step %.4f streams %s' % (d['value'], d['ms_per_step'], d['config']['streams_concurrent']))"
done; done
for l in 7 8; do
  GPU_MAX_HW_QUEUES=16 python bench.py --config 3 --lanes $l --no-cpu-baseline --no-single > $O/b.json 2> $O/err.txt || tail -3 $O/err.txt
  python -c "
import json; d=json.load(open('$O/b.json')); print('hw queues 16 config 3 lanes $l value %9.1f ms/step %.4f streams %s' % (d['value'], d['ms_per_step'], d['config']['streams_concurrent']))"
done
