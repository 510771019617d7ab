# how many sample streams per GPU for the other configurations (config 3 was swept in round 2: tools/hwq_sweep*.sh)
O=gpurun_out/r3_lanes_cfg; mkdir -p $O
for c in 2 5 4 3; do for l in 4 6 8; do
  st=64; [ $c = 2 ] && st=128
  python bench.py --config $c --lanes $l --steps $st --no-cpu-baseline --no-single > $O/c${c}_l$l.json 2> $O/err.txt || tail -3 $O/err.txt
  python -c "
import json; d=json.load(open('$O/c${c}_l$l.json')); print('config $c lanes $l value %9.1f ms/step %.4f streams %s' % (d['value'], d['ms_per_step'], d['config']['streams_concurrent']))"
done; done
