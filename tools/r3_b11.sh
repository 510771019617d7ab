O=gpurun_out/r3_b11; mkdir -p $O
run() { cfg=$1; shift; envs=""; args=""; for a in "$@"; do case $a in *=*) envs="$envs $a";; *) args="$args $a";; esac; done
  env $envs python bench.py --config $cfg --steps 96 --no-cpu-baseline $args > $O/b.json 2>$O/err.txt || tail -3 $O/err.txt; python -c "
import json; d=json.load(open('$O/b.json')); print('config $cfg %-40s lanes %d: %8.1f one context %8.1f streams %s' % ('$*', d['config']['lanes'], d['value'], d['value_single_context'] or 0, d['config']['streams_concurrent']))"; }
run 3 --no-single
run 3 --lanes 6 --persist-blocks 1 --no-single
run 3 --lanes 5 --persist-blocks 1 --no-single
run 3 --lanes 3 --persist-blocks 2 --no-single
run 3 --lanes 4 --persist-blocks 3 --no-single
run 5 --persist-blocks 1 --no-single
run 5 --persist-blocks 3 --no-single
run 4 --lanes 6 --persist-blocks 1 --no-single
