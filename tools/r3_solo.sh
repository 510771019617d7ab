python tools/solo_bench.py 3
python tools/solo_bench.py 3 --torch-first
python tools/solo_bench.py 3 --torch-after
python tools/solo_bench.py 5
python tools/solo_bench.py 5 --torch-first
for m in "1" "2,1" "2,2"; do RT355_OVERLAP_CONNECT=$m python bench.py --steps 64 --warmup 4 --no-cpu-baseline --lanes 1 --no-single > gpurun_out/ov.json 2>gpurun_out/ov.err || tail -3 gpurun_out/ov.err; python -c "
import json; d=json.load(open('gpurun_out/ov.json')); print('overlap connect mode $m: lanes1', d['value'], d['accum_rgb_sum'])"; done
python bench.py --steps 64 --warmup 4 --no-cpu-baseline --lanes 1 --no-single > gpurun_out/ov.json 2>gpurun_out/ov.err; python -c "
import json; d=json.load(open('gpurun_out/ov.json')); print('no overlap           : lanes1', d['value'], d['accum_rgb_sum'])"
