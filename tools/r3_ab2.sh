O=$GRAFT_REPO_ROOT/gpurun_out/r3_ab2; mkdir -p $O
for order in lib-first torch-first lib-first torch-first; do
RT355_IMPORT_ORDER=$order python bench.py --steps 64 --warmup 4 --no-cpu-baseline --lanes 1 --no-single > $O/b.json 2>$O/err.txt; grep libamdhip $O/err.txt; python -c "
import json; d=json.load(open('$O/b.json')); print('$order lanes1 ', d['value'])"
done
RT355_IMPORT_ORDER=torch-first python bench.py --steps 64 --warmup 4 --no-cpu-baseline > $O/b.json 2>$O/err.txt; python -c "
import json; d=json.load(open('$O/b.json')); print('torch-first 4 lanes', d['value'], d['value_single_context'], d['config']['streams_concurrent'])"
./examples/headless_tick --size 1920 1080 --spp 256 --out $O/t.png | tail -1
./examples/headless_tick --size 1920 1080 --spp 256 --lanes 4 --out $O/t.png | tail -1
