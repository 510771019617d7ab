# per-visit cost of the traversal against scene size (one context): is it the memory system or the machine?
for d in 0.02 0.05 0.1 0.2 0.5 1.0; do
  python bench.py --detail $d --steps 32 --lanes 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); s=d['stage_ms_per_step']; pr=d['roofline']['per_ray']
print('detail $d', d['config']['workload'][:60], 'single', d['value'], s, pr, 'extend ms per (node visit + prim test) per ray', round(s['extend']/(pr['node_visits']+pr['prim_tests']),4))"
done
