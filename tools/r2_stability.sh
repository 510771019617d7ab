# final-state stability evidence: deep fuzz over fresh seeds + soak runs (outputs under gpurun_out/r2_stab/)
set -e
O=gpurun_out/r2_stab; mkdir -p $O
{ echo "## mixed"; python tools/deep_fuzz.py 8000 3000 mixed | tail -1; echo "## tri"; python tools/deep_fuzz.py 8000 3000 | tail -1;
  echo "## bigmixed"; python tools/deep_fuzz.py 800 300 big mixed | tail -1; echo "## big"; python tools/deep_fuzz.py 800 300 big | tail -1; } > $O/deep_fuzz.txt 2>&1
{ python bench.py --steps 8192 --no-cpu-baseline --no-repeat --no-single 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('python bench.py --steps 8192 --no-cpu-baseline --no-repeat:', d['value'], 'M samples/s over', round(d['ms_per_step']*8192/1000,2), 's of continuous rendering')"
  python tools/soak_lanes.py 2 3 600; python tools/soak_lanes.py 1 3 1500; } > $O/soak.txt 2>&1
cat $O/deep_fuzz.txt $O/soak.txt
