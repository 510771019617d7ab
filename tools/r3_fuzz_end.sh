# deep fuzz of the end-state library (random scenes vs the oracle, bit for bit): multi-BLAS, triangle soups, mixed primitives
O=gpurun_out/r3_fuzz_end; mkdir -p $O
python tools/deep_fuzz.py 200000 6000 multi > $O/fuzz_multi_small.txt 2>&1; tail -1 $O/fuzz_multi_small.txt
python tools/deep_fuzz.py 210000 1200 multi big > $O/fuzz_multi_big.txt 2>&1; tail -1 $O/fuzz_multi_big.txt
python tools/deep_fuzz.py 220000 6000 > $O/fuzz_soups_small.txt 2>&1; tail -1 $O/fuzz_soups_small.txt
python tools/deep_fuzz.py 230000 600 big > $O/fuzz_soups_big.txt 2>&1; tail -1 $O/fuzz_soups_big.txt
python tools/deep_fuzz.py 240000 3000 mixed > $O/fuzz_mixed.txt 2>&1; tail -1 $O/fuzz_mixed.txt
