O=gpurun_out/r3_fuzz2; mkdir -p $O
python tools/deep_fuzz.py 2000 6000 multi > $O/fuzz_multi_small.txt 2>&1; tail -2 $O/fuzz_multi_small.txt
python tools/deep_fuzz.py 10000 1500 multi big > $O/fuzz_multi_big.txt 2>&1; tail -2 $O/fuzz_multi_big.txt
python tools/deep_fuzz.py 20000 1500 > $O/fuzz_soups_small.txt 2>&1; tail -1 $O/fuzz_soups_small.txt
python tools/deep_fuzz.py 30000 400 big > $O/fuzz_soups_big.txt 2>&1; tail -1 $O/fuzz_soups_big.txt
python tools/deep_fuzz.py 40000 800 mixed > $O/fuzz_mixed.txt 2>&1; tail -1 $O/fuzz_mixed.txt
