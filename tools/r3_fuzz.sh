O=gpurun_out/r3_fuzz; mkdir -p $O
python -m pytest tests -m gpu -q -k "multi_blas or deeper_tlas" > $O/t.log 2>&1 || { grep -E "^FAILED|^ERROR|Error|assert" $O/t.log | head -20; }
tail -2 $O/t.log
python tools/deep_fuzz.py 100 400 multi 2>&1 | tee $O/fuzz_multi_small.txt | tail -12
python tools/deep_fuzz.py 1000 120 multi big 2>&1 | tee $O/fuzz_multi_big.txt | tail -8
