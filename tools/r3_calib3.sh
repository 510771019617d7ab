O=gpurun_out/r3_calib3; mkdir -p $O
./tools/pmc_calib.bin > $O/calib_bytes.txt; cat $O/calib_bytes.txt | head -3
bash tools/r3_pmc.sh r3_calib3/pmc_calib calib
grep "calib_reread" $O/pmc_calib/pmc_summary.csv | grep "TCC_HIT\|TCC_MISS\|TCC_REQ\|FETCH_SIZE\|TCC_EA0"
