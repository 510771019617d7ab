O=gpurun_out/r3_next; mkdir -p $O
./tools/pmc_calib.bin > $O/calib_bytes.txt; cat $O/calib_bytes.txt
bash tools/r3_pmc.sh r3_next/pmc_calib calib
bash tools/r3_lanes_cfg.sh
