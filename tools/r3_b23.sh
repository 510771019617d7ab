O=gpurun_out/r3_b23; mkdir -p $O
bash tools/r3_pmc.sh r3_b23/pmc_config5_lanes8 --config 5
bash tools/r3_pmc.sh r3_b23/pmc_config5_lanes1 --config 5 --lanes 1
bash tools/r3_pmc.sh r3_b23/pmc_config2_lanes8 --config 2
bash tools/r3_pmc.sh r3_b23/pmc_config2_lanes1 --config 2 --lanes 1
