O=gpurun_out/r3_b27; mkdir -p $O
bash tools/r3_pmc.sh r3_b27/pmc_config3_lanes1 --lanes 1
bash tools/r3_pmc.sh r3_b27/pmc_config3_lanes4
bash tools/r3_pmc.sh r3_b27/pmc_config4_lanes1 --config 4 --lanes 1
bash tools/r3_pmc.sh r3_b27/pmc_config4_lanes4 --config 4
