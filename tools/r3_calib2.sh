O=gpurun_out/r3_calib2; mkdir -p $O
./tools/pmc_calib.bin > $O/calib_bytes.txt; cat $O/calib_bytes.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- $GRAFT_REPO_ROOT/tools/pmc_calib.bin > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); cut -d, -f1-4 $f | cut -c1-120
bash tools/r3_pmc.sh r3_calib2/pmc_calib calib
grep "calib_valu" $O/pmc_calib/pmc_summary.csv | grep "INSTS_VALU\|GRBM\|ACTIVE_INST_VALU\|BUSY_CYCLES"
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
