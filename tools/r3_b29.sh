O=gpurun_out/r3_b29; mkdir -p $O
bash tools/trace_default.sh r3_b29/trace_c5 --config 5 --no-single --steps 64 > $O/trace_c5.txt 2>&1; tail -3 $O/trace_c5.txt | cut -c1-150
bash tools/trace_default.sh r3_b29/trace_c4 --config 4 --no-single --steps 64 > $O/trace_c4.txt 2>&1; tail -3 $O/trace_c4.txt | cut -c1-150
bash tools/trace_default.sh r3_b29/trace_c2 --config 2 --no-single > $O/trace_c2.txt 2>&1; tail -3 $O/trace_c2.txt | cut -c1-150
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
