# kernel stats of the default command and of one context, per-bounce table: end-state library
O=gpurun_out/r3_end2; mkdir -p $O
bash tools/trace_default.sh r3_end2/trace_default --no-single > $O/trace_default.txt 2>&1; tail -12 $O/trace_default.txt | cut -c1-180
bash tools/trace.sh r3_end2/trace_lanes1 --lanes 1 --no-single --no-repeat > $O/trace_lanes1.txt 2>&1; tail -7 $O/trace_lanes1.txt
bash tools/trace.sh r3_end2/trace_c2 --config 2 --lanes 1 --no-single --no-repeat > $O/trace_c2.txt 2>&1; tail -7 $O/trace_c2.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
