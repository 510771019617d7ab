# which HIP runtime knob separates the un-profiled ROCm 7.2 runtime from rocprofv3 / the ROCm 7.0 runtime for a single in-order stream?
O=$GRAFT_REPO_ROOT/gpurun_out/r3_envsweep; mkdir -p $O
run() { env "$@" python bench.py --steps 64 --warmup 4 --no-cpu-baseline --lanes 1 --no-single > $O/b.json 2>$O/err.txt; python -c "
import json; d=json.load(open('$O/b.json')); print('%-40s config3 lanes1 %8.1f' % ('$*', d['value']))"; }
run RT355_X=0
run GPU_FORCE_QUEUE_PROFILING=1
run AMD_OPT_FLUSH=0
run AMD_OPT_FLUSH=1
run ROC_SYSTEM_SCOPE_SIGNAL=0
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_HIP_KERNARG_COPY_OPT=0
run ROC_ACTIVE_WAIT_TIMEOUT=0
run AMD_DIRECT_DISPATCH=0
run DEBUG_CLR_MAX_BATCH_SIZE=1
run GPU_FLUSH_ON_EXECUTION=1
run ROC_USE_FGS_KERNARG=0
run DEBUG_HIP_DYNAMIC_QUEUES=0
run RT355_IMPORT_ORDER=torch-first
