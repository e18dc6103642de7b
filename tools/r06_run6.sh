cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# which other runtime switches matter?  one clip (tools/ab_lib.py: best of 4) per setting, same box
run() { echo "$1 : $(env $1 timeout 300 python tools/ab_lib.py 2>&1 | tail -1)"; }
{
run "HIP_FORCE_DEV_KERNARG=0"
run "HIP_FORCE_DEV_KERNARG=1"
run "UG_NOP=1"
run "ROC_SKIP_KERNEL_ARG_COPY=1"
run "HSA_NO_SCRATCH_RECLAIM=1"
run "GPU_MAX_HW_QUEUES=1"
run "GPU_MAX_HW_QUEUES=2"
run "AMD_DIRECT_DISPATCH=0"
run "ROC_ACTIVE_WAIT_TIMEOUT=1000"
run "HSA_ENABLE_INTERRUPT=0"
run "UG_NOP=2"
} > gpurun_out/r06_runtime_switches.txt 2>&1
cat gpurun_out/r06_runtime_switches.txt
