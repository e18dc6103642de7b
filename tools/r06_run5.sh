cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
for v in 0 1; do
echo "HIP_FORCE_DEV_KERNARG=$v clip: $(HIP_FORCE_DEV_KERNARG=$v timeout 300 python tools/ab_lib.py 2>&1 | tail -1)"
done; done > gpurun_out/r06_dev_kernarg.txt
for v in 0 1; do
echo "HIP_FORCE_DEV_KERNARG=$v sn: $(HIP_FORCE_DEV_KERNARG=$v timeout 300 python tools/sn_wall.py 2>&1 | tail -2 | tr '\n' ' ')"
done >> gpurun_out/r06_dev_kernarg.txt
cat gpurun_out/r06_dev_kernarg.txt
