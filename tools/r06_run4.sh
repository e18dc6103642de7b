cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_stages_gpu.py -x -q 2>&1 | tail -6 ) > gpurun_out/r06_t_after_fold_removal.txt
for i in 1 2; do
UG_LIB_PATH=$GRAFT_REPO_ROOT/unigeo_amd/csrc/build/libunigeo_r5.so timeout 300 python tools/ab_lib.py 2>&1 | tail -1
timeout 300 python tools/ab_lib.py 2>&1 | tail -1
done > gpurun_out/r06_ab_after_fold_removal.txt
cat gpurun_out/r06_t_after_fold_removal.txt gpurun_out/r06_ab_after_fold_removal.txt
