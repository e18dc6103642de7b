cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "layernorm_folded" 2>&1 | tail -5 ) > gpurun_out/r06_t_ops.txt
( timeout 900 python -m pytest tests/test_pipeline_gpu.py -x -q 2>&1 | tail -8 ) > gpurun_out/r06_t_pipe.txt
( timeout 1200 python -m pytest tests/test_fullsize_golden_gpu.py -x -q 2>&1 | tail -8 ) > gpurun_out/r06_t_full.txt
for i in 1 2; do
UG_LIB_PATH=$GRAFT_REPO_ROOT/unigeo_amd/csrc/build/libunigeo_r5.so timeout 300 python tools/ab_lib.py 2>&1 | tail -1
timeout 300 python tools/ab_lib.py 2>&1 | tail -1
done > gpurun_out/r06_ab_dup_proj.txt
cat gpurun_out/r06_t_ops.txt gpurun_out/r06_t_pipe.txt gpurun_out/r06_t_full.txt gpurun_out/r06_ab_dup_proj.txt
