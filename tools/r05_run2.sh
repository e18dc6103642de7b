cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "layernorm or linear or tile_configs" 2>&1 | tail -25 ) > gpurun_out/r05_lnfold_ops.txt
( timeout 900 python -m pytest tests/test_stages_gpu.py -x -q -k "unet" 2>&1 | tail -25 ) > gpurun_out/r05_lnfold_stages.txt
cat gpurun_out/r05_lnfold_ops.txt gpurun_out/r05_lnfold_stages.txt
