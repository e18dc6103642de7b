cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "layernorm or linear or tile_configs" 2>&1 | tail -8 ) > gpurun_out/r05_lnf_ops.txt
( timeout 900 python -m pytest tests/test_stages_gpu.py -x -q -k "unet" 2>&1 | tail -8 ) >> gpurun_out/r05_lnf_ops.txt
python tools/bench_lnf.py > gpurun_out/r05_bench_lnf2.txt 2>&1
timeout 600 python tools/ab_clip.py lnfold 3 > gpurun_out/r05_ab_clip_lnfold3.txt 2>&1
cat gpurun_out/r05_lnf_ops.txt gpurun_out/r05_ab_clip_lnfold3.txt gpurun_out/r05_bench_lnf2.txt
