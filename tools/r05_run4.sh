cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "layernorm or linear or tile_configs or conv" 2>&1 | tail -5 ) > gpurun_out/r05_pre_ops.txt
UG_LN_FOLD=0 timeout 600 python tools/ab_clip.py epipre 3 > gpurun_out/r05_ab_clip_epipre.txt 2>&1
UG_LN_FOLD=0 timeout 600 python tools/profile_shapes.py 25 > gpurun_out/r05_shapes_pre1.txt 2>&1
UG_LN_FOLD=0 UG_TUNE_KNOBS=2097152 timeout 600 python tools/profile_shapes.py 25 > gpurun_out/r05_shapes_pre0.txt 2>&1
cat gpurun_out/r05_pre_ops.txt gpurun_out/r05_ab_clip_epipre.txt
