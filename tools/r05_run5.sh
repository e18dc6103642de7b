cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "layernorm" 2>&1 | tail -5 ) > gpurun_out/r05_pre_ops.txt
timeout 600 python tools/ab_clip.py lnfold 3 > gpurun_out/r05_ab_clip_lnfold2.txt 2>&1
UG_LN_FOLD=1 timeout 600 python tools/profile_shapes.py 25 > gpurun_out/r05_shapes_lnfold1b.txt 2>&1
cat gpurun_out/r05_pre_ops.txt gpurun_out/r05_ab_clip_lnfold2.txt
grep -n "^total\|lnf\|layernorm\|gemm_linear:19200x640x640\|gemm_linear:4800x1280x1280 \|gemm_linear:19200x640x2560\|gemm_linear:4800x1280x5120" gpurun_out/r05_shapes_lnfold1b.txt | cut -c1-130
