cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/bench_lnf.py > gpurun_out/r05_bench_lnf3.txt 2>&1
timeout 600 python tools/ab_clip.py lnfold 3 > gpurun_out/r05_ab_clip_lnfold4.txt 2>&1
UG_LN_FOLD=1 timeout 600 python tools/profile_shapes.py 25 > gpurun_out/r05_shapes_lnfold1c.txt 2>&1
UG_LN_FOLD=0 timeout 600 python tools/profile_shapes.py 25 > gpurun_out/r05_shapes_lnfold0c.txt 2>&1
cat gpurun_out/r05_ab_clip_lnfold4.txt; grep -v "GEGLU\|plain" gpurun_out/r05_bench_lnf3.txt | head -8
for f in 1c 0c; do echo == $f; grep -n "^total\|lnf\|layernorm\|gemm_linear:19200x\|gemm_linear:4800x" gpurun_out/r05_shapes_lnfold$f.txt | cut -c1-130; done
