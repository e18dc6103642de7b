cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/ab_clip.py lnfold 3 > gpurun_out/r05_ab_clip_lnfold.txt 2>&1
timeout 600 python tools/profile_shapes.py 25 > gpurun_out/r05_shapes_lnfold1.txt 2>&1
UG_LN_FOLD=0 timeout 600 python tools/profile_shapes.py 25 > gpurun_out/r05_shapes_lnfold0.txt 2>&1
cat gpurun_out/r05_ab_clip_lnfold.txt; head -45 gpurun_out/r05_shapes_lnfold1.txt | cut -c1-150
