cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_ops_gpu.py -x -q 2>&1 | tail -15 ) > gpurun_out/r05_ops_tests.txt
timeout 900 python tools/ab_gemm.py "-1,-1k128,-1k1048576" > gpurun_out/r05_ab_walk_gemm.txt 2>&1
timeout 600 python tools/ab_clip.py walk 3 > gpurun_out/r05_ab_clip_walk.txt 2>&1
timeout 400 python tools/ab_clip.py rowmajor 2 > gpurun_out/r05_ab_clip_rowmajor.txt 2>&1
SHAPES="ff2l1" ONLY="1 2 3 6 7" OUT=gpurun_out/r05_l2_ff2l1_new.txt bash tools/pmc_l2.sh > /dev/null 2>&1
UG_KNOBS=128 SHAPES="ff2l1" ONLY="1 2 3 6 7" OUT=gpurun_out/r05_l2_ff2l1_roundstrided.txt bash tools/pmc_l2.sh > /dev/null 2>&1
UG_KNOBS=1048704 SHAPES="ff2l1" ONLY="1 2 3 6 7" OUT=gpurun_out/r05_l2_ff2l1_r4walk.txt bash tools/pmc_l2.sh > /dev/null 2>&1
tail -5 gpurun_out/r05_ops_tests.txt; cat gpurun_out/r05_ab_clip_walk.txt gpurun_out/r05_ab_clip_rowmajor.txt; cat gpurun_out/r05_ab_walk_gemm.txt
