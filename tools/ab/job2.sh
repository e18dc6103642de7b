python -m pytest tests/test_ops_gpu.py -x -q -k "groupnorm" 2>&1 | tail -5 > gpurun_out/j2_gn_tests.txt
python tools/bench_groupnorm.py > gpurun_out/j2_gn_bench.txt 2>&1
python tools/ablate_halo.py > gpurun_out/j2_halo_ablation.txt 2>&1
python tools/time_clip.py 3 > gpurun_out/j2_clip_default.txt 2>&1
UG_GN_NOFOLD=1 UG_GN_NOT2=1 python tools/time_clip.py 3 > gpurun_out/j2_clip_gn_old.txt 2>&1
python tools/time_clip.py 3 >> gpurun_out/j2_clip_default.txt 2>&1
UG_GN_NOT2=1 python tools/time_clip.py 3 > gpurun_out/j2_clip_gn_not2.txt 2>&1
cp gpurun_out/gemm_l2_counters.txt gpurun_out/gemm_l2_counters_prev.txt 2>/dev/null
SHAPES="ff1l1 ff2l1" ONLY="2 3" bash tools/pmc_l2.sh > gpurun_out/j2_pmc_l2.log 2>&1
cat gpurun_out/j2_gn_tests.txt gpurun_out/j2_gn_bench.txt gpurun_out/j2_halo_ablation.txt gpurun_out/j2_clip_*.txt; grep -v "^[EWIF]2026" gpurun_out/gemm_l2_counters.txt | tail -12
