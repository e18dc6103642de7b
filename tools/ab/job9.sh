cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_trajectory_gpu.py -x -q 2>&1 | tail -4 > gpurun_out/j9_traj.txt; cat gpurun_out/j9_traj.txt
python tools/ab_halo_l0.py > gpurun_out/j9_halo_l0.txt 2>&1; cat gpurun_out/j9_halo_l0.txt
python tools/profile_shapes.py 25 > gpurun_out/r04_per_shape_25step.txt 2>&1; head -30 gpurun_out/r04_per_shape_25step.txt | cut -c1-150
python tools/profile_sn.py 1 > gpurun_out/r04_sn_per_shape.txt 2>&1; head -50 gpurun_out/r04_sn_per_shape.txt | cut -c1-150
rm -rf gpurun_out/prof_r04
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04 -o bench -- python bench.py --steps 14 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r04_bench_line_under_rocprof.json 2> gpurun_out/r04_bench_under_rocprof.err
f=$(find gpurun_out/prof_r04 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r04_bench_kernel_stats_rocprofv3.csv; head -12 gpurun_out/r04_bench_kernel_stats_rocprofv3.csv | cut -c1-160
rm -rf gpurun_out/prof_r04
bash tools/pmc_traffic.sh 5 > gpurun_out/j9_pmc_traffic.log 2>&1; cat gpurun_out/pmc_traffic_gemm.json | head -30
bash tools/pmc_mfma_util.sh 5 > gpurun_out/j9_mfma_util.log 2>&1; cat gpurun_out/pmc_mfma_util.txt
