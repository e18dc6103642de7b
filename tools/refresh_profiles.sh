#!/bin/bash
# usage (through gpurun): bash tools/refresh_profiles.sh  -> gpurun_out/r06_*; copy what is to be judged into profiles/
# round-6 measurement refresh: default bench line, rocprofv3 kernel stats of the same workload (one clip per run call = the headline), PMC traffic / MFMA
# utilisation, per-shape table
python bench.py > gpurun_out/r06_bench_line_final.json 2> gpurun_out/r06_bench_final.err; tail -c 600 gpurun_out/r06_bench_line_final.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_r06
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r06 -o bench -- python bench.py --steps 10 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r06_bench_line_under_rocprof.json 2> gpurun_out/r06_bench_under_rocprof.err
f=$(find gpurun_out/prof_r06 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r06_bench_kernel_stats_rocprofv3.csv; head -12 gpurun_out/r06_bench_kernel_stats_rocprofv3.csv | cut -c1-160
rm -rf gpurun_out/prof_r06
bash tools/pmc_traffic.sh 5 2>&1 | tail -3; cp gpurun_out/pmc_traffic_gemm.json gpurun_out/r06_pmc_traffic_gemm.json
bash tools/pmc_mfma_util.sh 5 2>&1 | tail -3; cp gpurun_out/pmc_mfma_util.txt gpurun_out/r06_pmc_mfma_util_5step.txt; head -16 gpurun_out/pmc_mfma_util.txt
python tools/profile_shapes.py 25 > gpurun_out/r06_per_shape_hip_events_25step.txt 2>&1; head -5 gpurun_out/r06_per_shape_hip_events_25step.txt
