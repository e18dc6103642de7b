#!/bin/bash
# usage (through gpurun): bash tools/refresh_profiles.sh  -> gpurun_out/r05_*; copy what is to be judged into profiles/
# final round-5 measurement refresh: default bench line, rocprofv3 kernel stats of the same workload, PMC traffic / MFMA utilisation, per-shape table
python bench.py > gpurun_out/r05_bench_line_final.json 2> gpurun_out/r05_bench_final.err; tail -c 600 gpurun_out/r05_bench_line_final.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_r05
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05 -o bench -- python bench.py --steps 14 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r05_bench_line_under_rocprof.json 2> gpurun_out/r05_bench_under_rocprof.err
f=$(find gpurun_out/prof_r05 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r05_bench_kernel_stats_rocprofv3.csv; head -12 gpurun_out/r05_bench_kernel_stats_rocprofv3.csv | cut -c1-160
rm -rf gpurun_out/prof_r05
# the same with ONE clip in flight: the per-kernel durations that bench.py's roofline pass (HIP events on context 0 alone) must agree with - with three clips in flight
# the kernels of different clips share the CUs and every one of them lasts longer than it does alone
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05 -o bench -- python bench.py --steps 10 --warmup 1 --in-flight 1 --no-extras --no-cpu-baseline > gpurun_out/r05_bench_line_under_rocprof_one_in_flight.json 2> gpurun_out/r05_bench_under_rocprof1.err
f=$(find gpurun_out/prof_r05 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r05_bench_kernel_stats_rocprofv3_one_in_flight.csv; head -6 gpurun_out/r05_bench_kernel_stats_rocprofv3_one_in_flight.csv | cut -c1-160
rm -rf gpurun_out/prof_r05
bash tools/pmc_traffic.sh 5 2>&1 | tail -3
bash tools/pmc_mfma_util.sh 5 2>&1 | tail -3; head -16 gpurun_out/pmc_mfma_util.txt
python tools/profile_shapes.py 25 > gpurun_out/r05_per_shape_25step.txt 2>&1; head -5 gpurun_out/r05_per_shape_25step.txt
