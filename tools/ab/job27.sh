python -m pytest tests/test_ops_gpu.py -x -q -k "statistics_from_conv or groupnorm" 2>&1 | tail -2
python -m pytest tests/test_pipeline_gpu.py tests/test_fullsize_golden_gpu.py -x -q 2>&1 | grep -E "passed|failed"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_fin
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fin -o fin -- python tools/one_clip.py 2 > /dev/null 2>&1
f=$(find gpurun_out/prof_fin -name "*kernel_stats.csv" | head -1); grep "gn_finalize" "$f"
rm -rf gpurun_out/prof_fin
for i in 1 2; do python tools/time_clip.py 3 2>&1 | tail -1; done
