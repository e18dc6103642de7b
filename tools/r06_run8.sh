cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out; rm -rf /tmp/prof_gap
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_gap -- python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r06_gap_bench.json 2> gpurun_out/r06_gap_bench.err
python tools/gap_analysis.py /tmp/prof_gap > gpurun_out/r06_gap_analysis.txt 2>&1
cat gpurun_out/r06_gap_analysis.txt
