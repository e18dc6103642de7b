timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r04_gpu_suite.txt
cat gpurun_out/r04_gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
