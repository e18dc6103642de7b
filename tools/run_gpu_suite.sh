python -m pytest tests -q -m gpu > gpurun_out/r06_gpu_suite.txt 2>&1
grep -E "passed|failed|rror" gpurun_out/r06_gpu_suite.txt | tail -5
