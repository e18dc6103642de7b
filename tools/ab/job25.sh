python -m pytest tests/test_fullsize_golden_gpu.py tests/test_trajectory_gpu.py tests/test_pipeline_gpu.py tests/test_stablenormal_gpu.py -x -q > gpurun_out/job25_tests.log 2>&1
grep -E "passed|failed|Error" gpurun_out/job25_tests.log | tail -5
