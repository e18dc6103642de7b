for i in 1 2; do
python tools/time_clip.py 3 2>&1 | tail -1
UG_STAT_MINM=4096 python tools/time_clip.py 3 2>&1 | tail -1
UG_TUNE_KNOBS=131072 python tools/time_clip.py 3 2>&1 | tail -1
done
python -m pytest tests/test_fullsize_golden_gpu.py tests/test_trajectory_gpu.py tests/test_pipeline_gpu.py -x -q 2>&1 | tail -3
