timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "ln_linear or stream_gemm" 2>&1 | tail -8
python tools/time_clip.py 3 2>&1 | tail -1
UG_TUNE_KNOBS=262144 python tools/time_clip.py 3 2>&1 | tail -1
python tools/time_clip.py 3 2>&1 | tail -1
UG_TUNE_KNOBS=262144 python tools/time_clip.py 3 2>&1 | tail -1
