timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "stream_gemm" 2>&1 | tail -4
python tools/profile_shapes.py 5 2>&1 | grep "total\|11264\|76800x320x320\|76800x960x320" | cut -c1-140
echo "--- knob 65536 (streaming off)"
UG_TUNE_KNOBS=65536 python tools/profile_shapes.py 5 2>&1 | grep "total\|11264\|76800x320x320\|76800x960x320" | cut -c1-140
python tools/time_clip.py 3 2>&1 | tail -1
UG_TUNE_KNOBS=65536 python tools/time_clip.py 3 2>&1 | tail -1
python tools/time_clip.py 3 2>&1 | tail -1
