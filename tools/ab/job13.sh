python tools/time_clip.py 3 > gpurun_out/j13_a.txt 2>&1
UG_TUNE_KNOBS=65536 python tools/time_clip.py 3 > gpurun_out/j13_b.txt 2>&1
python tools/time_clip.py 3 >> gpurun_out/j13_a.txt 2>&1
UG_TUNE_KNOBS=65536 python tools/time_clip.py 3 >> gpurun_out/j13_b.txt 2>&1
cat gpurun_out/j13_a.txt gpurun_out/j13_b.txt
python -m pytest tests/test_ops_gpu.py -x -q -k "stream_gemm or linear" 2>&1 | tail -3
