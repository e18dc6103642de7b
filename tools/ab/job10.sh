python tools/time_clip.py 3 > gpurun_out/j10_clip_a.txt 2>&1
UG_TUNE_KNOBS=32768 python tools/time_clip.py 3 > gpurun_out/j10_clip_l0off.txt 2>&1
python tools/time_clip.py 3 >> gpurun_out/j10_clip_a.txt 2>&1
UG_TUNE_KNOBS=16384 python tools/time_clip.py 3 > gpurun_out/j10_clip_halooff.txt 2>&1
cat gpurun_out/j10_clip_*.txt
python -m pytest tests/test_ops_gpu.py -x -q -k "conv" 2>&1 | tail -3
