python tools/time_clip.py 3 > gpurun_out/j6_clip_halo.txt 2>&1
UG_TUNE_KNOBS=16384 python tools/time_clip.py 3 > gpurun_out/j6_clip_nohalo.txt 2>&1
python tools/time_clip.py 3 >> gpurun_out/j6_clip_halo.txt 2>&1
UG_TUNE_KNOBS=32768 python tools/time_clip.py 3 > gpurun_out/j6_clip_halo_l0.txt 2>&1
cat gpurun_out/j6_clip_*.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/j6_gpu_suite.txt
cat gpurun_out/j6_gpu_suite.txt
