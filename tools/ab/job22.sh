python -m pytest tests/test_ops_gpu.py -x -q -k "statistics_from_conv or conv_halo or groupnorm" 2>&1 | tail -15
for i in 1 2; do
python tools/time_clip.py 3 2>&1 | tail -1
UG_TUNE_KNOBS=131072 python tools/time_clip.py 3 2>&1 | tail -1
done
