python -m pytest tests/test_ops_gpu.py -x -q -k "statistics_from_conv" 2>&1 | tail -3
UG_STAT_DEBUG=1 python tools/one_clip.py 1 2>&1 | grep "^\[stat\]" | sort | uniq -c | sort -rn
