python -m pytest tests/test_ops_gpu.py -x -q -k "statistics_from_conv" 2>&1 | tail -3
python tools/profile_shapes.py 2 > gpurun_out/shapes_new.txt 2>&1; grep "gemm_tconv:1572\|gemm_tconv:3932\|gemm_tconv:98304\|groupnorm:T8\|^total" gpurun_out/shapes_new.txt
UG_STAT_DEBUG=1 python tools/one_clip.py 1 2>&1 | grep "^\[stat\]" | grep "rb 0" | sort | uniq -c | sort -rn
python -m pytest tests/test_pipeline_gpu.py tests/test_fullsize_golden_gpu.py -x -q 2>&1 | grep -E "passed|failed"
for i in 1 2; do python tools/time_clip.py 3 2>&1 | tail -1; UG_TUNE_KNOBS=524288 python tools/time_clip.py 3 2>&1 | tail -1; done
