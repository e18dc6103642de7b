python tools/profile_shapes.py 2 > gpurun_out/shapes_new.txt 2>&1; grep "gemm_tconv\|^total" gpurun_out/shapes_new.txt
UG_STAT_DEBUG=1 python tools/one_clip.py 1 2>&1 | grep "^\[stat\]" | grep "kt 3" | sort | uniq -c | sort -rn
for i in 1 2; do python tools/time_clip.py 3 2>&1 | tail -1; done
