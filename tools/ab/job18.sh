python -m pytest tests/test_ops_gpu.py -x -q -k "layernorm or ln_ff" 2>&1 | tail -2
for r in 1 2 4; do UG_LN_RPW=$r python tools/profile_shapes.py 5 2>&1 | grep "^total\|^layernorm" | cut -c1-130 | tr '\n' ' '; echo " RPW=$r"; done
for r in 2 1 4 2 1; do UG_LN_RPW=$r python tools/time_clip.py 3 2>&1 | tail -1; done
