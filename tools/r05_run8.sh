cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/attribute_rounding.py > gpurun_out/r05_rounding_attribution.txt 2>&1
( timeout 1500 python -m pytest tests/test_fullsize_golden_gpu.py -x -q -s 2>&1 | grep -v "^PARITY" | tail -25 ) > gpurun_out/r05_fullsize_fp16emu.txt
cat gpurun_out/r05_rounding_attribution.txt gpurun_out/r05_fullsize_fp16emu.txt
