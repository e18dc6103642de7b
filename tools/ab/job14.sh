for f in 0 1 2 3 4; do echo "== UG_STREAM_FORM=$f"; UG_STREAM_FORM=$f python tools/ab_stream.py 2>&1 | grep "^76800\|^19200"; done
python -m pytest tests/test_ops_gpu.py -x -q -k "stream_gemm" 2>&1 | tail -2
for f in 2 3; do UG_STREAM_FORM=$f python -m pytest tests/test_ops_gpu.py -x -q -k "stream_gemm" 2>&1 | tail -1; done
for f in 0 2 3; do UG_STREAM_FORM=$f python tools/time_clip.py 3 2>&1 | tail -1; done
UG_STREAM_FORM=0 python tools/time_clip.py 3 2>&1 | tail -1
