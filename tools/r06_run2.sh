cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/half_clips_in_flight.py 3 > gpurun_out/r06_half_clips.txt 2>&1
UG_COSCHED=1 timeout 900 python tools/half_clips_in_flight.py 3 >> gpurun_out/r06_half_clips.txt 2>&1
( timeout 300 python -m pytest tests/test_pipeline_gpu.py -x -q -k normals 2>&1 | tail -4 ) >> gpurun_out/r06_half_clips.txt
cat gpurun_out/r06_half_clips.txt
