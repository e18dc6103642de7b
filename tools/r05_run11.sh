cd $GRAFT_REPO_ROOT
for i in 1 2; do
  echo "default:" $(python tools/two_clips_in_flight.py 2 4 2>&1 | tail -1)
  echo "cosched:" $(UG_COSCHED=1 python tools/two_clips_in_flight.py 2 4 2>&1 | tail -1)
done
echo "cosched 3:" $(UG_COSCHED=1 python tools/two_clips_in_flight.py 3 3 2>&1 | tail -1)
python -m pytest tests/test_fullsize_golden_gpu.py -x -q 2>&1 | tail -3
grep coscheduled gpurun_out/parity_measured.jsonl | tail -1
