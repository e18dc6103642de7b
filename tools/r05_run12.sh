cd $GRAFT_REPO_ROOT
export UG_COSCHED=1
echo "base      :" $(python tools/two_clips_in_flight.py 3 3 2>&1 | tail -1)
echo "split 1   :" $(UG_TUNE_SPLIT=1 python tools/two_clips_in_flight.py 3 3 2>&1 | tail -1)
echo "k2048     :" $(UG_COSCHED_KNOBS=2048 python tools/two_clips_in_flight.py 3 3 2>&1 | tail -1)
echo "k8192+512 :" $(UG_COSCHED_KNOBS=8704 python tools/two_clips_in_flight.py 3 3 2>&1 | tail -1)
echo "k1024     :" $(UG_COSCHED_KNOBS=1024 python tools/two_clips_in_flight.py 3 3 2>&1 | tail -1)
echo "k131072   :" $(UG_COSCHED_KNOBS=131072 python tools/two_clips_in_flight.py 3 3 2>&1 | tail -1)
echo "base      :" $(python tools/two_clips_in_flight.py 3 3 2>&1 | tail -1)
