cd $GRAFT_REPO_ROOT
export UG_COSCHED=1
for i in 1 2; do
echo "rule on  :" $(python tools/two_clips_in_flight.py 3 3 2>&1 | tail -1)
echo "rule off :" $(UG_COSCHED_KNOBS=8388608 python tools/two_clips_in_flight.py 3 3 2>&1 | tail -1)
done
