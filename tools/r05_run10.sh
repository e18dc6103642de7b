cd $GRAFT_REPO_ROOT
for k in 0 4194304 2048 1024 8192 512 4096; do
  echo "knobs $k:" $(UG_FF_NOSPLIT=1 UG_TUNE_KNOBS=$k python tools/two_clips_in_flight.py 2 4 2>&1 | tail -1)
done
