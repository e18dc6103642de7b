cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "split_k_fixup" 2>&1 | tail -12 ) > gpurun_out/r06_t_ticket.txt
( timeout 1500 python -m pytest tests/test_ops_gpu.py -x -q 2>&1 | tail -6 ) > gpurun_out/r06_t_ops_all.txt
timeout 600 python tools/ab_clip.py ticket 3 > gpurun_out/r06_ab_ticket_clip.txt 2>&1
timeout 600 python tools/ab_sn.py 8388608 3 > gpurun_out/r06_ab_ticket_sn.txt 2>&1
cat gpurun_out/r06_t_ticket.txt gpurun_out/r06_t_ops_all.txt; tail -7 gpurun_out/r06_ab_ticket_clip.txt; tail -7 gpurun_out/r06_ab_ticket_sn.txt
