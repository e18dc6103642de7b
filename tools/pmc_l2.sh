#!/bin/bash
# L2 / vector-cache counter passes for the two mid-shape GEMMs of VERDICT r3 item 2 (prove or kill the "L2 -> CU bandwidth" reading):
#   ff1l1 = 19200x5120x640, ff2l1 = 19200x640x2560, planner's tile choice (cfg -1).  Counters only, one group per pass (gpurun rule).
#   round 5: passes 6 - 8 = the L2's memory-side requests (reads / writes separately: TCC_MISS counts write misses too); UG_KNOBS=<mask> is passed on to
#   tools/gemm_one.py (128 = the round-strided XCD walk of rounds 1 - 4), OUT=<file> names the output.
# usage (through gpurun): bash tools/pmc_l2.sh -> gpurun_out/gemm_l2_counters.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${OUT:-gpurun_out/gemm_l2_counters.txt}
[ -n "$ONLY" ] || : > $OUT
for N in ${SHAPES:-ff1l1 ff2l1 sq8k}; do
  i=0
  for G in "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCC_HIT_sum TCC_MISS_sum" \
           "TCC_REQ_sum TCC_READ_sum" \
           "SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "FETCH_SIZE" \
           "WRITE_SIZE" \
           "TCC_WRITE_sum TCC_WRITEBACK_sum"; do
    i=$((i+1))
    if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $i "; then continue; fi
    rm -rf gpurun_out/pmcl2_${N}_$i
    timeout 300 rocprofv3 --pmc $G --output-format csv -d gpurun_out/pmcl2_${N}_$i -o pmc -- python tools/gemm_one.py $N ${CFG:--1} 3 > gpurun_out/pmcl2_${N}_$i.log 2>&1 || echo "pass $N/$i failed (see log)" >> $OUT
  done
  python - >> $OUT <<PY
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmcl2_${N}_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list); kn = None
        for r in csv.DictReader(open(f)):
            if "gemm" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"])); kn = r["Kernel_Name"][:70]
        for k, v in acc.items():
            print(f"${N} [{kn}]: {k:36s} per-launch mean {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
  tail -n 3 gpurun_out/pmcl2_${N}_1.log >> $OUT
  rm -rf gpurun_out/pmcl2_${N}_*/
done
cat $OUT
