#!/bin/bash
# SQ counter passes for one GEMM problem: usage pmc_gemm.sh <name> <cfg>.  Counters only (no trace domains), one group per pass.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
N=$1; C=$2; i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
         "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d gpurun_out/pmcg_${N}_${C}_$i -o pmc -- python tools/gemm_one.py $N $C 3 > gpurun_out/pmcg_${N}_${C}_$i.log 2>&1 || true
done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmcg_${N}_${C}_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "gemm" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(f"${N} cfg ${C}: {k:32s} per-launch mean {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
