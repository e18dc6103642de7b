#!/bin/bash
# LDS / issue counters of the flash-attention kernel (tools/bench_flash.py on the clip's shapes): bank conflicts vs LDS-array cycles, instruction mix
# and wait cycles.  Counters only (no trace domains), one pass per counter group.  usage (through gpurun): bash tools/pmc_flash.sh -> gpurun_out/pmc_flash.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_fa; : > gpurun_out/pmc_flash.txt
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc_fa/g$i -o pmc -- python tools/bench_flash.py one > gpurun_out/pmc_fa_$i.log 2>&1 || echo "group $i failed: $grp" >> gpurun_out/pmc_flash.txt
done
python - <<'PY'
import csv, glob, collections
acc, n = collections.defaultdict(float), collections.Counter()
for f in glob.glob("gpurun_out/pmc_fa/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "flash_attn64_kernel" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
with open("gpurun_out/pmc_flash.txt", "a") as o:
    o.write("flash_attn64_kernel (B25 x H5 x S3072), sums over the profiled dispatches / per dispatch\n")
    for k in sorted(acc): o.write(f"{k:28s} {acc[k]:16.0f}   dispatches {n[k]:4d}   per dispatch {acc[k] / max(n[k], 1):14.0f}\n")
print(open("gpurun_out/pmc_flash.txt").read())
PY
rm -rf gpurun_out/pmc_fa
