#!/bin/bash
# Instruction mix and issue-stall split per kernel (round 6): rocprofv3 --pmc, counters only, two passes over tools/one_clip.py N.
# pass 1: SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_IFETCH
# pass 2: SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_INSTS_LDS SQ_BUSY_CYCLES
# usage (through gpurun): bash tools/pmc_sq_mix.sh [denoise_steps] -> gpurun_out/pmc_sq_mix.txt
STEPS=${1:-1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pmc_sq1 /tmp/pmc_sq2
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_IFETCH --output-format csv -d /tmp/pmc_sq1 -o pmc -- python tools/one_clip.py $STEPS > gpurun_out/pmc_sq1.log 2>&1 || true
timeout 900 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_sq2 -o pmc -- python tools/one_clip.py $STEPS > gpurun_out/pmc_sq2.log 2>&1 || true
python - <<'PY'
import csv, glob, collections, re
val = collections.defaultdict(lambda: collections.Counter()); n = collections.Counter()
for d in ("/tmp/pmc_sq1", "/tmp/pmc_sq2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", ""))[:70]
            val[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
with open("gpurun_out/pmc_sq_mix.txt", "w") as o:
    o.write("per kernel (sum over its dispatches, one clip, tools/one_clip.py): share of all wave cycles | WAIT_ANY / WAIT_INST_ANY / ACTIVE_INST_ANY as parts of WAVE_CYCLES | "
            "instructions per MFMA: VALU, SALU, branch, VMEM, LDS | instruction fetches per 1000 instructions | I-cache miss rate\n")
    tot = sum(v["SQ_WAVE_CYCLES"] for v in val.values())
    for k, v in sorted(val.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:45]:
        wc = v["SQ_WAVE_CYCLES"] or 1; m = v["SQ_INSTS_MFMA"]
        ins = v["SQ_INSTS_VALU"] + v["SQ_INSTS_SALU"] + v["SQ_INSTS_VMEM"] + v["SQ_INSTS_LDS"] + v["SQ_INSTS_BRANCH"]
        per = (lambda x: f"{x / m:6.2f}") if m else (lambda x: f"{x / 1e6:6.1f}M")
        o.write(f"{k:70s} n {n[k]:5d} share {wc / tot:6.1%} | wait {v['SQ_WAIT_ANY'] / wc:5.1%} stall {v['SQ_WAIT_INST_ANY'] / wc:5.1%} active {v['SQ_ACTIVE_INST_ANY'] / wc:5.1%} | "
                f"valu {per(v['SQ_INSTS_VALU'])} salu {per(v['SQ_INSTS_SALU'])} br {per(v['SQ_INSTS_BRANCH'])} vmem {per(v['SQ_INSTS_VMEM'])} lds {per(v['SQ_INSTS_LDS'])} | "
                f"ifetch/kinst {1000 * v['SQ_IFETCH'] / max(ins, 1):6.1f} | icache miss {v['SQC_ICACHE_MISSES'] / max(v['SQC_ICACHE_REQ'], 1):6.2%} ({v['SQC_ICACHE_MISSES'] / max(n[k], 1):9.0f} per dispatch)\n")
print(open("gpurun_out/pmc_sq_mix.txt").read())
PY
