#!/bin/bash
# Matrix-pipe utilisation of the whole clip from rocprofv3 PMC counters (north_star: "evidenced by rocprof ... MFMA utilisation against gfx950 peak"):
# SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE in one pass (SQ + GRBM blocks; counters only, no trace domains), every dispatch of
# tools/one_clip.py N.  usage (through gpurun): bash tools/pmc_mfma_util.sh [denoise_steps] -> gpurun_out/pmc_mfma_util.txt
# MfmaUtil of a kernel = MFMA_BUSY / (GUI_ACTIVE x 128): the normalisation that gives 44 % for the 512-channel VAE conv whose s_memtime
# trace shows 2176 of 4350 cycles per K step in MFMAs (profiles/r01_gemm_sq_counters.txt, r01_gemm_kstep_trace.txt).
set -e
STEPS=${1:-5}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_mfma
timeout 1500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma -o pmc -- python tools/one_clip.py $STEPS > gpurun_out/pmc_mfma.log 2>&1 || true
python - <<'PY'
import csv, glob, collections, re
busy, act, n = collections.Counter(), collections.Counter(), collections.Counter()
for f in glob.glob("gpurun_out/pmc_mfma/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        fam = re.sub(r"<.*", "", k.replace("void ", "")).split("(")[0]
        v = float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES": busy[fam] += v; n[fam] += 1
        elif r["Counter_Name"] == "GRBM_GUI_ACTIVE": act[fam] += v
tot_b, tot_a = sum(busy.values()), sum(act.values())
with open("gpurun_out/pmc_mfma_util.txt", "w") as o:
    o.write("matrix-pipe utilisation per kernel family, one 25x384x512 clip (tools/one_clip.py), rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE\n")
    o.write("MfmaUtil = MFMA_BUSY / (GUI_ACTIVE x 128); share = the family's part of all GUI_ACTIVE cycles (profiled passes run at lower clocks: ratios, not times)\n")
    o.write(f"{'kernel family':40s} {'dispatches':>10s} {'share':>7s} {'MfmaUtil':>9s}\n")
    for fam, a in sorted(act.items(), key=lambda kv: -kv[1]):
        if a <= 0: continue
        o.write(f"{fam:40s} {n[fam]:10d} {a / tot_a:7.1%} {busy[fam] / (a * 128):9.1%}\n")
    o.write(f"{'ALL KERNELS':40s} {sum(n.values()):10d} {1:7.1%} {tot_b / (tot_a * 128):9.1%}\n")
    gem = [f for f in act if f.startswith("gemm_") or f.startswith("ff_fused") or f.startswith("conv_halo")]
    o.write(f"{'GEMM family (gemm_* + ff_fused)':40s} {sum(n[f] for f in gem):10d} {sum(act[f] for f in gem) / tot_a:7.1%} {sum(busy[f] for f in gem) / (sum(act[f] for f in gem) * 128):9.1%}\n")
print(open("gpurun_out/pmc_mfma_util.txt").read())
PY
rm -rf gpurun_out/pmc_mfma
