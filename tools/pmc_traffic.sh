#!/bin/bash
# HBM traffic of the GEMM kernels via rocprofv3 PMC counters, FETCH_SIZE and WRITE_SIZE in SEPARATE passes
# (they do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots"); counters only, no trace domains.
# usage (on the GPU box, through gpurun): bash tools/pmc_traffic.sh [denoise_steps | sn]   -> gpurun_out/pmc_traffic_gemm.json
# ("sn": the StableNormal workload, one 576 x 576 image per call -> gpurun_out/pmc_traffic_gemm_sn.json)
# (copy it to profiles/rNN_pmc_traffic_gemm.json; bench.py cites that file by sha256).  The same step mix is profiled with
# HIP events first, so the summary carries the algorithmic bytes per launch of exactly the launches that were counted.
set -e
STEPS=${1:-25}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "$STEPS" = "sn" ]; then WL="tools/one_image_sn.py"; ARG=""; else WL="tools/one_clip.py"; ARG="$STEPS"; fi
python $WL $ARG --events gpurun_out/pmc_events.json > gpurun_out/pmc_events.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$C
  timeout 1500 rocprofv3 --pmc $C --kernel-include-regex "gemm_(kernel|ldr_kernel|ws_kernel|stream320_kernel)|ff_fused_kernel|conv_halo_kernel" --output-format csv -d gpurun_out/pmc_$C -o pmc -- python $WL $ARG > gpurun_out/pmc_$C.log 2>&1 || true
done
python tools/pmc_summarise.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
if [ "$STEPS" = "sn" ]; then mv gpurun_out/pmc_traffic_gemm.json gpurun_out/pmc_traffic_gemm_sn.json; fi
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE   # tens of MB of per-dispatch rows; the summary carries their hashes
