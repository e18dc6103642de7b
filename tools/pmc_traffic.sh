#!/bin/bash
# HBM traffic of the GEMM kernels via rocprofv3 PMC counters, FETCH_SIZE and WRITE_SIZE in SEPARATE passes
# (they do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots"); counters only, no trace domains.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc_$C -o pmc -- python tools/one_clip.py > gpurun_out/pmc_$C.log 2>&1 || true
done
python tools/pmc_summarise.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
