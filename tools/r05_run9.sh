cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
UG_BENCH_GEGLU=1 timeout 900 python tools/ab_gemm.py "35,20,21,15,64,0" > gpurun_out/r05_ab_geglu_2wg.txt 2>&1
timeout 600 python tools/ab_gemm.py "-1,20,21" > gpurun_out/r05_ab_plain_2wg.txt 2>&1
grep -v "^vae\|^unet\|^tconv" gpurun_out/r05_ab_geglu_2wg.txt; grep -v "^vae\|^unet\|^tconv" gpurun_out/r05_ab_plain_2wg.txt
