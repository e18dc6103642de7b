for s in l1sq l1tc ff2l1 l1qkv; do python tools/gemm_trace.py $s -1 2>&1 | tail -80; done
