cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# A/B/C on one box, consecutive processes, twice round: build with the fast SiLU (base) / + epilogue after the barrier (move) / + lean epilogue (in-tree)
B=$GRAFT_REPO_ROOT/unigeo_amd/csrc/build/base
{
for i in 1 2; do
echo "silu  : $(UG_LIB_PATH=$B/libunigeo_base.so timeout 300 python tools/ab_lib.py 2>&1 | tail -1)"
echo "move  : $(UG_LIB_PATH=$B/libunigeo_move.so timeout 300 python tools/ab_lib.py 2>&1 | tail -1)"
echo "tree  : $(timeout 300 python tools/ab_lib.py 2>&1 | tail -1)"
done
} > gpurun_out/r06_ab_${1:-x}.txt 2>&1
cat gpurun_out/r06_ab_${1:-x}.txt
