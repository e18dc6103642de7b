cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# A/B on one box, consecutive processes, twice round: $2 = library under unigeo_amd/csrc/build/base (the build before a change) against the in-tree build
B=$GRAFT_REPO_ROOT/unigeo_amd/csrc/build/base
{
for i in 1 2; do
echo "$2 : $(UG_LIB_PATH=$B/$2 timeout 300 python tools/ab_lib.py 2>&1 | tail -1)"
echo "tree  : $(timeout 300 python tools/ab_lib.py 2>&1 | tail -1)"
done
} > gpurun_out/r06_ab_${1:-x}.txt 2>&1
cat gpurun_out/r06_ab_${1:-x}.txt
