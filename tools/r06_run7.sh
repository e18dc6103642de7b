cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# A/B: the build before a change (unigeo_amd/csrc/build/base/libunigeo_base.so) against the in-tree build, consecutive processes on one box
BASE=$GRAFT_REPO_ROOT/unigeo_amd/csrc/build/base/libunigeo_base.so
{
for i in 1 2; do
echo "base  : $(UG_LIB_PATH=$BASE timeout 300 python tools/ab_lib.py 2>&1 | tail -1)"
echo "tree  : $(timeout 300 python tools/ab_lib.py 2>&1 | tail -1)"
done
} > gpurun_out/r06_ab_${1:-x}.txt 2>&1
cat gpurun_out/r06_ab_${1:-x}.txt
