cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# A/B/... on one box, consecutive processes, twice round: libraries under unigeo_amd/csrc/build/base named on the command line (after the tag), then the in-tree build
B=$GRAFT_REPO_ROOT/unigeo_amd/csrc/build/base
tag=$1; shift
{
for i in 1 2; do
for l in "$@"; do echo "$l : $(UG_LIB_PATH=$B/$l timeout 300 python tools/ab_lib.py 2>&1 | tail -1)"; done
echo "tree  : $(timeout 300 python tools/ab_lib.py 2>&1 | tail -1)"
done
} > gpurun_out/r06_ab_$tag.txt 2>&1
cat gpurun_out/r06_ab_$tag.txt
