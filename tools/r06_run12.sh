cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# A/B of an environment switch on the in-tree build, consecutive processes, twice round: bash tools/r06_run12.sh <tag> <VAR=value>
tag=$1; shift
{
for i in 1 2; do
echo "$1 : $(env $1 timeout 300 python tools/ab_lib.py 2>&1 | tail -1)"
echo "default : $(timeout 300 python tools/ab_lib.py 2>&1 | tail -1)"
done
} > gpurun_out/r06_ab_$tag.txt 2>&1
cat gpurun_out/r06_ab_$tag.txt
