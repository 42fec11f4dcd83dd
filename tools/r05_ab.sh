# same-box A/B of engine builds (round 5):  bash tools/r05_ab.sh <tag> lib1 lib2 ...
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
TAG=$1; shift
for rep in 1 2; do for L in "$@"; do
  n=$(basename $(dirname $L))
  IAF_HIP_LIB=$GRAFT_REPO_ROOT/$L python tools/fused_stamps.py --hw 16 2>&1 | grep -v amdgpu.ids | grep "us per call\|per-WG total\|compute waves\|prologue\|of which" | sed "s/^/$n /"
  IAF_HIP_LIB=$GRAFT_REPO_ROOT/$L python bench.py --no-cpu-baseline > $O/bench_ab_${TAG}_${n}_$rep.json 2>/dev/null; python tools/show_bench.py $O/bench_ab_${TAG}_${n}_$rep.json | head -1 | sed "s/^/$n /"
done; done > $O/ab_${TAG}.txt 2>&1
cat $O/ab_${TAG}.txt
