# same-box A/B of two builds of the engine (box-to-box spread is +-5 %: only alternating runs on ONE box compare kernels):
#   gpurun --timeout 600 -- 'timeout 550 bash tools/ab_libs.sh iaf_amd/_lib_base/libiaf_hip.so iaf_amd/_lib/libiaf_hip.so'
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A=$R/$1; B=$R/$2; TAG=${3:-ab}
for rep in 1 2; do for L in $A $B; do
  n=$(basename $(dirname $L))
  for hw in 16 8; do IAF_HIP_LIB=$L python $R/tools/fused_stamps.py --hw $hw 2>&1 | grep -v amdgpu.ids | sed "s/^/$n /"; done
  IAF_HIP_LIB=$L python $R/bench.py --no-cpu-baseline > $O/bench_${TAG}_${n}_$rep.json 2>/dev/null; python $R/tools/show_bench.py $O/bench_${TAG}_${n}_$rep.json | sed "s/^/$n /"
done; done > $O/${TAG}.txt 2>&1
grep -v "^+" $O/${TAG}.txt
