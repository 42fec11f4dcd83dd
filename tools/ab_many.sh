# same-box A/B of several builds of the engine (IAF_BUILD_TAG builds in iaf_amd/_lib_<tag>/), alternating, bench.py headline + in-situ kernel times:
#   gpurun --timeout 900 -- 'bash tools/ab_many.sh out_name "" rd344 rd455'        ("" = the product library)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
mkdir -p $(dirname $OUT)
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for tag in "$@"; do
  L=$R/iaf_amd/_lib${tag:+_$tag}/libiaf_hip.so
  IAF_HIP_LIB=$L python $R/bench.py --no-cpu-baseline $AB_BENCH_ARGS > /tmp/ab_$$.json 2>/dev/null
  python $R/tools/show_bench.py /tmp/ab_$$.json | grep -v "^{" | sed "s/^/[${tag:-product} $rep] /"
done; done > $OUT.txt 2>&1
cat $OUT.txt
