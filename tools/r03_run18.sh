set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench_n1.json 2> /dev/null
python $R/tools/show_bench.py $O/bench_n1.json
