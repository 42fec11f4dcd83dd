# round 3, GPU call 16: full suite with the deep stack's halo-exchange kernels, config table
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 1200 python -m pytest $R/tests -q -m gpu 2>&1 | tail -12 > $O/pytest_gpu.txt
tail -4 $O/pytest_gpu.txt
timeout 400 python $R/tools/bench_configs.py > $O/bench_configs.md 2>/dev/null; grep "config4" $O/bench_configs.md | cut -c1-140
