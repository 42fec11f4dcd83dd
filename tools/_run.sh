R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 1500 python -m pytest $R/tests -q -m gpu -x 2>&1 | tail -4 > $O/pytest_gpu_b.txt
timeout 600 python $R/tools/bench_configs.py > $O/bench_configs_helpers.md 2>/dev/null
tail -4 $O/pytest_gpu_b.txt; cat $O/bench_configs_helpers.md
