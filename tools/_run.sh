R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl -o lt -- python $R/tools/layer_train_bench.py > /dev/null 2>&1
head -14 /tmp/pl/*kernel_stats.csv | cut -c1-150
