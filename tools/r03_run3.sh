# round 3, GPU call 3: unconditional context loads (prologue), product-group accumulators, final operands prefetched early,
# KL finish with 16-byte loads, capture-time descriptor upload
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 500 python -m pytest $R/tests -q -m gpu -x 2>&1 | tail -25 > $O/pytest_gpu_run3.txt
for hw in 16 8; do python $R/tools/fused_stamps.py --hw $hw; done > $O/fused_step_stamps_run3.txt 2>&1
python $R/bench.py > $O/bench_n1_run3.json 2> $O/bench_n1_run3.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o bench -- python $R/bench.py --no-cpu-baseline > $O/bench_rocprof_line_run3.json 2>/dev/null
cp /tmp/pb/*kernel_stats.csv $O/bench_kernel_stats_run3.csv
python $R/tools/show_bench.py $O/bench_n1_run3.json
tail -4 $O/pytest_gpu_run3.txt; grep -v amdgpu.ids $O/fused_step_stamps_run3.txt; head -8 $O/bench_kernel_stats_run3.csv | cut -c1-200
