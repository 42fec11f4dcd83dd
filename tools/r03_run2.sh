# round 3, GPU call 2: deeper weight rings + output pair without K split (iaf_step_fused.hpp), RCCL behind the C ABI
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 500 python -m pytest $R/tests -q -m gpu 2>&1 | tail -25 > $O/pytest_gpu_run2.txt
for hw in 16 8; do python $R/tools/fused_stamps.py --hw $hw; done > $O/fused_step_stamps_run2.txt 2>&1
IAF_FUSE_STEP_R=2 python $R/tools/fused_stamps.py --hw 8 > $O/fused_step_stamps_8x8_R2.txt 2>&1
python $R/bench.py > $O/bench_n1_run2.json 2> $O/bench_n1_run2.err
IAF_BENCH_FORCE_DIST=1 python $R/bench.py --train --steps 50 > $O/bench_train_n1.json 2> $O/bench_train_n1.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o bench -- python $R/bench.py --no-cpu-baseline > $O/bench_rocprof_line_run2.json 2>/dev/null
cp /tmp/pb/*kernel_stats.csv $O/bench_kernel_stats_run2.csv
python $R/tools/show_bench.py $O/bench_n1_run2.json
tail -4 $O/pytest_gpu_run2.txt; cat $O/fused_step_stamps_run2.txt $O/fused_step_stamps_8x8_R2.txt | grep -v amdgpu.ids; head -8 $O/bench_kernel_stats_run2.csv | cut -c1-200; tail -3 $O/bench_train_n1.err; cut -c1-600 $O/bench_train_n1.json
