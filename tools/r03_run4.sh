# round 3, GPU call 4: two waves per SIMD in the one-launch step (IAF_FUSE_STEP_WV=8) against four, same box
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
IAF_FUSE_STEP_WV=8 timeout 400 python -m pytest $R/tests/test_hip_fused_step.py $R/tests/test_hip_baseline_configs.py $R/tests/test_hip_dynamic_range.py $R/tests/test_hip_parity.py -q -m gpu 2>&1 | tail -15 > $O/pytest_gpu_wv8.txt
for wv in 4 8 4 8; do for hw in 16 8; do IAF_FUSE_STEP_WV=$wv python $R/tools/fused_stamps.py --hw $hw 2>&1 | grep -v amdgpu.ids | sed "s/^/wv$wv /"; done; done > $O/fused_step_stamps_wv.txt 2>&1
for wv in 4 8 4 8; do IAF_FUSE_STEP_WV=$wv python $R/bench.py --no-cpu-baseline > $O/bench_wv${wv}.json 2>/dev/null; python $R/tools/show_bench.py $O/bench_wv${wv}.json | sed "s/^/wv$wv /"; done > $O/bench_wv.txt 2>&1
tail -5 $O/pytest_gpu_wv8.txt; cat $O/fused_step_stamps_wv.txt; cat $O/bench_wv.txt
