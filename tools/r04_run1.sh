set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python -m pytest $R/tests/test_hip_halo_exchange.py $R/tests/test_hip_fused_step.py -x -q 2>&1 | tail -25 > $O/t1.txt
timeout 300 python $R/tools/xch_ab.py > $O/xch_ab.txt 2>&1
for k in 0 4 -1; do timeout 120 python $R/tools/fused_stamps.py --hw 16 --knob $k; done 2>&1 | grep -v amdgpu.ids > $O/stamps1.txt
timeout 300 python $R/bench.py --no-cpu-baseline > $O/bench_a.json 2> $O/bench_a.err
tail -25 $O/t1.txt; cat $O/xch_ab.txt; cat $O/stamps1.txt; python $R/tools/show_bench.py $O/bench_a.json; tail -5 $O/bench_a.err
