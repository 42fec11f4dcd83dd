R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python -m pytest $R/tests/test_hip_halo_exchange.py $R/tests/test_hip_fused_step.py -x -q 2>&1 | tail -5 > $O/t7.txt
timeout 300 python $R/tools/xch_ab.py --rounds 2 --knobs=0,1,-1,0s > $O/xch_ab7.txt 2>&1
for k in 0; do timeout 120 python $R/tools/fused_stamps.py --hw 16 --knob $k; done 2>&1 | grep -v amdgpu.ids > $O/stamps7.txt
timeout 300 python $R/bench.py --no-cpu-baseline > $O/bench_g.json 2> $O/bench_g.err
tail -5 $O/t7.txt; cat $O/xch_ab7.txt; cat $O/stamps7.txt; python $R/tools/show_bench.py $O/bench_g.json | head -1
