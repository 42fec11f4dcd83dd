R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python -m pytest $R/tests/test_hip_fused_step.py $R/tests/test_hip_baseline_configs.py -x -q 2>&1 | tail -3 > $O/t10.txt
timeout 300 python $R/tools/xch_ab.py --rounds 2 --hw 8 --knobs=-1,-1s > $O/xch_ab10.txt 2>&1
timeout 120 python $R/tools/fused_stamps.py --hw 8 2>&1 | grep -v amdgpu.ids > $O/stamps10.txt
timeout 300 python $R/bench.py --no-cpu-baseline > $O/bench_l.json 2> $O/bench_l.err
tail -3 $O/t10.txt; cat $O/xch_ab10.txt; cat $O/stamps10.txt; python $R/tools/show_bench.py $O/bench_l.json
