set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python -m pytest $R/tests/test_hip_halo_exchange.py -x -q 2>&1 | tail -5 > $O/t2.txt
timeout 300 python $R/tools/xch_ab.py --rounds 2 > $O/xch_ab2.txt 2>&1
for k in 0 4; do timeout 120 python $R/tools/fused_stamps.py --hw 16 --knob $k; done 2>&1 | grep -v amdgpu.ids > $O/stamps2.txt
timeout 300 python $R/bench.py --no-cpu-baseline > $O/bench_b.json 2> $O/bench_b.err
tail -5 $O/t2.txt; cat $O/xch_ab2.txt; cat $O/stamps2.txt; python $R/tools/show_bench.py $O/bench_b.json; tail -5 $O/bench_b.err
