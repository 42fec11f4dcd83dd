# round 3, GPU call 14: halo exchange between the row blocks of the one-launch step (XCH) -- parity, then on/off on one box
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 python -m pytest $R/tests/test_hip_fused_step.py $R/tests/test_hip_baseline_configs.py -q -m gpu -x 2>&1 | tail -12 > $O/pytest_xch.txt
tail -6 $O/pytest_xch.txt
for rep in 1 2; do for x in 0 1 2; do
  for hw in 16 8; do IAF_FUSE_XCH=$x timeout 120 python $R/tools/fused_stamps.py --hw $hw 2>&1 | grep -v amdgpu.ids | sed "s/^/xch$x /"; done
  IAF_FUSE_XCH=$x timeout 300 python $R/bench.py --no-cpu-baseline > $O/bench_xch${x}_$rep.json 2>/dev/null; python $R/tools/show_bench.py $O/bench_xch${x}_$rep.json | sed "s/^/xch$x /"
done; done > $O/ab_xch.txt 2>&1
grep -v "^+" $O/ab_xch.txt | grep "iaf_step\|ms/step\|per-WG\|posterior"
