R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python -m pytest $R/tests/test_hip_halo_exchange.py -x -q 2>&1 | tail -5 > $O/t3.txt
for k in 0 4; do timeout 120 python $R/tools/fused_stamps.py --hw 16 --knob $k; done 2>&1 | grep -v amdgpu.ids > $O/stamps3.txt
tail -5 $O/t3.txt; cat $O/stamps3.txt
