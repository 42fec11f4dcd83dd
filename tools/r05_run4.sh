cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 400 python -m pytest tests/test_hip_halo_exchange.py tests/test_hip_pair_step.py tests/test_hip_fused_step.py -x -q 2>&1 | tail -8 > $O/t_lists.txt
python tools/fused_stamps.py --hw 16 2>&1 | grep -v amdgpu.ids > $O/stamps4_16.txt
IAF_FUSE_PAIR=1 python tools/fused_stamps.py --hw 8 2>&1 | grep -v amdgpu.ids > $O/stamps4_pair.txt
python bench.py --no-cpu-baseline > $O/bench4.json 2> /dev/null
cat $O/t_lists.txt $O/stamps4_16.txt; head -12 $O/stamps4_pair.txt; python tools/show_bench.py $O/bench4.json
