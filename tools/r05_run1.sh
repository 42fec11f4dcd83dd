cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python -m pytest tests/test_hip_pair_step.py -x -q 2>&1 | tail -15 > gpurun_out/r05/t_pair.txt
python -m pytest tests/test_hip_fused_step.py tests/test_hip_halo_exchange.py tests/test_hip_baseline_configs.py -x -q 2>&1 | tail -8 > gpurun_out/r05/t_fused.txt
for hw in 8 16; do python tools/fused_stamps.py --hw $hw; done 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/stamps1.txt
IAF_FUSE_PAIR=0 python tools/fused_stamps.py --hw 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/stamps1_nopair.txt
python bench.py --no-cpu-baseline > gpurun_out/r05/bench1.json 2> gpurun_out/r05/bench1.err
IAF_FUSE_PAIR=0 python bench.py --no-cpu-baseline > gpurun_out/r05/bench1_nopair.json 2> /dev/null
cat gpurun_out/r05/t_pair.txt gpurun_out/r05/t_fused.txt gpurun_out/r05/stamps1.txt gpurun_out/r05/stamps1_nopair.txt
python tools/show_bench.py gpurun_out/r05/bench1.json; python tools/show_bench.py gpurun_out/r05/bench1_nopair.json
