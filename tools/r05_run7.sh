cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
timeout 400 python -m pytest tests/test_hip_halo_exchange.py tests/test_hip_fused_step.py tests/test_hip_baseline_configs.py tests/test_hip_parity.py tests/test_hip_pair_step.py -x -q 2>&1 | tail -4
for rep in 1 2; do
  for h in 0 1; do
    IAF_STEP_HELPERS=$h python tools/fused_stamps.py --hw 8 2>&1 | grep -v amdgpu.ids | grep "us per call\|per-WG total\|of which" | sed "s/^/helpers=$h /"
    IAF_STEP_HELPERS=$h python bench.py --no-cpu-baseline > $O/bench_h8_$h_$rep.json 2>/dev/null; python tools/show_bench.py $O/bench_h8_$h_$rep.json | grep "ms/step\|8x8 IAF" | sed "s/^/helpers=$h /"
  done
done
