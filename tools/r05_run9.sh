cd $GRAFT_REPO_ROOT
timeout 500 python -m pytest tests/test_hip_halo_exchange.py tests/test_hip_pair_step.py tests/test_hip_fused_step.py tests/test_hip_baseline_configs.py -x -q 2>&1 | tail -4
bash tools/r05_ab4.sh
