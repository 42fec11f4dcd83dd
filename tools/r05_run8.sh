cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_hip_halo_exchange.py tests/test_hip_fused_step.py tests/test_hip_baseline_configs.py tests/test_hip_parity.py -x -q 2>&1 | tail -4
bash tools/r05_ab.sh hout iaf_amd/_lib_nohout/libiaf_hip.so iaf_amd/_lib/libiaf_hip.so
