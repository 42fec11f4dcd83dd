cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_hip_halo_exchange.py tests/test_hip_fused_step.py tests/test_hip_baseline_configs.py -x -q 2>&1 | tail -4
bash tools/r05_ab.sh ilv iaf_amd/_lib_base/libiaf_hip.so iaf_amd/_lib/libiaf_hip.so iaf_amd/_lib_rdo5/libiaf_hip.so
