cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_hip_halo_exchange.py tests/test_hip_fused_step.py tests/test_hip_baseline_configs.py tests/test_hip_parity.py -x -q 2>&1 | tail -4
bash tools/r05_ab.sh hleft iaf_amd/_lib_nohl/libiaf_hip.so iaf_amd/_lib/libiaf_hip.so
for L in iaf_amd/_lib_nohl/libiaf_hip.so iaf_amd/_lib/libiaf_hip.so; do python - <<PY
import json
d = json.load(open("gpurun_out/r05/bench_ab_hleft_%s_2.json" % "$L".split("/")[1]))
print("$L", json.dumps(d["roofline"].get("extended_unit"))[:600])
PY
done
