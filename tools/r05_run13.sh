cd $GRAFT_REPO_ROOT; O=gpurun_out/r05x; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_fused_step.py tests/test_hip_baseline_configs.py -x -q 2>&1 | tail -3
run() {
  if [ -n "$3" ]; then export IAF_HIP_LIB=$GRAFT_REPO_ROOT/$3; else unset IAF_HIP_LIB; fi
  python $2/bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']; x=d['roofline']['extended_unit']
print('%-10s %.4f ms/step  %s  posterior %.1f / %.1f' % ('$1', d['ms_per_step'], ['%.2f' % y.get('avg_launch_us', y.get('us', 0)) for y in k], x[0]['us'], x[1]['us']))"
}
for rep in 1 2 3; do
  run no_hout8 . iaf_amd/_lib_noho8/libiaf_hip.so
  run hout8 . iaf_amd/_lib/libiaf_hip.so
done 2>&1 | tee $O/ab_hout8_same_box.txt
IAF_HIP_LIB=$GRAFT_REPO_ROOT/iaf_amd/_lib/libiaf_hip.so python tools/fused_stamps.py --hw 8 2>&1 | grep "per-WG total\|second epilogue\|output conv"
