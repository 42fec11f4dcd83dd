cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_fused_step.py tests/test_hip_halo_exchange.py -x -q 2>&1 | tail -3
run() {
  if [ -n "$3" ]; then export IAF_HIP_LIB=$GRAFT_REPO_ROOT/$3; else unset IAF_HIP_LIB; fi
  python $2/bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('%-10s %.4f ms/step  %s' % ('$1', d['ms_per_step'], ['%.2f' % x.get('avg_launch_us', x.get('us', 0)) for x in k]))"
}
for rep in 1 2 3; do
  run r04 _r04 ""
  run no_hl0 . iaf_amd/_lib_nohl0/libiaf_hip.so
  run hl0 . iaf_amd/_lib/libiaf_hip.so
done 2>&1 | tee $O/ab_helpers_take_first_layer_leftover_same_box.txt
for L in iaf_amd/_lib_nohl0/libiaf_hip.so iaf_amd/_lib/libiaf_hip.so; do for hw in 16 8; do IAF_HIP_LIB=$GRAFT_REPO_ROOT/$L python tools/fused_stamps.py --hw $hw 2>&1 | grep "per-WG total\|of which\|first epilogue"; done; done
