cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
run() {
  if [ -n "$3" ]; then export IAF_HIP_LIB=$GRAFT_REPO_ROOT/$3; else unset IAF_HIP_LIB; fi
  python $2/bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('%-10s %.4f ms/step  %s' % ('$1', d['ms_per_step'], ['%.2f' % x.get('avg_launch_us', x.get('us', 0)) for x in k]))"
}
for rep in 1 2 3; do
  run r04 _r04 ""
  run head . iaf_amd/_lib/libiaf_hip.so
  run hout_by_tap . iaf_amd/_lib_hout2/libiaf_hip.so
done 2>&1 | tee $O/ab_hout_split_same_box.txt
for L in iaf_amd/_lib/libiaf_hip.so iaf_amd/_lib_hout2/libiaf_hip.so; do IAF_HIP_LIB=$GRAFT_REPO_ROOT/$L python tools/fused_stamps.py --hw 16 2>&1 | grep "per-WG total\|compute waves\|second epilogue"; done
