cd $GRAFT_REPO_ROOT; O=gpurun_out/r05x; mkdir -p $O
run() {
  if [ -n "$3" ]; then export IAF_HIP_LIB=$GRAFT_REPO_ROOT/$3; else unset IAF_HIP_LIB; fi
  python $2/bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('%-10s %.4f ms/step  %s' % ('$1', d['ms_per_step'], ['%.2f' % x.get('avg_launch_us', x.get('us', 0)) for x in k]))"
}
for rep in 1 2 3; do
  run prio0 . iaf_amd/_lib/libiaf_hip.so
  run prio1 . iaf_amd/_lib_prio1/libiaf_hip.so
  run prio2 . iaf_amd/_lib_prio2/libiaf_hip.so
done 2>&1 | tee $O/ab_helper_priority_same_box.txt
