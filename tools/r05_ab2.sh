# bench.py headline on one box: round 4's tree and builds of the current tree (IAF_HIP_LIB), alternating
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
run() { # name, dir of bench.py, lib or ""
  if [ -n "$3" ]; then export IAF_HIP_LIB=$GRAFT_REPO_ROOT/$3; else unset IAF_HIP_LIB; fi
  python $2/bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('%-10s %.4f ms/step  %s' % ('$1', d['ms_per_step'], ['%.2f' % x.get('avg_launch_us', x.get('us', 0)) for x in k]))"
}
for rep in 1 2 3; do
  run r04 _r04 ""
  run v0 . iaf_amd/_lib_v0/libiaf_hip.so
  run v1_hleft . iaf_amd/_lib_v1/libiaf_hip.so
  run v2_hout . iaf_amd/_lib_v2/libiaf_hip.so
  run head . iaf_amd/_lib/libiaf_hip.so
done 2>&1 | tee $O/ab_variants_vs_round4_same_box.txt
