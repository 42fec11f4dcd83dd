# bench.py headline on one box: trees at several commits (each with its own bench.py and library), alternating
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
run() {
  unset IAF_HIP_LIB
  python $2/bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('%-10s %.4f ms/step  %s' % ('$1', d['ms_per_step'], ['%.2f' % x.get('avg_launch_us', x.get('us', 0)) for x in k]))"
}
for rep in 1 2 3; do
  run r04_e7090da _r04
  run c1_c09e1b6 _c1
  run c2_6885a5e _c2
  run head .
done 2>&1 | tee $O/ab_bisect_same_box.txt
