cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
for rep in 1 2; do
for T in _r04 _c1 .; do
  echo "== tree $T"
  (cd $T && python tools/fused_stamps.py --hw 16 2>&1 | grep -v amdgpu.ids | cut -c1-230)
  python $T/bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('%-10s %.4f ms/step  %s' % ('$T', d['ms_per_step'], ['%.2f' % x.get('avg_launch_us', x.get('us', 0)) for x in k]))"
done; done 2>&1 | tee $O/ab_stamps_r04_c1_head_same_box.txt
