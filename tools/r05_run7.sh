cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
true
for rep in 1 2; do
  for h in 0 1; do
    true
    IAF_STEP_HELPERS=$h python bench.py --no-cpu-baseline > $O/bench_h8_${h}_${rep}.json 2>/dev/null; python tools/show_bench.py $O/bench_h8_${h}_${rep}.json | grep "ms/step\|8x8 IAF" | sed "s/^/helpers=$h /"
  done
done
