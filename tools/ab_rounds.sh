# same-box A/B of round 4's tree (a git worktree of its last commit under _r04/, built there) against the current tree:
#   git worktree add -f _r04 e7090da && (cd _r04 && python -m iaf_amd.build)
#   gpurun --timeout 900 -- 'bash tools/ab_rounds.sh'
# bench.py of each tree (the headline line: same workload, same timing), alternating, then one rocprofv3 kernel-trace pass each.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for T in _r04 .; do
  python $R/$T/bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; x=r['extended_unit']
print('rep $rep tree %-5s %.4f ms/step  %6.0f samples/s   posterior block %.1f / %.1f us' % ('$T', d['ms_per_step'], d['value'], x[0]['us'], x[1]['us']))"
done; done
for T in _r04 .; do
  rm -rf /tmp/pb_$T; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_$T -o bench -- python $R/$T/bench.py --no-cpu-baseline > /dev/null 2>&1
  echo "rocprofv3 kernel averages, tree $T:"; python - /tmp/pb_$T <<'PY'
import csv, glob, sys
for r in list(csv.DictReader(open(glob.glob(sys.argv[1] + '/*kernel_stats.csv')[0])))[:4]:
    print('    %-70s calls %6s  avg %8.2f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
