set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r01
cd /tmp && export TMPDIR=/tmp
python -m pytest $R/tests -q -m gpu 2>&1 | tail -4 > $R/gpurun_out/r01/pytest_gpu.txt
python $R/bench.py > $R/gpurun_out/r01/bench_n1.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o bench -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/r01/bench_rocprof_line.json 2>/dev/null
cp /tmp/pb/*kernel_stats.csv $R/gpurun_out/r01/bench_kernel_stats.csv
python - <<'PY'
import csv, collections, glob, os
f = glob.glob('/tmp/pb/*kernel_trace.csv')[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    acc[(r['Kernel_Name'], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size',''))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
out = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r01/bench_kernel_trace_by_grid.csv'
with open(out, 'w') as o:
    o.write('kernel,grid_x,calls,avg_ns,total_ns\n')
    for (k, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        o.write('"%s",%s,%d,%.1f,%d\n' % (k, g, len(v), sum(v) / len(v), sum(v)))
PY
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o pmc -- python $R/tools/run_step.py --hw 16 --reps 10 > /dev/null 2>&1
  cp /tmp/pmc_$c/*counter_collection.csv $R/gpurun_out/r01/${c}_counter_collection.csv
done
python $R/tools/bench_configs.py > $R/gpurun_out/r01/bench_configs.md 2>/dev/null
tail -3 $R/gpurun_out/r01/pytest_gpu.txt; cut -c1-250 $R/gpurun_out/r01/bench_n1.json
