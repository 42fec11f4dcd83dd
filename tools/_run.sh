R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 1500 python -m pytest $R/tests -q -m gpu -x 2>&1 | tail -4 > $O/pytest_gpu_c.txt
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o bench -- python $R/bench.py --no-cpu-baseline > $O/bench_m.json 2>/dev/null
head -5 /tmp/pb/*kernel_stats.csv | cut -c1-140 > $O/kstats_m.txt
timeout 300 python $R/bench.py --no-cpu-baseline > $O/bench_m2.json 2>/dev/null
tail -4 $O/pytest_gpu_c.txt; cat $O/kstats_m.txt; python $R/tools/show_bench.py $O/bench_m2.json | head -1
