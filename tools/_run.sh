R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py > $O/bench_i.json 2> $O/bench_i.err
python $R/tools/show_bench.py $O/bench_i.json; tail -3 $O/bench_i.err; python -c "
import json; d=json.load(open('$O/bench_i.json')); r=d['roofline']; print({k:r[k] for k in ('achieved','peak','frac','frac_of_f32_mfma_peak','avg_launch_us','avg_launch_us_relaunched_hot')}); print(r['step']); print(d['cpu_baseline'])"
