# round 3, GPU call 7: data gradients on the bf16 matrix cores (transposed bf16x3 packs) -- parity + training benches
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 python -m pytest $R/tests/test_hip_parity.py $R/tests/test_hip_layer.py -q -m gpu -x -k "gradient or backward or trains or deferred or autotune" -s 2>&1 | grep -v "^$" | tail -30 > $O/pytest_dgrad.txt
timeout 500 python -m pytest $R/tests -q -m gpu 2>&1 | tail -8 > $O/pytest_gpu_run7.txt
IAF_BENCH_FORCE_DIST=1 python $R/bench.py --train --steps 50 > $O/bench_train_n1.json 2> /dev/null
IAF_BENCH_FORCE_DIST=1 python $R/bench.py --train --layers --steps 20 --warmup 5 > $O/bench_train_layers_n1.json 2> $O/bench_train_layers_n1.err
python $R/bench.py --train --layers --steps 20 --warmup 5 --precision f32 > $O/bench_train_layers_f32_n1.json 2> /dev/null
python $R/tools/layer_train_bench.py > $O/layer_train_bench.txt 2>&1
tail -12 $O/pytest_dgrad.txt; tail -4 $O/pytest_gpu_run7.txt; for f in bench_train_n1 bench_train_layers_n1 bench_train_layers_f32_n1; do python -c "
import json; d=json.load(open('$O/$f.json')); print('$f', d['ms_per_step'], d['value'], d['config'].get('exchange', {}).get('via'))"; done; cat $O/layer_train_bench.txt | tail -6
