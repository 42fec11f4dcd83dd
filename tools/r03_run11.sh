# round 3, GPU call 11: tap-row weight gradient on the bf16 matrix cores -- parity, then A/B on one box
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 500 python -m pytest $R/tests/test_hip_parity.py $R/tests/test_hip_layer.py -q -m gpu -x -k "gradient or backward or trains or deferred or autotune" 2>&1 | tail -60 > $O/pytest_wgrad.txt
tail -8 $O/pytest_wgrad.txt
for v in 0 1 0 1; do echo "IAF_WGRAD_BF3=$v"; IAF_WGRAD_BF3=$v python $R/tools/layer_train_bench.py 2>&1 | grep -v amdgpu.ids; done > $O/ab_wgrad.txt 2>&1
for v in 0 1; do IAF_WGRAD_BF3=$v python $R/bench.py --train --layers --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_train_layers_wgrad$v.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_train_layers_wgrad$v.json')); print('train-layers IAF_WGRAD_BF3=$v', d['ms_per_step'])" >> $O/ab_wgrad.txt; done
cat $O/ab_wgrad.txt
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl -o lt -- python $R/tools/layer_train_bench.py > /dev/null 2>&1
cp /tmp/pl/*kernel_stats.csv $O/layer_train_kernel_stats_wgrad.csv; head -14 $O/layer_train_kernel_stats_wgrad.csv | cut -c1-150
cd $R; for a in "160 160 5" "160 64 5" "32 160 5" "192 160 9" "160 160 9"; do timeout 60 tools/probe/bin/wgrad_probe_ns0 $a | sed 's/per K block, mean over workgroups: .*//; s/           worst .*outputs: /   err /' | paste - - ; done 2>&1 | tee $O/wgrad_probe_small.txt
