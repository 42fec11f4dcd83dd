# Refresh the measured evidence under gpurun_out/$ROUND/ (copied to profiles/$ROUND/ afterwards):
#   gpurun --timeout 1700 -- 'ROUND=r06 timeout 1650 bash tools/refresh_profiles.sh'
# Every rocprofv3 pass runs under its own `timeout`: counter collection serialises the launches, and a PMC pass over the
# hipGraph-replayed bench.py did not finish in 400 s (round 2) -- PMC passes go over tools/run_step.py (eager, a few launches).
set -x
R=$GRAFT_REPO_ROOT
RD=${ROUND:-r06}
O=$R/gpurun_out/$RD
mkdir -p $O/pmc
cd /tmp && export TMPDIR=/tmp
python -m pytest $R/tests -q -m gpu 2>&1 | tail -4 > $O/pytest_gpu.txt
python $R/bench.py --precision f32 > $O/bench_f32_n1.json 2> /dev/null
python $R/bench.py --precision bf16x3 > $O/bench_bf16x3_n1.json 2> /dev/null
python $R/bench.py --no-fuse-step > $O/bench_layer_by_layer_n1.json 2> /dev/null
python $R/bench.py --layers > $O/bench_layers_n1.json 2> /dev/null
python $R/bench.py --layers --model > $O/bench_model_n1.json 2> /dev/null
python $R/bench.py --iw-eval --steps 100 > $O/bench_iw_eval_n1.json 2> /dev/null
python $R/bench.py --iw-eval --model --steps 50 --warmup 5 > $O/bench_iw_eval_model_n1.json 2> /dev/null
python $R/bench.py --train --steps 50 > $O/bench_train_n1.json 2> /dev/null
python $R/bench.py --train --layers --steps 20 --warmup 5 > $O/bench_train_layers_n1.json 2> /dev/null
python $R/bench.py --train --model --steps 20 --warmup 5 > $O/bench_train_model_n1.json 2> /dev/null
# kernel trace of the SAME command as the headline bench line
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o bench -- python $R/bench.py --no-cpu-baseline > $O/bench_rocprof_line.json 2>/dev/null
cp /tmp/pb/*kernel_stats.csv $O/bench_kernel_stats.csv
# ... whose average the headline line repeats (roofline.rocprof): regenerate the JSON it reads BEFORE that line is measured
python $R/tools/make_profile_json.py $O $RD > /dev/null 2>&1; cp $O/rocprof_dominant_kernel.json $R/profiles/rocprof_dominant_kernel.json
python $R/bench.py > $O/bench_n1.json 2> /dev/null
python - <<'PY'
import csv, collections, glob, os
f = glob.glob('/tmp/pb/*kernel_trace.csv')[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    acc[(r['Kernel_Name'], r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '')))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
out = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/' + os.environ.get('ROUND', 'r06') + '/bench_kernel_trace_by_grid.csv'
with open(out, 'w') as o:
    o.write('kernel,grid_x,wg_x,calls,avg_ns,total_ns\n')
    for (k, g, w), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        o.write('"%s",%s,%s,%d,%.1f,%d\n' % (k, g, w, len(v), sum(v) / len(v), sum(v)))
PY
# PMC passes (one counter set per pass) over the one-launch step (the layer-by-layer kernels' counters: profiles/r02/pmc/)
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  n=$(echo $c | tr ' ' '_')
  timeout 90 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcs_$n -o pmc -- python $R/tools/run_step.py --hw 16 --reps 10 --precision ${PMC_PREC:-f16x2} > /dev/null 2>&1
  cp /tmp/pmcs_$n/*counter_collection.csv $O/pmc/step_${n}_counter_collection.csv
done
python $R/tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1
python $R/tools/make_profile_json.py $O $RD > $O/make_profile_json.txt 2>&1
python $R/tools/bench_configs.py > $O/bench_configs.md 2>/dev/null
for hw in 16 8; do python $R/tools/fused_stamps.py --hw $hw; done 2>&1 | grep -v amdgpu.ids > $O/fused_step_stamps.txt
(IAF_STEP_HELPERS=0 python $R/tools/fused_stamps.py --hw 8; IAF_FUSE_PAIR=1 python $R/tools/fused_stamps.py --hw 8) 2>&1 | grep -v amdgpu.ids > $O/fused_step_stamps_8x8_other_forms.txt
python $R/tools/layer_bench.py > $O/layer_bench.txt 2>&1
python $R/tools/layer_train_bench.py > $O/layer_train_bench.txt 2>&1
# which kernels a training step of one whole IAFLayer runs (data gradients on the bf16 matrix cores: EPI = 2 instantiations)
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl -o lt -- python $R/tools/layer_train_bench.py > /dev/null 2>&1
cp /tmp/pl/*kernel_stats.csv $O/layer_train_kernel_stats.csv
(python $R/tools/soak.py --iters 20000 --fresh 1000) > $O/soak_prod.txt 2>&1
python $R/tools/ds_layer_time.py 2>&1 | grep -v amdgpu.ids > $O/ds_layer_time.txt
python $R/tools/inverse_bench.py 2>&1 | grep -v amdgpu.ids > $O/inverse_bench.txt
(for d in 3 0; do IAF_PREP_DBG=$d python $R/tools/prep_time.py f16; done) 2>&1 | grep -v amdgpu.ids > $O/prep_time.txt
python $R/tools/conv_stamps.py 2>&1 | grep -v amdgpu.ids > $O/plain_conv_stamps.txt
IAF_XCH_DEBUG=3 python -m pytest $R/tests -q -m gpu 2>&1 | tail -4 > $O/pytest_gpu_scrambled.txt
# one steady-state training step of the 20-layer model, kernel by kernel
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/ptl -o tl -- python $R/bench.py --train --layers --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/step_breakdown.py $(ls /tmp/ptl/*/*kernel_trace.csv /tmp/ptl/*kernel_trace.csv 2>/dev/null | head -1) --json $O/train_kernels_layers.json > $O/train_layers_step_breakdown.txt 2>&1
# ... of the whole model's training step (CVAE1.forward_backward + Adamax), and of its forward (marker: the image scaling launch)
timeout 250 rocprofv3 --kernel-trace --output-format csv -d /tmp/ptm -o tm -- python $R/bench.py --train --model --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/step_breakdown.py $(ls /tmp/ptm/*/*kernel_trace.csv /tmp/ptm/*kernel_trace.csv 2>/dev/null | head -1) --json $O/train_kernels_model.json > $O/train_model_step_breakdown.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pmf -o mf -- python $R/bench.py --layers --model --steps 6 --warmup 2 --settle-seconds 0 --repeats 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/step_breakdown.py $(ls /tmp/pmf/*/*kernel_trace.csv /tmp/pmf/*kernel_trace.csv 2>/dev/null | head -1) iaf_image_to_float_kernel > $O/model_forward_step_breakdown.txt 2>&1
tail -3 $O/pytest_gpu.txt; python $R/tools/show_bench.py $O/bench_n1.json; cat $O/make_profile_json.txt; cat $O/fused_step_stamps.txt; cat $O/layer_train_bench.txt | tail -3; head -12 $O/layer_train_kernel_stats.csv | cut -c1-160; tail -2 $O/soak_prod.txt
