# round 3, GPU call 1: parity of the round's host / prep / KL changes + A/B of the weight-prep variants
#   gpurun --timeout 1200 -- 'timeout 1150 bash tools/r03_run1.sh'
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
U=$R/iaf_amd/_lib_prepunits/libiaf_hip.so
timeout 500 python -m pytest $R/tests -q -m gpu 2>&1 | tail -25 > $O/pytest_gpu.txt
timeout 200 python -m pytest $R/tests/test_hip_dynamic_range.py -q -m gpu -s 2>&1 | grep -v "^$" | tail -40 > $O/pytest_dynamic_range.txt
IAF_HIP_LIB=$U timeout 500 python -m pytest $R/tests -q -m gpu 2>&1 | tail -25 > $O/pytest_gpu_prepunits.txt
python $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python $R/bench.py --keep-f32-pack --no-cpu-baseline > $O/bench_keepf32.json 2> /dev/null
IAF_HIP_LIB=$U python $R/bench.py --no-cpu-baseline > $O/bench_units.json 2> /dev/null
IAF_HIP_LIB=$U python $R/bench.py --no-cpu-baseline --keep-f32-pack > $O/bench_units_keepf32.json 2> /dev/null
python $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> /dev/null
for hw in 16 8; do python $R/tools/fused_stamps.py --hw $hw; done > $O/fused_step_stamps.txt 2>&1
for f in bench_n1 bench_keepf32 bench_units bench_units_keepf32 bench_driver_flags; do echo $f; python $R/tools/show_bench.py $O/$f.json; done
tail -5 $O/pytest_gpu.txt; tail -5 $O/pytest_gpu_prepunits.txt; tail -3 $O/fused_step_stamps.txt
