# round 3, GPU call 12: full GPU suite with the tap-row bf16x3 weight gradient on
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 1500 python -m pytest $R/tests -q -m gpu 2>&1 | tail -25 > $O/pytest_gpu.txt
tail -5 $O/pytest_gpu.txt
