# round 3, GPU call 13: dead centre-tap blocks of the triangular hidden layers skipped in the one-launch step -- parity, then A/B
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python -m pytest $R/tests -q -m gpu -x -k "not layer_backward and not trains" 2>&1 | tail -8 > $O/pytest_tri.txt
tail -4 $O/pytest_tri.txt
bash $R/tools/ab_libs.sh iaf_amd/_lib_base/libiaf_hip.so iaf_amd/_lib/libiaf_hip.so ab_tri 2>&1 | grep -v "^+" | grep "iaf_step\|ms/step\|per-WG\|second conv K loop" | tee $O/ab_tri_summary.txt
