R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 1500 python -m pytest $R/tests -q -m gpu -x 2>&1 | tail -4 > $O/pytest_gpu_d.txt
tail -4 $O/pytest_gpu_d.txt
