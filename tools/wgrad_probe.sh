cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
for b in $PROBES; do echo "=== $b"; for a in "160 224 9" "160 448 9" "160 384 9" "160 160 5" "160 64 5" "32 160 5"; do timeout 60 tools/probe/bin/$b $a | sed 's/per K block, mean over workgroups: //; s/           worst .*outputs: /   err /' | paste - - ; done; done 2>&1 | tee gpurun_out/r03/wgrad_probe.txt
