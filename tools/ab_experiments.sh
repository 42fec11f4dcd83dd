# Build the staged experiment libraries next to the product one (CPU, a few minutes each) and print the GPU command that
# A/Bs them in one short gpurun call.   bash tools/ab_experiments.sh [build]
#   prepscalar  -DIAF_EXP_PREP_SCALAR                      descriptor read in place (round 2: 456.9 vs 465.6 us per step)
#   prepunits   -DIAF_EXP_PREP_SCALAR -DIAF_EXP_PREP_UNITS  + both packs from one pass over the weights (never run)
#   klfold      -DIAF_EXP_FUSED_KL                         KL reductions inside the one-launch step (first form: right, slower)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  IAF_EXTRA_CFLAGS="-DIAF_EXP_PREP_SCALAR" IAF_BUILD_TAG=prepscalar python -m iaf_amd.build > /dev/null
  IAF_EXTRA_CFLAGS="-DIAF_EXP_PREP_SCALAR -DIAF_EXP_PREP_UNITS" IAF_BUILD_TAG=prepunits python -m iaf_amd.build > /dev/null
  IAF_EXTRA_CFLAGS="-DIAF_EXP_FUSED_KL" IAF_BUILD_TAG=klfold python -m iaf_amd.build > /dev/null
  rm -rf iaf_amd/_lib_prepscalar/obj iaf_amd/_lib_prepunits/obj iaf_amd/_lib_klfold/obj      # only the .so needs to travel
fi
cat <<'CMD'
gpurun --timeout 420 -- 'for t in "" prepscalar prepunits; do L=${t:+$GRAFT_REPO_ROOT/iaf_amd/_lib_$t/libiaf_hip.so}; echo "== ${t:-product}";
  IAF_HIP_LIB=$L timeout 60 python bench.py --no-cpu-baseline 2>/dev/null | cut -c90-200;
  IAF_HIP_LIB=$L timeout 60 python bench.py --layers --no-cpu-baseline 2>/dev/null | cut -c120-230; done;
  for t in prepscalar prepunits; do IAF_HIP_LIB=$GRAFT_REPO_ROOT/iaf_amd/_lib_$t/libiaf_hip.so timeout 200 python -m pytest tests -m gpu -x -q 2>&1 | tail -2; done;
  for t in "" klfold; do IAF_HIP_LIB=${t:+$GRAFT_REPO_ROOT/iaf_amd/_lib_$t/libiaf_hip.so} timeout 30 python tools/kl_fold_check.py | tail -7; done'
CMD
