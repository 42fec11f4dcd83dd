# sha256 of the gfx950 device code (.text of the embedded code object) of every object of the product build:
#   bash tools/device_code_digest.sh > /tmp/a.txt ; (edit sources, python -m iaf_amd.build) ; bash tools/device_code_digest.sh | diff /tmp/a.txt -
# An experiment switch (#ifdef IAF_EXP_*) must leave every line unchanged: that is how this round kept staging experiments
# after the GPU budget was spent without touching what the GPU suite had verified.
L=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
for o in "$(dirname "$0")"/../iaf_amd/_lib/obj/*.o; do
  n=$(basename $o .o)
  cp $o $T/$n.o
  $L/llvm-objcopy --dump-section .hip_fatbin=$T/$n.bin $T/$n.o $T/$n.copy.o 2>/dev/null &&
  $L/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/$n.bin --output=$T/$n.co --unbundle 2>/dev/null &&
  $L/llvm-objcopy -O binary --only-section=.text $T/$n.co $T/$n.text &&
  echo "$(sha256sum < $T/$n.text | cut -c1-16)  $n"
done
rm -rf $T
