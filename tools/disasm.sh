# gfx950 disassembly of one object file:  bash tools/disasm.sh obj.o > out.s
L=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$L/llvm-objcopy --dump-section .hip_fatbin=$T/a.bin "$1" $T/copy.o 2>/dev/null &&
$L/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/a.bin --output=$T/a.co --unbundle 2>/dev/null &&
$L/llvm-objdump -d $T/a.co
rm -rf $T
