#!/bin/bash
# tools/kernel_isa.sh <object file> <mangled-name substring> : disassembly of one gfx950 kernel of a library object (CPU, no GPU needed)
set -e
LLVM=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$LLVM/llvm-objcopy --dump-section .hip_fatbin=$T/f.fatbin "$1"
$LLVM/clang-offload-bundler --unbundle --type=o --input=$T/f.fatbin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/f.co
$LLVM/llvm-objdump -d --no-show-raw-insn $T/f.co | awk -v pat="$2" '/^[0-9a-f]+ <.*>:/ { on = index($0, pat) > 0 } on { print }'
rm -rf $T
