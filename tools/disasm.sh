#!/bin/bash
# device code of one object of the build:  tools/disasm.sh conv_stream [build_dir_name] > out.s
obj=${1:-conv_stream}; bd=${2:-build}
root=$(cd "$(dirname "$0")/.." && pwd)
t=$(mktemp -d)
llvm=/opt/rocm/lib/llvm/bin
cp $root/realtime-pose-estimation_amd/$bd/$obj.hip.o $t/s.o
$llvm/llvm-objcopy --dump-section .hip_fatbin=$t/s.fatbin $t/s.o $t/s2.o
$llvm/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$t/s.fatbin --output=$t/s.co
$llvm/llvm-objdump -d $t/s.co
rm -rf $t
