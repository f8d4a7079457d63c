#!/bin/bash
# VGPR / scratch use of every kernel in an object built by __graft_entry__.build():  tools/kernel_regs.sh build/frost_dw3.o [filter]
obj=$1; filt=${2:-.}
tmp=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin $obj $tmp/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fat.bin --output=$tmp/dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/dev.co | awk '/\.name:/{n=$2} /\.private_segment_fixed_size:/{s=$2} /\.vgpr_count:/{v=$2} /\.agpr_count:/{a=$2} /\.vgpr_spill_count:/{sp=$2; print v, a, s, sp, n}' | while read v a s sp n; do echo "vgpr=$v agpr=$a scratch=$s spill=$sp $(echo $n | c++filt | cut -c1-120)"; done | grep -E "$filt"
rm -rf $tmp
