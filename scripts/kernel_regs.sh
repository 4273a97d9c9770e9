#!/bin/bash
# VGPR / SGPR / scratch / spills of every kernel in a built library (the code object's own metadata).
#   scripts/kernel_regs.sh [lib.so] [name filter (egrep)]
LIB=$(readlink -f ${1:-ct_mapreduce_amd/libctmr.so}); F=${2:-.}
T=$(mktemp -d); cd $T
objcopy -O binary --only-section=.hip_fatbin "$LIB" fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=co.o
/opt/rocm/lib/llvm/bin/llvm-readelf --notes co.o | awk '/\.name:/{n=$2} /\.vgpr_count:/{v=$2} /\.sgpr_count:/{s=$2} /\.private_segment_fixed_size:/{p=$2} /\.vgpr_spill_count:/{sp=$2; print v, "vgpr", s, "sgpr", p, "scratch", sp, "spill", n}' | c++filt | grep -E "$F" | sort -k1,1n | cut -c1-170
rm -rf $T
