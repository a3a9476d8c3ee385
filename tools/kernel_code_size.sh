#!/bin/bash
# kernel_code_size.sh [extra hipcc flags]: bytes of machine code of every kernel in pt_kernel.hip (gfx950), from the
# symbol table of the device code object.  No GPU needed.  (The instruction cache of a CU pair is 64 KB.)
D=gpu_pathtracer_amd/csrc
FP="-ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize"
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden $FP "$@" --cuda-device-only -c $D/pt_kernel.hip -o $T/dev.co || exit 1
/opt/rocm/lib/llvm/bin/llvm-readelf -sW $T/dev.co | awk '$4=="FUNC" {print $3, $8}' |
  while read sz nm; do echo "$sz $(echo $nm | c++filt | sed 's/void pt:://; s/(pt::DevParams[^)]*)//')"; done | sort -n
rm -rf $T
