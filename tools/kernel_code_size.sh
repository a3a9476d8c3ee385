#!/bin/bash
# kernel_code_size.sh [extra hipcc flags]: registers, scratch and bytes of machine code of every non-counting render kernel in pt_kernel.hip
# (gfx950), from the compiler's resource remarks and the symbol table of the device code object.  No GPU needed.
# (The instruction cache shared by a pair of CUs is 64 KB.)
cd "$(dirname "$0")/.."
D=gpu_pathtracer_amd/csrc
FP="-ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize"
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden $FP "$@" --cuda-device-only -Rpass-analysis=kernel-resource-usage \
  -c $D/pt_kernel.hip -o $T/dev.co 2> $T/remarks.txt || { grep -E "error" -A5 $T/remarks.txt | head -40; exit 1; }
grep "remark:" $T/remarks.txt | sed 's/ \[-Rpass.*//' |
  awk '/Function Name/ {name=$NF} / VGPRs:/ {v=$NF} /ScratchSize/ {sc=$NF} /LDS Size/ {print name, "vgpr", v, "scratch", sc}' |
  while read name rest; do echo "$(echo $name | c++filt | sed 's/void pt:://; s/(pt::DevParams[^)]*)//; s/, /,/g') $rest"; done | grep "render_kernel<false" | sort > $T/a.txt
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/dev.co --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.elf
/opt/rocm/lib/llvm/bin/llvm-readelf -sW $T/dev.elf | awk '$4=="FUNC" {print $3, $8}' |
  while read sz nm; do echo "$(echo $nm | c++filt | sed 's/void pt:://; s/(pt::DevParams[^)]*)//; s/, /,/g') code $sz"; done | sort -u | grep "render_kernel<false" > $T/b.txt
join -j1 $T/a.txt $T/b.txt
rm -rf $T
