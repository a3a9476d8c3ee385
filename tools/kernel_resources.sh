#!/bin/bash
# kernel_resources.sh [extra hipcc flags]: registers / scratch / LDS of every kernel in pt_kernel.hip (gfx950),
# from the compiler's own resource-usage remarks.  No GPU needed.
D=gpu_pathtracer_amd/csrc
FP="-ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize"
for f in ${KERNEL_FILES:-pt_kernel.hip}; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden $FP "$@" --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage -c $D/$f -o /dev/null 2>&1 | grep "remark:" | sed 's/ \[-Rpass.*//' |
  awk '/Function Name/ {name=$NF} / VGPRs:/ {v=$NF} /AGPRs:/ {a=$NF} /TotalSGPRs:/ {s=$NF} /ScratchSize/ {sc=$NF} /Occupancy/ {o=$NF} /LDS Size/ {print name, "vgpr", v, "agpr", a, "sgpr", s, "scratch", sc, "occ", o, "lds", $NF}' |
  while read name rest; do echo "$(echo $name | c++filt | sed 's/void pt:://; s/(pt::DevParams[^)]*)//') $rest"; done
done
