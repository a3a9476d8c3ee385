#!/bin/bash
# isa_diff.sh <base.s | --save base.s> [extra hipcc flags]: device ISA of pt_kernel.hip (gfx950) as text, compared with a saved
# copy (the compile-unit id lines aside).  Used to show that a refactoring leaves the product kernels byte-identical.  No GPU.
D=gpu_pathtracer_amd/csrc
FP="-ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize"
if [ "$1" = "--save" ]; then OUT=$2; shift 2; else BASE=$1; OUT=$(mktemp /tmp/isa_XXXX.s); shift; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden $FP "$@" --cuda-device-only -S $D/pt_kernel.hip -o $OUT 2>&1 | grep -v hip-link
[ -z "$BASE" ] && exit 0
if diff <(grep -v __hip_cuid $BASE) <(grep -v __hip_cuid $OUT) > /dev/null; then echo "ISA identical ($(grep -c . $OUT) lines)"; else echo "ISA DIFFERS"; diff <(grep -v __hip_cuid $BASE) <(grep -v __hip_cuid $OUT) | head -20; exit 1; fi
