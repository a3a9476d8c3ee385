#!/bin/bash
# build_variant.sh <name> [extra hipcc flags...]  ->  build/variants/libgpt_<name>.so  (kernel experiments)
set -e
NAME=$1; shift
D=gpu_pathtracer_amd/csrc
mkdir -p build/variants
FP="-ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden $FP "$@" -c $D/pt_kernel.hip -o build/variants/pt_kernel_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -ldl -o build/variants/libgpt_$NAME.so build/variants/pt_kernel_$NAME.o $D/render_api.o $D/host_prep.o $D/host_util.o $D/scene_loader.o $D/imageio.o $D/pathtracer_cxx.o
