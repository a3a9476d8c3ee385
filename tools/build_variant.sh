#!/bin/bash
# build_variant.sh <name> [extra hipcc flags...]  ->  var/libgpt_<name>.so  (kernel experiments: the kernel translation unit
# is rebuilt with the flags, the host objects are the product's)
set -e
NAME=$1; shift
D=gpu_pathtracer_amd/csrc
mkdir -p var
FP="-ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden $FP "$@" -c $D/pt_kernel.hip -o var/pt_kernel_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -ldl -o var/libgpt_$NAME.so var/pt_kernel_$NAME.o $D/render_api.o $D/host_prep.o $D/sbvh_build.o $D/host_util.o $D/scene_loader.o $D/imageio.o $D/pathtracer_cxx.o
