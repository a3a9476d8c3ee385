#!/bin/bash
# A/B of var/libgpt_<name>.so builds through the decoupled scheduler: bash tools/gpu_wf_variants.sh <tag> <stand-in> <order> <wf_paths> name...
TAG=$1; W=$2; M=$3; P=$4; shift 4
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python tools/gpu_wavefront.py $W $M 0 $P 2>&1 | grep "^WF" | sed "s/^/product  /" | tee -a $OUT/variants.log
for n in "$@"; do
  GPT_LIB_PATH=$PWD/var/libgpt_$n.so python tools/gpu_wavefront.py $W $M 0 $P 2>&1 | grep "^WF.*stages" | sed "s/^/$n  /" | tee -a $OUT/variants.log
done
