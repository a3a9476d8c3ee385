#!/bin/bash
# gpu_ab.sh <tag> name...: A/B of var/libgpt_<name>.so builds on the GPU box: the headline (bit-compared films) and the c3 / c4 / c5
# stand-ins in the default order.  Output in gpurun_out/<tag>/ab.log
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export GPT_ALLOW_OLD_LIB=1
python tools/bench_variants.py "$@" 2>&1 | tee $OUT/ab.log
for v in "$@"; do
  export GPT_LIB_PATH=$PWD/var/libgpt_$v.so
  echo "== $v" | tee -a $OUT/ab.log
  python tools/gpu_configs.py 2>&1 | grep "SURVEY stand-in" | sed 's/SURVEY stand-in: //' | cut -c1-150 | tee -a $OUT/ab.log
done
# Volpath: the shipped scene's shape on the one-ray kernel, the fog box on both kernels
for v in "$@"; do
  export GPT_LIB_PATH=$PWD/var/libgpt_$v.so
  echo "== volpath $v" | tee -a $OUT/ab.log
  python tools/gpu_volpath.py 2>&1 | grep "Msamples" | cut -c1-110 | tee -a $OUT/ab.log
done
