#!/bin/bash
# tracking-step cap sweep for the general Volpath kernel (var/libgpt_ts*.so from tools/build_variant.sh)
for v in ${VARIANTS:-default ts12 ts24 ts96 ts1000000}; do
  if [ $v = default ]; then unset GPT_LIB_PATH; else export GPT_LIB_PATH=$PWD/var/libgpt_$v.so; fi
  echo "== $v"; timeout 300 python tools/gpu_volpath.py 2>&1 | grep -E "one-ray"
done
