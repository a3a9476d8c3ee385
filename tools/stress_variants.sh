# usage: bash tools/stress_variants.sh name...   (4K, 8 spp timing of the 253k-triangle stand-in per variant lib)
for v in "$@"; do
  if [ $v = cur ]; then unset GPT_LIB_PATH; else export GPT_LIB_PATH=$PWD/var/libgpt_$v.so; fi
  echo "== $v: $(python tools/gpu_stress.py 2>&1 | grep 3840x2160 | cut -c1-60)"
done
