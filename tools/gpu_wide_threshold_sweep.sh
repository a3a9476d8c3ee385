export GPT_ALLOW_OLD_LIB=1 TMPDIR=/tmp
mkdir -p gpurun_out/c5
for v in cur st16 st32 st40 ss8 ss18 ss24 lm4 lm12 ft8 ft16 cur; do
  export GPT_LIB_PATH=$PWD/var/libgpt_$v.so
  echo "== $v $(timeout 200 python tools/gpu_configs.py 2>&1 | grep 'SURVEY stand-in' | grep wide | sed 's/.*: *\([0-9.]*\) Msamples.*/\1/' | tr '\n' ' ')"
done | tee gpurun_out/c5/wide_threshold_sweep.log
