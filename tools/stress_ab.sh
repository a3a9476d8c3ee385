for v in asm7 cur; do
  if [ $v = cur ]; then unset GPT_LIB_PATH; else export GPT_LIB_PATH=$PWD/var/libgpt_$v.so; fi
  echo "== $v"; python tools/gpu_stress.py 2>&1 | grep -v amdgpu.ids | tail -2
done
