for v in "$@"; do
  if [ $v = cur ]; then unset GPT_LIB_PATH; else export GPT_LIB_PATH=$PWD/var/libgpt_$v.so; fi
  echo "== $v"; python tools/gpu_configs.py 2>&1 | grep "config [345]" | grep -v near | cut -c1-75
done
