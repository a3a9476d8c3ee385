export GPT_ALLOW_OLD_LIB=1 TMPDIR=/tmp
mkdir -p gpurun_out/d1
(for v in cur wl4 wl5 wl6 cur wl5; do
  export GPT_LIB_PATH=$PWD/var/libgpt_$v.so
  echo "== $v $(timeout 150 python tools/gpu_volpath.py 2>&1 | grep 'Msamples' | grep 'one-ray\|shipped' | sed 's/: .* -> /: /; s/ Msamples.*//' | tr '\n' ';')"
done) 2>&1 | tee gpurun_out/d1/volpath_waves.log
# the Volpath tests on the 5- and 6-wave builds (film == oracle bit for bit)
for v in wl5 wl6; do GPT_LIB_PATH=$PWD/var/libgpt_$v.so timeout 300 python -m pytest tests -m gpu -q -k "volpath" 2>&1 | grep -a "passed\|failed" | tail -1 | sed "s/^/$v volpath tests: /"; done | tee -a gpurun_out/d1/volpath_waves.log
(for v in cur st8 st12 st16 st20 cur st16 st12; do
  export GPT_LIB_PATH=$PWD/var/libgpt_$v.so
  echo "== $v $(timeout 200 python tools/gpu_configs.py 2>&1 | grep 'SURVEY stand-in' | grep wide | sed 's/.*: *\([0-9.]*\) Msamples.*/\1/' | tr '\n' ' ')"
done) 2>&1 | tee gpurun_out/d1/wide_stop_sweep.log
