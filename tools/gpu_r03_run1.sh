export TMPDIR=/tmp
O=gpurun_out/${TAG:-r03a}; mkdir -p $O
F='^Load\|^Merge\|^Bvh\|^Scene'
timeout 900 python -m pytest tests/test_gpu_parity.py -k "wide or traversal_operators" -x -q 2>&1 | grep -v "$F" | tail -40 > $O/pytest_wide.log
for m in reference wide; do timeout 300 python tools/gpu_probe_standin.py c5 $m 2>&1 | grep -v "$F" >> $O/probe_c5.log; done
for m in reference wide; do timeout 300 python tools/gpu_standin.py c5 $m 8 3 2>&1 | grep STANDIN >> $O/speed.log; done
for m in reference wide; do timeout 300 python tools/gpu_standin.py c3 $m 32 3 2>&1 | grep STANDIN >> $O/speed.log; done
cat $O/pytest_wide.log $O/probe_c5.log $O/speed.log
