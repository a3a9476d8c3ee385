export TMPDIR=/tmp; mkdir -p gpurun_out/${1:-vp}
for pmc in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pmc | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d gpurun_out/${1:-vp}/vp_$tag -o p -- python tools/gpu_volpath.py shipped > gpurun_out/${1:-vp}/vp_$tag.log 2>&1
done
D=gpurun_out/${1:-vp} python - <<'PY'
import csv, glob, collections, os
D = os.environ["D"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(D + "/vp_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pt_render_kernel" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, a in acc.items():
    b = {c: max(v) for c, v in a.items()}
    print("==", k)
    if "SQ_INSTS_VALU" in b:
        print(f"   VALU {b['SQ_INSTS_VALU']:.4g}, lanes {b['SQ_THREAD_CYCLES_VALU']/b['SQ_ACTIVE_INST_VALU']:.1f}, wait {b['SQ_WAIT_ANY']/b['SQ_WAVE_CYCLES']:.2f}, issue stalls {b['SQ_WAIT_INST_ANY']/b['SQ_WAVE_CYCLES']:.2f}, vmem rd {b['SQ_INSTS_VMEM_RD']:.4g}, lds {b['SQ_INSTS_LDS']:.4g}")
    if "TCC_HIT_sum" in b: print(f"   L2 hit {b['TCC_HIT_sum']/(b['TCC_HIT_sum']+b['TCC_MISS_sum']):.3f}")
    if "FETCH_SIZE" in b: print(f"   fetch {b['FETCH_SIZE']*1024/1e9:.2f} GB raw, write {b.get('WRITE_SIZE',0)*1024/1e9:.2f} GB per 16.8 M-sample launch (512 x 512 x 64; compulsory sample planes 0.27 GB)")
    if "SQ_INSTS_VALU" in b: print(f"   VALU wave-instructions per sample {b['SQ_INSTS_VALU']/(512*512*64):.0f}")
PY
grep "Msamples" gpurun_out/${1:-vp}/vp_FETCH_SIZE.log
find gpurun_out/${1:-vp}/vp_* -name "*.csv" -delete
