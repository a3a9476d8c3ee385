#!/bin/bash
# SQ / TCC counters of the decoupled scheduler's two stages on one stand-in (each --pmc set in its own pass, kernel-trace only), summed over
# the launches of ONE batch.  usage (GPU box): bash tools/gpu_pmc_wf.sh <tag> <c3|c4|c5> <reference|wide> [iterations] [wf_paths]
TAG=$1; WHICH=${2:-c5}; MODE=${3:-wide}; SPP=${4:-8}; PATHS=${5:-1048576}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SMEM"; do
  tag=$(echo $pmc | cut -d' ' -f1)
  GPT_WF_ONE_BATCH=1 timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/wf_${WHICH}_${MODE}_$tag -o p -- python tools/gpu_wavefront.py $WHICH $MODE $SPP $PATHS 2>/dev/null | grep "^WF" > $OUT/wf_${WHICH}_${MODE}_$tag.log
done
python - <<PY | tee $OUT/wf_${WHICH}_${MODE}_pmc_summary.txt
import csv, glob, collections
out, which, mode, spp = "$OUT", "$WHICH", "$MODE", $SPP
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(f"{out}/wf_{which}_{mode}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "phases" if "wf_render" in r["Kernel_Name"] else None
        if k:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k][r["Counter_Name"]] += 1
print(open(glob.glob(f"{out}/wf_{which}_{mode}_FETCH_SIZE.log")[0]).read().strip())
for k in ("phases",):
    a = acc[k]
    print(f"== wf_render_kernel ({k}), all launches of the profiled run summed ({max(calls[k].values()) if calls[k] else 0} launches)")
    for c in sorted(a): print(f"   {c:26s} {a[c]:.5g}")
    if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
        print(f"   HBM-side traffic: fetch {a['FETCH_SIZE']*1024/1e9:.1f} GB raw ({2*a['FETCH_SIZE']*1024/1e9:.1f} GB with the gfx950 x2 correction), write {a['WRITE_SIZE']*1024/1e9:.1f} GB")
    if "TCC_HIT_sum" in a: print(f"   L2 hit rate {a['TCC_HIT_sum']/(a['TCC_HIT_sum']+a['TCC_MISS_sum']):.3f}")
    if "SQ_INSTS_VALU" in a and a.get("SQ_ACTIVE_INST_VALU"):
        print(f"   VALU wave-instructions {a['SQ_INSTS_VALU']:.4g}, lanes active {a['SQ_THREAD_CYCLES_VALU']/a['SQ_ACTIVE_INST_VALU']:.1f} of 64, waiting (s_waitcnt) {a['SQ_WAIT_ANY']/a['SQ_WAVE_CYCLES']:.2f} of wave cycles, issue stalls {a['SQ_WAIT_INST_ANY']/a['SQ_WAVE_CYCLES']:.2f}, "
              f"vector-memory reads {a['SQ_INSTS_VMEM_RD']:.4g}, writes {a['SQ_INSTS_VMEM_WR']:.4g}")
    if "SQ_BUSY_CYCLES" in a:
        print(f"   SQ_BUSY_CYCLES {a['SQ_BUSY_CYCLES']:.4g}, GRBM_GUI_ACTIVE {a['GRBM_GUI_ACTIVE']:.4g}, waves {a['SQ_WAVES']:.4g}, SALU {a['SQ_INSTS_SALU']:.4g}, LDS {a['SQ_INSTS_LDS']:.4g}, SMEM {a['SQ_INSTS_SMEM']:.4g}")
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +1M -delete; find $OUT -name "*agent_info.csv" -delete
