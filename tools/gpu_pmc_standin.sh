#!/bin/bash
# SQ / TCC counters of one stand-in in one traversal mode (each --pmc set in its own pass, kernel-trace only).
# usage (GPU box): bash tools/gpu_pmc_standin.sh <tag> <c3|c4|c5> <reference|wide> [iterations]
TAG=$1; WHICH=${2:-c5}; MODE=${3:-wide}; SPP=${4:-8}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"; do
  tag=$(echo $pmc | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/${WHICH}_${MODE}_$tag -o p -- python tools/gpu_standin.py $WHICH $MODE $SPP 2 2>/dev/null | grep STANDIN > $OUT/${WHICH}_${MODE}_$tag.log
done
python - <<PY | tee $OUT/${WHICH}_${MODE}_pmc_summary.txt
import csv, glob, collections, re
out, which, mode, spp = "$OUT", "$WHICH", "$MODE", $SPP
acc = collections.defaultdict(list)
for f in glob.glob(f"{out}/{which}_{mode}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pt_render_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
big = {k: max(v) for k, v in acc.items()}
line = open(glob.glob(f"{out}/{which}_{mode}_FETCH_SIZE.log")[0]).read().strip()
ms = float(re.search(r"([0-9.]+) ms per launch", line).group(1))
w, h = map(int, re.search(r"(\d+)x(\d+)", line).groups())
n = w * h * spp
print(f"== {which} stand-in, {mode} order: one launch = {w}x{h} x {spp} iterations = {n/1e6:.1f} M samples; {ms:.1f} ms per launch (profiled pass) = {n/ms/1e3:.0f} Msamples/s")
for k in sorted(big): print(f"   {k:26s} {big[k]:.5g}")
if "FETCH_SIZE" in big and "WRITE_SIZE" in big:
    f, wr = big["FETCH_SIZE"] * 1024, big["WRITE_SIZE"] * 1024
    print(f"   HBM-side traffic per launch: fetch {f/1e9:.1f} GB raw ({2*f/1e9:.1f} GB with the gfx950 x2 correction), write {wr/1e9:.2f} GB "
          f"(compulsory sample planes: {n*16/1e9:.2f} GB) -> {(f+wr)/ms/1e6:.0f} - {(2*f+wr)/ms/1e6:.0f} GB/s = {(f+wr)/ms/1e6/8000*100:.0f} - {(2*f+wr)/ms/1e6/8000*100:.0f} % of the 8 TB/s peak")
if "TCC_HIT_sum" in big: print(f"   L2 hit rate {big['TCC_HIT_sum']/(big['TCC_HIT_sum']+big['TCC_MISS_sum']):.3f}")
if "SQ_INSTS_VALU" in big:
    print(f"   VALU wave-instructions {big['SQ_INSTS_VALU']:.4g} ({big['SQ_INSTS_VALU']/n*64:.0f} per 64 samples), lanes active {big['SQ_THREAD_CYCLES_VALU']/big['SQ_ACTIVE_INST_VALU']:.1f} of 64, "
          f"issue rate {big['SQ_INSTS_VALU']/ms/1e6:.0f} G/s of 1228.8 = {big['SQ_INSTS_VALU']/ms/1e6/1228.8:.2f}, waiting (s_waitcnt) {big['SQ_WAIT_ANY']/big['SQ_WAVE_CYCLES']:.2f} of wave cycles, issue stalls {big['SQ_WAIT_INST_ANY']/big['SQ_WAVE_CYCLES']:.2f}")
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +1M -delete; find $OUT -name "*agent_info.csv" -delete
