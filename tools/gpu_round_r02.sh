#!/bin/bash
# One GPU-box session of round 2: smoke, both test suites, bench (with its in-run counters), rocprofv3 kernel stats of the same
# command, all configurations, and the memory counters of the config-5 stand-in (each --pmc set in its own pass).
# Usage (from the repo root on the GPU box): bash tools/gpu_round_r02.sh <tag>
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
F='^Load\|^Merge\|^Bvh\|^Scene'
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" > $OUT/smoke.log; tail -1 $OUT/smoke.log
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" > $OUT/pytest_gpu.log; grep "passed\|failed" $OUT/pytest_gpu.log
timeout 1500 python -m pytest tests -m gpu -q --gpt-opt lds_scene=0 2>&1 | grep -v "$F" > $OUT/pytest_gpu_nolds.log; grep "passed\|failed" $OUT/pytest_gpu_nolds.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-300 $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o stats -- python bench.py --no-cpu-baseline --no-counters --no-parity --no-square > $OUT/bench_under_rocprof.json 2> $OUT/prof_stats.err
for f in $(find $OUT/prof_stats -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; head -6 $f; done
python tools/gpu_configs.py 2>/dev/null | grep -v "$F" > $OUT/configs.log; cat $OUT/configs.log
for mode in reference wide; do
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
    tag=$(echo $pmc | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/c5_${mode}_$tag -o p -- python tools/gpu_standin.py c5 $mode 8 2 2>/dev/null | grep STANDIN > $OUT/c5_${mode}_$tag.log
  done
done
python - <<PY > $OUT/c5_pmc_summary.txt
import csv, glob, collections, re
out = "$OUT"
for mode in ("reference", "wide"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/c5_{mode}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "pt_render_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    # the 4K 8-iteration launches are the large ones (the 2-iteration warm-up launch comes first)
    big = {k: max(v) for k, v in acc.items()}
    line = open(glob.glob(f"{out}/c5_{mode}_FETCH_SIZE.log")[0]).read().strip()
    ms = float(re.search(r"([0-9.]+) ms per launch", line).group(1))
    print(f"== config-5 stand-in, {mode} order: one launch = 3840x2160 x 8 iterations = 66.4 M samples; un-profiled-pass time {ms:.1f} ms per launch")
    for k in sorted(big): print(f"   {k:26s} {big[k]:.5g}")
    if "FETCH_SIZE" in big and "WRITE_SIZE" in big:
        f, w = big["FETCH_SIZE"] * 1024, big["WRITE_SIZE"] * 1024
        print(f"   HBM-side traffic per launch: fetch {f/1e9:.1f} GB raw ({2*f/1e9:.1f} GB with the gfx950 x2 correction), write {w/1e9:.2f} GB "
              f"(compulsory sample planes: {66.36e6*16/1e9:.2f} GB) -> {(f+w)/ms/1e6:.0f} - {(2*f+w)/ms/1e6:.0f} GB/s = "
              f"{(f+w)/ms/1e6/8000*100:.0f} - {(2*f+w)/ms/1e6/8000*100:.0f} % of the 8 TB/s peak")
    if "TCC_HIT_sum" in big: print(f"   L2 hit rate {big['TCC_HIT_sum']/(big['TCC_HIT_sum']+big['TCC_MISS_sum']):.3f}")
    if "SQ_INSTS_VALU" in big:
        print(f"   VALU wave-instructions {big['SQ_INSTS_VALU']:.4g} ({big['SQ_INSTS_VALU']/66.36e6*64:.0f} per 64 samples), lanes active {big['SQ_THREAD_CYCLES_VALU']/big['SQ_ACTIVE_INST_VALU']:.1f} of 64, "
              f"issue rate {big['SQ_INSTS_VALU']/ms/1e6:.0f} G/s of 1228.8 = {big['SQ_INSTS_VALU']/ms/1e6/1228.8:.2f}, waiting (s_waitcnt) {big['SQ_WAIT_ANY']/big['SQ_WAVE_CYCLES']:.2f} of wave cycles")
PY
cat $OUT/c5_pmc_summary.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +1M -delete
du -sh $OUT
