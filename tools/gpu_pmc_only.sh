#!/bin/bash
# HBM traffic counters only (each in its own pass) -> gpurun_out/<tag>/pmc_summary.txt, pmc_traffic.json
TAG=${1:-pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -o fetch -- python bench.py --no-cpu-baseline --steps 8 --warmup 4 > /dev/null 2> $OUT/prof_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -o write -- python bench.py --no-cpu-baseline --steps 8 --warmup 4 > /dev/null 2> $OUT/prof_write.err
python tools/pmc_summary.py $OUT | tee $OUT/pmc_summary.txt
find $OUT -name "*kernel_trace.csv" -delete
