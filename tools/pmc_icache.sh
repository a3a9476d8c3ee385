#!/bin/bash
# instruction-cache / fetch counters of the path kernel (own pass, --kernel-trace only).  usage: bash tools/pmc_icache.sh <tag>
TAG=${1:-ic}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_BRANCH --output-format csv -d $OUT/p1 -o p1 -- python bench.py --no-cpu-baseline --steps 8 --warmup 4 > /dev/null 2> $OUT/p1.err
rocprofv3 --kernel-trace --pmc SQ_IFETCH_LEVEL SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS --output-format csv -d $OUT/p2 -o p2 -- python bench.py --no-cpu-baseline --steps 8 --warmup 4 > /dev/null 2> $OUT/p2.err
python tools/pmc_sq_summary.py $OUT | tee $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -delete
