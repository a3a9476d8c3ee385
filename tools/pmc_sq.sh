#!/bin/bash
# SQ-level counters of the path kernel (own passes, --kernel-trace only): lane utilisation, VALU busy, waits.
# usage: bash tools/pmc_sq.sh <tag> [variant-name]   (variant = var/libgpt_<name>.so)
TAG=${1:-sq}
if [ -n "$2" ]; then export GPT_LIB_PATH=$PWD/var/libgpt_$2.so; fi
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/p1 -o p1 -- python bench.py --no-cpu-baseline --steps 8 --warmup 4 > /dev/null 2> $OUT/p1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -o p2 -- python bench.py --no-cpu-baseline --steps 8 --warmup 4 > /dev/null 2> $OUT/p2.err
python tools/pmc_sq_summary.py $OUT | tee $OUT/sq_summary.txt
find $OUT -name "*kernel_trace.csv" -delete
