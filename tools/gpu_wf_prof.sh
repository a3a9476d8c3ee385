#!/bin/bash
# rocprofv3 kernel stats of one stand-in through the decoupled scheduler: bash tools/gpu_wf_prof.sh <tag> <c3|c4|c5> <order> [iterations] [wf_paths]
TAG=$1; W=$2; M=$3; N=${4:-8}; P=${5:-1048576}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_wf_$W -o stats -- python tools/gpu_wavefront.py $W $M $N $P > $OUT/prof_wf_${W}_$M.out 2>&1
grep "^WF" $OUT/prof_wf_${W}_$M.out
for f in $(find $OUT/prof_wf_$W -name "*kernel_stats.csv"); do cp $f $OUT/wf_${W}_${M}_kernel_stats.csv; head -8 $f; done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
