#!/bin/bash
# usage (on the GPU box, through gpurun): tools/r4.sh <tag> ["ENV=V ..." ...]   probe variants + a kernel trace of the default variant
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/r4_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python tools/probe.py --steps 5 "" "$@" > $OUT/probe.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $REPO/tools/probe.py --steps 2 "" > $OUT/trace_probe.txt 2> $OUT/trace.err
cd $REPO
python tools/trace_summary.py $OUT/trace 60 2 > $OUT/timeline.txt 2>&1
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete 2>/dev/null
cat $OUT/probe.txt $OUT/timeline.txt
