#!/bin/bash
# usage (GPU box): tools/r4t.sh <tag> "ENV=V ..."    kernel trace of probe.py under the given environment, summary printed
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/r4_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
env $1 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $REPO/tools/probe.py --steps 2 "" > $OUT/trace_probe.txt 2> $OUT/trace.err
cd $REPO
python tools/trace_summary.py $OUT/trace 60 1 > $OUT/timeline.txt 2>&1
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete 2>/dev/null
cat $OUT/timeline.txt
