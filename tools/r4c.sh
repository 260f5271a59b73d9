#!/bin/bash
# usage (GPU box): tools/r4c.sh <tag>    coalesced mode: probe + kernel trace summary
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/r4_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python tools/probe.py --coalesce --steps 5 "" "$@" > $OUT/probe.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $REPO/tools/probe.py --coalesce --steps 2 "" > $OUT/trace_probe.txt 2> $OUT/trace.err
cd $REPO
python tools/trace_summary.py $OUT/trace 4 1 > $OUT/timeline.txt 2>&1
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete 2>/dev/null
grep -v "us per call\|k_voxelize of\|amdgpu" $OUT/probe.txt; head -40 $OUT/timeline.txt
