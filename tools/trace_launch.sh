#!/bin/bash
# usage (GPU box, through gpurun): tools/trace_launch.sh <tag> ["ENV=V ..."]
# rocprofv3 kernel trace of tools/launch_cost.py (one batch per kernel_construct launch), one mid-sequence launch printed kernel by kernel
# (tools/launch_timeline.py).  Output: gpurun_out/trace_<tag>/launch_timeline.txt (printed).
TAG=$1; ENVS=${2:-}
REPO=$(pwd); OUT=$REPO/gpurun_out/trace_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
env $ENVS rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $REPO/tools/launch_cost.py > $OUT/launch_cost.txt 2> $OUT/trace.err
cd $REPO
python tools/launch_timeline.py $OUT/trace 20 2 > $OUT/launch_timeline.txt 2>&1
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete 2>/dev/null
cat $OUT/launch_timeline.txt
