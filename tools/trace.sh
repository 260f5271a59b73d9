#!/bin/bash
# usage (GPU box, through gpurun): tools/trace.sh <tag> ["ENV=V ..."] [probe flags]
# rocprofv3 kernel trace of tools/probe.py (2 ingests of the bench terrain) under the given environment (SIMLOD_HIP_LIB=... picks another build of the
# library), folded by tools/trace_summary.py: per-kernel averages over the launches with work, the timeline of one mid-ingest batch on both streams,
# the per-batch cycle of the last ingest.  Output: gpurun_out/trace_<tag>/timeline.txt (printed).
TAG=$1; ENVS=${2:-}; shift; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/trace_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
env $ENVS rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $REPO/tools/probe.py --steps 2 "$@" "" > $OUT/trace_probe.txt 2> $OUT/trace.err
cd $REPO
python tools/trace_summary.py $OUT/trace 60 1 > $OUT/timeline.txt 2>&1
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete 2>/dev/null
cat $OUT/timeline.txt
