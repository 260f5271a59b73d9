#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + the two PMC passes the HBM-traffic figure needs.
# Usage: tools/profile.sh <tag>      outputs land in gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 2 --warmup 1 --frames 4 --no-cpu-baseline --no-profile"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace_bench.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- $CMD > /dev/null 2> $OUT/pmc_l2.err
cd $REPO
python tools/summarize_profile.py $OUT $TAG
find $OUT -name "*.csv" -size +3M -delete
ls -la $OUT $OUT/* | head -40
