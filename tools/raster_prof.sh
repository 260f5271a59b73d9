#!/bin/bash
# per-kernel times of the rasteriser on the bench octree: rocprofv3 kernel trace of tools/raster_prof.py -> gpurun_out/raster_<tag>.txt
tag=${1:-r}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/raster_prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out -o run --output-format csv -- python $root/tools/raster_prof.py > $out/stdout.txt 2> $out/stderr.txt
grep -h "ms/frame" $out/stdout.txt
f=$(find $out -name "*kernel_stats.csv" | head -1)
grep "r_" $f | awk -F'","' '{gsub(/"/,"",$1); printf "%-50s calls %5s avg %9.1f us\n", $1, $2, $4/1000}'
