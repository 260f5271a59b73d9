#!/bin/bash
# A/B of two builds of the library on ONE box (box-to-box spread is ~10 %): tools/raster_ab.sh libA.so libB.so [rounds]
# prints ms/frame of tools/raster_prof.py for each, alternating
root=$(cd "$(dirname "$0")/.." && pwd)
A=$1; B=$2; R=${3:-2}
for r in $(seq $R); do
  for L in $A $B; do
    echo "== $L"; SIMLOD_HIP_LIB=$root/$L python $root/tools/raster_prof.py 36000000 40 2>/dev/null | grep ms/frame
  done
done
