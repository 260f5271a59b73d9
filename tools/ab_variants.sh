#!/bin/bash
# usage (GPU box): tools/ab_variants.sh <variant> ...     frame times of tools/raster_close.py with simlod_amd/lib/variants/<variant>.so ("-" = the in-tree library), same box, same run
for v in "$@"; do
  if [ "$v" != "-" ]; then export SIMLOD_HIP_LIB=$PWD/simlod_amd/lib/variants/$v.so; else unset SIMLOD_HIP_LIB; fi
  echo "== variant $v"
  timeout 200 python tools/raster_close.py 30 "" "X=1" 2>&1 | grep "X=1" | cut -c1-110
done
