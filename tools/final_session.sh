#!/bin/bash
# usage (GPU box): tools/final_session.sh <tag>     every measurement profiles/<tag>/ keeps, in one call, on one box (then: python tools/fold_profiles.py <tag>; cp gpurun_out/final_<tag>/* profiles/<tag>/)
set -u
TAG=${1:-r04}
REPO=$(pwd); F=$REPO/gpurun_out/final_$TAG; mkdir -p $F; export TMPDIR=/tmp
tools/profile.sh $TAG > $F/profile_sh.txt 2>&1
PMC=0 tools/profile.sh ${TAG}_coalesced --coalesce > $F/profile_sh_coalesced.txt 2>&1
timeout 600 python tools/config3.py --out gpurun_out/final_$TAG/config3_350m > $F/config3_stdout.txt 2>&1
timeout 300 python tools/hotspot_ab.py 200000000 > $F/config5_200m.json 2> $F/config5_200m.err
timeout 300 python tools/hotspot_ab.py 20000000 > $F/config5_20m.json 2>> $F/config5_200m.err
timeout 300 python bench.py --stream --steps 2 --warmup 1 --no-cpu-baseline --no-profile --profiles $TAG > $F/bench_stream_500m_1gpu.json 2> $F/bench_stream.err
timeout 200 python tools/raster_close.py 30 "" "X=1" "SIMLOD_RASTER_SCREEN_BINS=0" 2>&1 | grep -v amdgpu > $F/raster_presets.txt
PRESETS=close timeout 100 python tools/raster_items.py 2>&1 | grep -v amdgpu > $F/raster_items_close.txt
PRESETS=bird timeout 100 python tools/raster_items.py 2>&1 | grep -v amdgpu > $F/raster_items_bird.txt
timeout 100 python tools/raster_bins.py 2>&1 | grep -v amdgpu > $F/raster_bins_close.txt
tools/raster_trace.sh ${TAG}_close close > $F/raster_kernels_close.txt 2>&1
tools/raster_trace.sh ${TAG}_bird bird > $F/raster_kernels_bird.txt 2>&1
tools/trace.sh final > $F/ingest_timeline.txt 2>&1
python tools/fold_profiles.py $TAG > $F/fold.txt 2>&1
mkdir -p profiles/$TAG; cp gpurun_out/final_$TAG/config3_350m* profiles/$TAG/ 2>/dev/null
timeout 900 python bench.py --profiles $TAG > $F/bench_final.json 2> $F/bench_final.err
tail -c 600 $F/bench_final.json; ls $F
