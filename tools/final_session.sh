#!/bin/bash
# usage (GPU box): tools/final_session.sh <tag>     every measurement profiles/<tag>/ keeps, in one call, on one box (then: python tools/fold_profiles.py <tag>; cp gpurun_out/final_<tag>/* profiles/<tag>/)
set -u
TAG=${1:-r06}
REPO=$(pwd); F=$REPO/gpurun_out/final_$TAG; mkdir -p $F; export TMPDIR=/tmp
MEASURE=$REPO/simlod_amd/lib/variants/measure.so          # the library with the in-kernel clocks (make -C simlod_amd/csrc variant NAME=measure DEFS=-DSIMLOD_MEASURE=1)
tools/profile.sh $TAG > $F/profile_sh.txt 2>&1
PMC=0 tools/profile.sh ${TAG}_coalesced --coalesce > $F/profile_sh_coalesced.txt 2>&1
timeout 600 python tools/config3.py --out gpurun_out/final_$TAG/config3_350m > $F/config3_stdout.txt 2>&1
timeout 400 tools/config5_profile.sh $TAG 200000000 > $F/config5_profile_200m.txt 2>&1
cp gpurun_out/config5_$TAG/kernels.txt $F/config5_200m_kernels.txt; cp gpurun_out/config5_$TAG/atomics.txt $F/config5_200m_atomics.txt
timeout 300 python tools/hotspot_ab.py 200000000 > $F/config5_200m.json 2> $F/config5_200m.err
timeout 300 python tools/hotspot_ab.py 200000000 128 300 > $F/config5_200m_host_300mb.json 2>> $F/config5_200m.err
timeout 300 python tools/hotspot_ab.py 20000000 > $F/config5_20m.json 2>> $F/config5_200m.err
timeout 200 python tools/raster_close.py 30 "" "X=1" "SIMLOD_RASTER_SCREEN_BINS=0" 2>&1 | grep -v amdgpu > $F/raster_presets.txt
SIMLOD_HIP_LIB=$MEASURE PRESETS=close timeout 100 python tools/raster_items.py 2>&1 | grep -v amdgpu > $F/raster_items_close.txt
SIMLOD_HIP_LIB=$MEASURE PRESETS=bird timeout 100 python tools/raster_items.py 2>&1 | grep -v amdgpu > $F/raster_items_bird.txt
SIMLOD_HIP_LIB=$MEASURE timeout 100 python tools/raster_bins.py 2>&1 | grep -v amdgpu > $F/raster_bins_close.txt
SIMLOD_HIP_LIB=$MEASURE timeout 300 python tools/raster_big.py 2>&1 | grep -v amdgpu > $F/raster_500m_octree.txt
tools/raster_trace.sh ${TAG}_close close > $F/raster_kernels_close.txt 2>&1
tools/raster_trace.sh ${TAG}_bird bird > $F/raster_kernels_bird.txt 2>&1
tools/trace.sh final_$TAG > $F/ingest_timeline.txt 2>&1
tools/sol_table.sh $TAG > $F/sol_table.txt 2>&1      # every construct kernel alone on the chip: time, own HBM bytes, distance from the copy rate
SIMLOD_HIP_LIB=$MEASURE timeout 200 python tools/probe.py "" "SIMLOD_DEBUG_VOXELIZE_CLOCK=1" 2>&1 | grep -v amdgpu > $F/ingest_phases_measure_build.txt
timeout 200 python tools/probe.py "" "" 2>&1 | grep -v amdgpu > $F/ingest_probe_product_build.txt
timeout 200 python tools/launch_cost.py 2>&1 | grep -v amdgpu > $F/launch_cost.txt
timeout 200 tools/trace_launch.sh final_$TAG > $F/launch_timeline.txt 2>&1
SIMLOD_HOST_HINT=0 timeout 200 python tools/probe.py "" "" 2>&1 | grep "ms/ingest" | cut -c1-110 > $F/ingest_probe_without_host_hint.txt
timeout 200 python tools/las_bench.py 2>&1 | grep -v amdgpu > $F/las_decode.txt
timeout 200 python tools/batch_shape.py 2>&1 | grep -v amdgpu > $F/batch_shape.txt
python tools/fold_profiles.py $TAG > $F/fold.txt 2>&1
mkdir -p profiles/$TAG; cp gpurun_out/final_$TAG/config3_350m* profiles/$TAG/ 2>/dev/null
timeout 900 python bench.py --profiles $TAG > $F/bench_final.json 2> $F/bench_final.err
timeout 300 python bench.py --stream --steps 2 --warmup 1 --no-cpu-baseline --no-profile --profiles $TAG > $F/bench_stream_500m_1gpu.json 2> $F/bench_stream.err
tail -c 600 $F/bench_final.json; ls $F
