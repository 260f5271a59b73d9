#!/bin/bash
# usage (GPU box): tools/final_core.sh <tag>     the part of tools/final_session.sh that bench.py's gating and the headline numbers need — profile passes, fold,
# config 3, the two bench lines, launch and idle costs — for a last change that touched host logic only (the other files of profiles/<tag>/ keep their sha)
set -u
TAG=${1:-r06}
REPO=$(pwd); F=$REPO/gpurun_out/final_$TAG; mkdir -p $F; export TMPDIR=/tmp
tools/profile.sh $TAG > $F/profile_sh.txt 2>&1
python tools/fold_profiles.py $TAG > $F/fold.txt 2>&1
timeout 200 python tools/config3.py --out gpurun_out/final_$TAG/config3_350m > $F/config3_stdout.txt 2>&1
mkdir -p profiles/$TAG; cp gpurun_out/final_$TAG/config3_350m* profiles/$TAG/ 2>/dev/null
timeout 200 python bench.py --profiles $TAG > $F/bench_final.json 2> $F/bench_final.err
timeout 100 python bench.py --stream --steps 2 --warmup 1 --no-cpu-baseline --no-profile --profiles $TAG > $F/bench_stream_500m_1gpu.json 2> $F/bench_stream.err
timeout 60 python tools/launch_cost.py 2>&1 | grep -v amdgpu > $F/launch_cost.txt
timeout 60 python tools/idle_launch.py 2>&1 | grep -v amdgpu > $F/idle_launch.txt
SIMLOD_HOST_HINT=0 timeout 60 python tools/probe.py "" "" 2>&1 | grep "ms/ingest" | cut -c1-110 > $F/ingest_probe_without_host_hint.txt
cat $F/idle_launch.txt; tail -c 200 $F/bench_final.json
