#!/bin/bash
# usage (GPU box): tools/ab_ingest.sh [notests]    the GPU test tier, then ingest time (tools/probe.py) and one-batch launch cost (tools/launch_cost.py) of
# simlod_amd/lib/variants/r5base.so (round 5 before the launch-sizing change) against the in-tree library, with and without the host's pending-batches hint, same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "$1" != "notests" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q --timeout 400 > gpurun_out/ab5_tests.log 2>&1
  grep -E "passed|failed|error" gpurun_out/ab5_tests.log | tail -3
fi
for v in "r5base 0" "- 0" "- 1" "r5base 0" "- 0" "- 1"; do
  set -- $v
  if [ "$1" != "-" ]; then export SIMLOD_HIP_LIB=$PWD/simlod_amd/lib/variants/$1.so; else unset SIMLOD_HIP_LIB; fi
  export SIMLOD_HOST_HINT=$2
  echo "== library $1, host hint $2"
  timeout 120 python tools/probe.py "" 2>&1 | grep "ms/ingest" | cut -c1-120
  if [ "$2" = "0" ]; then timeout 120 python tools/launch_cost.py 2>&1 | grep "one batch per launch" | cut -c1-200; fi
done
