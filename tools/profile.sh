#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + the PMC passes behind the HBM-traffic, L2, wave and atomic figures.
# Counters are collected in their own runs (--kernel-trace only), one block's worth per pass (MI355X_MICROARCH.md: TCC has 4 slots,
# FETCH_SIZE costs 3, WRITE_SIZE 2; SQ has 8).
# Usage: tools/profile.sh <tag> [bench flags]      outputs land in gpurun_out/prof_<tag>/, the folded summary in gpurun_out/profile_summary_<tag>.json
set -u
TAG=${1:-r03}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 2 --warmup 1 --frames 4 --no-cpu-baseline --no-profile --raster-presets bird $*"
cd /tmp
rocprofv3 -L > $OUT/counters_available.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace_bench.json 2> $OUT/trace.err
pass() {   # name, counters...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- $CMD > /dev/null 2> $OUT/$name.err || echo "pass $name failed" >> $OUT/failed.txt
}
if [ "${PMC:-1}" = 1 ]; then      # PMC=0: the kernel trace alone
pass pmc_fetch FETCH_SIZE
pass pmc_write WRITE_SIZE
pass pmc_l2 TCC_HIT_sum TCC_MISS_sum
pass pmc_sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass pmc_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS
pass pmc_atomic TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum
pass pmc_atomic2 TCC_EA0_ATOMIC_LEVEL_sum TCC_EA0_RDREQ_sum
# the close-up camera preset (most samples of the near leaves leave their tiles): the byte passes of the raster kernels alone
CMDC="python $REPO/bench.py --steps 1 --warmup 0 --frames 4 --no-cpu-baseline --no-profile --raster-presets close"
passc() { local name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- $CMDC > /dev/null 2> $OUT/$name.err || echo "pass $name failed" >> $OUT/failed.txt; }
passc pmc_fetch_close FETCH_SIZE
passc pmc_write_close WRITE_SIZE
passc pmc_atomic_close TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum
fi
cd $REPO
python tools/summarize_profile.py $OUT $TAG
find $OUT -name "*.csv" -size +3M -delete
find $OUT -name "*.db" -delete 2>/dev/null
ls $OUT | head -40
