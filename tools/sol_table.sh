#!/bin/bash
# usage (GPU box, through gpurun): tools/sol_table.sh <tag>
# The speed-of-light table of the construct kernels: every kernel ALONE on the chip (SIMLOD_OVERLAP_TAIL=0: one stream, nothing beside it)
# in exact mode (groups of the default size) and in coalesced mode (groups of ten batches): rocprofv3 kernel trace + one FETCH_SIZE and one WRITE_SIZE pass of the
# same command, folded by tools/sol_table.py into time per launch, own HBM bytes per launch, bytes / time, and the time the device's copy rate would need for those bytes.
TAG=${1:-r06}
REPO=$(pwd); OUT=$REPO/gpurun_out/sol_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
python - > $OUT/copy_rate.txt <<'PY'
import torch, time
a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda:0"); b = torch.empty_like(a)
for _ in range(3): b.copy_(a)
torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): b.copy_(a)
e.record(); torch.cuda.synchronize()
print("copy_rate_TBps", 2 * 10 * (1 << 30) / (s.elapsed_time(e) * 1e-3) / 1e12)
PY
for mode in exact coalesced; do
  FLAGS=""; [ $mode = coalesced ] && FLAGS="--coalesce"
  for pass in trace fetch write sq; do
    case $pass in trace) PMC="";; fetch) PMC="--pmc FETCH_SIZE";; write) PMC="--pmc WRITE_SIZE";; sq) PMC="--pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY";; esac
    SIMLOD_OVERLAP_TAIL=0 timeout 600 rocprofv3 --kernel-trace $PMC --output-format csv -d $OUT/${mode}_$pass -- python $REPO/tools/probe.py --steps 2 $FLAGS "" > $OUT/${mode}_$pass.txt 2> $OUT/${mode}_$pass.err
  done
done
cd $REPO
python tools/sol_table.py $OUT > $OUT/sol_table.txt 2>&1
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete 2>/dev/null
cat $OUT/sol_table.txt
