#!/bin/bash
# usage (GPU box): tools/raster_trace.sh <tag> <preset>      rocprofv3 kernel trace of the rasteriser alone, one camera preset; per-kernel averages
TAG=$1; PRESET=${2:-close}
REPO=$(pwd); OUT=$REPO/gpurun_out/rtrace_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
PRESETS=$PRESET rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $REPO/tools/raster_close.py 20 > $OUT/out.txt 2> $OUT/trace.err
cd $REPO
python - "$OUT" <<'PY'
import csv, glob, sys, collections
p = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
d = collections.defaultdict(list)
for r in csv.DictReader(open(p)):
    k = r["Kernel_Name"].split("(")[0].replace("simlod::", "").replace("void ", "")
    if k.startswith("r_"): d[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v = v[len(v) // 4:]
    print("%-22s calls %4d  avg %7.1f us  min %7.1f  max %7.1f" % (k, len(v), sum(v) / len(v), min(v), max(v)))
PY
grep -v amdgpu $OUT/out.txt
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete 2>/dev/null
