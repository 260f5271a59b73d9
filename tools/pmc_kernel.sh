#!/bin/bash
# usage: tools/pmc_kernel.sh "<counter list>" <tag>   (runs tools/probe.py on the 36 M terrain under rocprofv3 --pmc, prints per-kernel sums)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_$2; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc $1 --output-format csv -d $OUT -- python $REPO/tools/probe.py --steps 2 "" > /dev/null 2> $OUT/err.txt
cd $REPO
python - "$OUT" <<'PY'
import csv, glob, sys, collections
p = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(p)):
    k = r["Kernel_Name"].split("(")[0].replace("simlod::build::", "").replace("simlod::", "").replace("void ", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    if k.startswith(("k_voxelize", "k_count", "k_queue", "k_hist", "k_insert", "k_expand", "r_draw")):
        print(k, {c: round(x) for c, x in v.items()})
PY
find $OUT -name "*.csv" -size +2M -delete
