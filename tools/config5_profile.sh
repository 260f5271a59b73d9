#!/bin/bash
# usage (GPU box): tools/config5_profile.sh <tag> [points]   BASELINE config 5 (all points in one level-6 cell) under rocprofv3: kernel trace (per-kernel totals
# of the ingest) + one PMC pass (memory-side atomics); output gpurun_out/config5_<tag>/{kernels.txt,atomics.txt,run.json}
TAG=${1:-r05}; N=${2:-200000000}
REPO=$(pwd); OUT=$REPO/gpurun_out/config5_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $REPO/tools/hotspot_ab.py $N > $OUT/run.json 2> $OUT/trace.err
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_ATOMIC_sum TCC_EA0_ATOMIC_LEVEL_sum --output-format csv -d $OUT/pmc -- python $REPO/tools/hotspot_ab.py $N > /dev/null 2> $OUT/pmc.err
cd $REPO
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
p = sorted(glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True))[-1]
d = collections.defaultdict(list)
for r in csv.DictReader(open(p)):
    k = r["Kernel_Name"].split("(")[0].replace("simlod::build::", "").replace("simlod::", "").replace("void ", "")
    if k.startswith("k_"): d[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in d.values())
with open(out + "/kernels.txt", "w") as f:
    f.write("kernel                      calls   total ms   share   avg us   max us\n")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        f.write("%-26s %6d  %9.2f  %5.1f %%  %7.1f  %7.1f\n" % (k, len(v), sum(v) / 1e3, 100 * sum(v) / tot, sum(v) / len(v), max(v)))
    f.write("sum of kernel times %.1f ms (two streams: the wall time is shorter)\n" % (tot / 1e3))
pm = sorted(glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True))
if pm:
    a = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(pm[-1])):
        k = r["Kernel_Name"].split("(")[0].replace("simlod::build::", "").replace("simlod::", "").replace("void ", "")
        if k.startswith("k_"): a[k][r["Counter_Name"]] += float(r["Counter_Value"])
    with open(out + "/atomics.txt", "w") as f:
        f.write("kernel                      memory-side atomics   cycles in flight per atomic (TCC_EA0_ATOMIC_LEVEL / TCC_EA0_ATOMIC)\n")
        for k, c in sorted(a.items(), key=lambda kv: -kv[1].get("TCC_EA0_ATOMIC_sum", 0)):
            n = c.get("TCC_EA0_ATOMIC_sum", 0.0)
            f.write("%-26s %14.0f   %10.0f\n" % (k, n, c.get("TCC_EA0_ATOMIC_LEVEL_sum", 0.0) / n if n else 0))
PY
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete 2>/dev/null
cat $OUT/kernels.txt $OUT/atomics.txt; head -c 400 $OUT/run.json
