"""Per-kernel averages and one mid-ingest timeline from a rocprofv3 kernel trace of tools/probe.py (or bench.py).

    python tools/trace_summary.py <trace dir> [first k_count to print from] [how many batches]

Averages count ACTIVE launches only (a chain enqueues kernels for batches that do not exist; those exit in a few us: anything
shorter than a third of the kernel's median is dropped from its average)."""
import csv, glob, os, sys, statistics, collections
d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 60
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 2
paths = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(paths[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("simlod::build::", "").replace("simlod::", "").replace("void ", "")
dur = collections.defaultdict(list)
for r in rows:
    dur[name(r)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("kernel                     calls  active   avg us   max us")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    if not k.startswith(("k_", "r_")):
        continue
    med = statistics.median(v)
    act = [x for x in v if x >= med / 3] if med > 8 else v
    print("%-26s %5d  %5d  %7.1f  %7.1f" % (k, len(v), len(act), sum(act) / len(act), max(v)))
counts = [i for i, r in enumerate(rows) if "k_count" in r["Kernel_Name"]]
if len(counts) > skip + nb:
    i0, i1 = counts[skip], counts[skip + nb]
    t0 = int(rows[i0]["Start_Timestamp"])
    for r in rows[i0:i1 + 1]:
        print("%-22s q%-3s start %7.1f us  end %7.1f us  (%5.1f us)" % (name(r), r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                                                   (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
