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

# the last ingest: start-to-start of consecutive k_count launches (the per-batch cycle), with the kernel times of that batch
resets = [i for i, r in enumerate(rows) if name(r) == "k_reset"]
if resets:
    seg = rows[resets[-1]:]
    cs = [r for r in seg if name(r).startswith("k_count")]
    starts = [int(r["Start_Timestamp"]) for r in cs]
    cyc = [(starts[i + 1] - starts[i]) / 1e3 for i in range(len(starts) - 1)]
    per = lambda k: [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in seg if name(r).split("<")[0] == k]
    print("last ingest: %.0f us from k_reset to the last kernel's end" % ((max(int(r["End_Timestamp"]) for r in seg) - int(seg[0]["Start_Timestamp"])) / 1e3))
    print("cycle  ", [round(c) for c in cyc])
    for k in ("k_count", "k_hist", "k_expand", "k_hist2", "k_insert", "k_voxelize"):      # (k_expand: a launch that takes groups has two instances per group, rounds (0, 1) and (1, MAX), with k_hist2 between them)
        print("%-8s" % k[2:], per(k))
    act = sorted(c for c in cyc if c > 30)
    if act:
        print("median active cycle %.1f us, mean %.1f us over %d" % (act[len(act) // 2], sum(act) / len(act), len(act)))
