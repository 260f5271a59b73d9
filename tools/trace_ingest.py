"""Every builder kernel of the LAST ingest in a rocprofv3 kernel trace, in start order: queue, start, end, duration (us from k_reset's start).
    python tools/trace_ingest.py <trace dir>"""
import csv, glob, os, sys
paths = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(paths[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("simlod::build::", "").replace("simlod::", "").replace("void ", "")
resets = [i for i, r in enumerate(rows) if name(r) == "k_reset"]
seg = rows[resets[-1]:]
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    if not name(r).startswith("k_"):
        continue
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("%-22s q%-3s %8.1f %8.1f  %6.1f" % (name(r), r.get("Queue_Id", "?"), s, e, e - s))
