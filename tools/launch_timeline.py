"""Kernel timeline of ONE kernel_construct launch out of a rocprofv3 kernel trace: every kernel from a k_begin to the k_finish behind it, start and
end relative to the k_begin, with its queue — where a one-batch launch's time goes.

    python tools/launch_timeline.py <trace dir> [which k_begin, default 20] [how many launches, default 1]"""
import csv, glob, os, sys
root = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 20
many = int(sys.argv[3]) if len(sys.argv) > 3 else 1
paths = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(paths[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda r: r["Kernel_Name"].split("(")[0].replace("simlod::build::", "").replace("void ", "")
begins = [i for i, r in enumerate(rows) if short(r).startswith("k_begin")]
for w in range(which, min(which + many, len(begins) - 1)):
    i0, i1 = begins[w], begins[w + 1]
    t0 = int(rows[i0]["Start_Timestamp"])
    prev_end = {}
    for r in rows[i0:i1]:
        q = r.get("Queue_Id", "?")
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        gap = s - prev_end[q] if q in prev_end else 0.0
        prev_end[q] = e
        print("%-22s q%-3s start %7.1f us  end %7.1f us  (%5.1f us)  gap on its queue %5.1f" % (short(r)[:22], q, s, e, e - s, gap))
    print("launch %d: %.1f us from k_begin's start to the last kernel's end; next k_begin starts at %.1f" % (w, max(prev_end.values()) if prev_end else 0.0, (int(rows[i1]["Start_Timestamp"]) - t0) / 1e3))
