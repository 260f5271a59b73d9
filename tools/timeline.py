"""Kernel timeline of one mid-ingest batch from a rocprofv3 kernel trace (gpurun_out/prof_<tag>/trace/**/kernel_trace.csv):
start and end of every kernel relative to the batch's k_count, with its stream/queue — shows what overlaps what."""
import csv, glob, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 60          # which k_count to start at
paths = sorted(glob.glob(os.path.join("gpurun_out", f"prof_{tag}", "trace", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(paths[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
counts = [i for i, r in enumerate(rows) if "k_count" in r["Kernel_Name"]]
i0, i1 = counts[skip], counts[skip + 1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1 + 1]:
    name = r["Kernel_Name"].split("(")[0].replace("simlod::build::", "").replace("void ", "")
    print("%-14s q%-3s start %7.1f us  end %7.1f us  (%5.1f us)" % (name, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                                               (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
