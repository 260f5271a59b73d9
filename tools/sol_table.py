"""Folds what tools/sol_table.sh collected into the speed-of-light table of the construct kernels: per kernel and ingest mode, over the launches WITH work of the last
ingest of the run (one stream: every kernel alone on the chip): launches, time per launch, HBM bytes per launch (FETCH_SIZE x 2 as MI355X_MICROARCH.md prescribes for
gfx950 + WRITE_SIZE, KiB), bytes / time, the time the measured device-to-device copy rate needs for those bytes, and the SQ wait fraction.

    python tools/sol_table.py gpurun_out/sol_<tag>"""
import collections, csv, glob, os, statistics, sys
root = sys.argv[1]
name = lambda s: s.split("(")[0].replace("simlod::build::", "").replace("simlod::", "").replace("void ", "")
copy = 5.0
try:
    for line in open(os.path.join(root, "copy_rate.txt")):
        if line.startswith("copy_rate_TBps"):
            copy = float(line.split()[1])
except OSError:
    pass
print(f"device-to-device copy rate measured in this run: {copy:.2f} TB/s (read + written bytes)")


def last(paths):
    return sorted(paths, key=os.path.getmtime)[-1] if paths else None


def trace_rows(d):
    p = last(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))
    rows = list(csv.DictReader(open(p)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


def counters(d):
    p = last(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))
    if p is None:
        return {}
    out = collections.defaultdict(lambda: collections.defaultdict(list))      # kernel -> counter -> per-dispatch values in dispatch order
    rows = list(csv.DictReader(open(p)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows:
        per[(int(r["Dispatch_Id"]), name(r["Kernel_Name"]))][r["Counter_Name"]] += float(r["Counter_Value"])
    for (did, k), v in sorted(per.items()):
        for c, x in v.items():
            out[k][c].append(x)
    return out


for mode in ("exact", "coalesced"):
    d = os.path.join(root, mode + "_trace")
    if not os.path.isdir(d):
        continue
    rows = trace_rows(d)
    resets = [i for i, r in enumerate(rows) if name(r["Kernel_Name"]) == "k_reset"]
    seg = rows[resets[-1]:]
    t0, t1 = int(seg[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in seg if name(r["Kernel_Name"]).startswith("k_"))
    dur = collections.defaultdict(list)
    for r in seg:
        k = name(r["Kernel_Name"])
        if k.startswith("k_"):
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    fetch, write, sq = counters(os.path.join(root, mode + "_fetch")), counters(os.path.join(root, mode + "_write")), counters(os.path.join(root, mode + "_sq"))
    print(f"\n== {mode} mode, one stream (every kernel alone on the chip); last ingest of the run: {(t1 - t0) / 1e3:.0f} us from k_reset to the last kernel's end")
    print("kernel                   launches  with work   us/launch   MB/launch (fetch x2 + write)   TB/s   us at copy rate   x off   SQ wait / wave cycles")
    tot_t = tot_sol = 0.0
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        med = statistics.median(v)
        work = [x for x in v if x >= max(med / 3, 6.0)] if max(v) > 12 else v
        if not work:
            continue
        # bytes of the launches with work: the counter passes ran the same command; a launch without work moves next to nothing, so the kernel's total / launches with work
        n_all = len(fetch.get(k, {}).get("FETCH_SIZE", []))
        runs = max(1, round(n_all / max(1, len(v))))      # ingests in the counter run per ingest here (the probe runs warm-up + steps)
        fb = sum(fetch.get(k, {}).get("FETCH_SIZE", [])) * 1024 * 2 / runs
        wb = sum(write.get(k, {}).get("WRITE_SIZE", [])) * 1024 / runs
        per = (fb + wb) / len(work)
        t = sum(work) / len(work)
        sol = per / (copy * 1e12) * 1e6
        s = sq.get(k, {})
        wait = sum(s.get("SQ_WAIT_ANY", [])) / max(1.0, sum(s.get("SQ_WAVE_CYCLES", [])))
        print("%-24s %8d  %9d   %9.1f   %12.1f                   %5.2f   %12.1f   %6.1f   %8.2f" % (k, len(v), len(work), t, per / 1e6, per / t / 1e6, sol, t / max(sol, 1e-9), wait))
        tot_t += sum(work); tot_sol += sol * len(work)
    print("sum over the launches with work: %.0f us of kernels, %.0f us at the copy rate" % (tot_t, tot_sol))
