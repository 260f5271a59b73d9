"""Fold rocprofv3 CSV output (kernel trace + PMC passes) into small per-kernel summaries that are committed under profiles/."""
import csv, glob, json, os, sys
from collections import defaultdict

out_dir, tag = sys.argv[1], sys.argv[2]


def find(pattern):
    r = glob.glob(os.path.join(out_dir, pattern), recursive=True)
    return r[0] if r else None


def short(name):
    n = name.split("(")[0]
    n = n.replace("simlod::", "").replace("build::", "").replace("batch::", "").replace("bulk::", "").replace("void ", "")
    return n.strip()

summary = {}
seen_first = {}
trace = find("trace/**/*kernel_trace.csv")
if trace:
    agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    with open(trace) as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            d = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
            a = agg[k]; a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    summary["kernel_trace"] = {k: {"calls": a[0], "total_us": round(a[1], 1), "avg_us": round(a[1] / a[0], 2), "min_us": round(a[2], 2),
                                   "max_us": round(a[3], 2), "pct": round(100 * a[1] / total, 2)} for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])}

for name, counters in (("pmc_fetch", ["FETCH_SIZE"]), ("pmc_write", ["WRITE_SIZE"]), ("pmc_l2", ["TCC_HIT_sum", "TCC_MISS_sum"]),
                       ("pmc_sq", ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"]),
                       ("pmc_lds", ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS"]),
                       ("pmc_atomic", ["TCC_EA0_ATOMIC_sum", "TCC_ATOMIC_sum"]), ("pmc_atomic2", ["TCC_EA0_ATOMIC_LEVEL_sum", "TCC_EA0_RDREQ_sum"]),
                       ("pmc_fetch_close", ["FETCH_SIZE"]), ("pmc_write_close", ["WRITE_SIZE"]), ("pmc_atomic_close", ["TCC_EA0_ATOMIC_sum", "TCC_ATOMIC_sum"])):
    path = find(f"{name}/**/*counter_collection.csv")
    if not path:
        continue
    agg = defaultdict(lambda: defaultdict(float)); calls = defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            c = row["Counter_Name"]
            if c in counters:
                agg[k][c] += float(row["Counter_Value"])
                first = seen_first.setdefault((name, k), c)
                if c == first:
                    calls[k] += 1
    summary[name] = {k: {"dispatches": calls[k], **{c: v[c] for c in counters if c in v}} for k, v in agg.items()}

# HBM traffic per kernel, as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE counts 64 B per
# 128-B request of a wide coalesced stream (read side may be under-counted by up to 2x; scattered 4-16 B accesses are uncalibrated).
traffic = {}
if "pmc_fetch" in summary and "pmc_write" in summary:
    for k, v in summary["pmc_fetch"].items():
        w = summary["pmc_write"].get(k, {"WRITE_SIZE": 0.0})
        traffic[k] = {"dispatches": v["dispatches"], "fetch_bytes_raw": v["FETCH_SIZE"] * 1024, "fetch_bytes_x2": v["FETCH_SIZE"] * 2048,
                      "write_bytes": w["WRITE_SIZE"] * 1024}
summary["hbm_traffic"] = traffic
traffic_close = {}
if "pmc_fetch_close" in summary and "pmc_write_close" in summary:          # the raster kernels of the close-up preset
    for k, v in summary["pmc_fetch_close"].items():
        if k.startswith("r_"):
            w = summary["pmc_write_close"].get(k, {"WRITE_SIZE": 0.0})
            traffic_close[k] = {"dispatches": v["dispatches"], "fetch_bytes_raw": v["FETCH_SIZE"] * 1024, "fetch_bytes_x2": v["FETCH_SIZE"] * 2048, "write_bytes": w["WRITE_SIZE"] * 1024}
summary["hbm_traffic_close"] = traffic_close
# wave-level picture per kernel: share of wave cycles spent waiting (s_waitcnt / barriers), issuing, stalled at issue
if "pmc_sq" in summary:
    for k, v in summary["pmc_sq"].items():
        wc = v.get("SQ_WAVE_CYCLES", 0.0)
        if wc > 0:
            v["wait_any_frac"] = v.get("SQ_WAIT_ANY", 0.0) / wc
            v["wait_inst_frac"] = v.get("SQ_WAIT_INST_ANY", 0.0) / wc
            v["active_inst_frac"] = v.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
if "pmc_l2" in summary:
    for k, v in summary["pmc_l2"].items():
        t = v.get("TCC_HIT_sum", 0.0) + v.get("TCC_MISS_sum", 0.0)
        if t > 0:
            v["hit_rate"] = v.get("TCC_HIT_sum", 0.0) / t
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simlod_amd.fingerprint import csrc_sha16
summary["_csrc_sha16"] = csrc_sha16()          # the kernel sources these counters were collected on (bench.py quotes them only while it matches)
os.makedirs("gpurun_out", exist_ok=True)
with open(os.path.join("gpurun_out", f"profile_summary_{tag}.json"), "w") as f:
    json.dump(summary, f, indent=1)
print(json.dumps(summary.get("kernel_trace", {}), indent=0)[:3000])
