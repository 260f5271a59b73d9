"""Copy what tools/profile.sh collected (gpurun_out/) into profiles/<round>/ and fold the two HBM-byte passes into profiles/traffic_<round>.json,
the per-launch figures bench.py quotes as `roofline.traffic`.      usage: python tools/fold_profiles.py r02 [n_ingests n_batches frames_per_mode]"""
import glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
n_ingests, n_batches, frames = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (3, 36, 6)   # bench.py --steps 2 --warmup 1 --frames 4
dst = os.path.join(ROOT, "profiles", tag)
os.makedirs(dst, exist_ok=True)
for mode, suffix in (("", ""), ("_coalesced", "_coalesced")):
    src = os.path.join(ROOT, "gpurun_out", f"profile_summary_{tag}{mode}.json")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(dst, f"rocprofv3_summary{suffix}.json"))
    stats = glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}{mode}", "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:                                  # gpurun merges into gpurun_out/: older runs may still lie there
        stats.sort(key=os.path.getmtime)
        shutil.copy(stats[-1], os.path.join(dst, f"kernel_stats{suffix}.csv"))
summ = json.load(open(os.path.join(dst, "rocprofv3_summary.json")))
per_batch = {}                                 # every kernel of the chain runs once per batch
out = {"_comment": "HBM bytes per ACTIVE launch of the ingest kernels (a launch that had a 1 M-point batch to process: %d ingests x %d batches; the chain also "
                   "launches early-exiting instances, which move nothing) and per frame for the draw kernels (%d frames per mode), from two separate "
                   "rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --steps 2 --warmup 1 --frames 4`.  FETCH_SIZE doubled as MI355X_MICROARCH.md "
                   "prescribes for gfx950 (64 B tallied per 128-B request; calibrated for wide coalesced streams only, scattered 4-16 B accesses are "
                   "uncalibrated)." % (n_ingests, n_batches, frames),
       "_source": f"profiles/{tag}/rocprofv3_summary.json (tools/profile.sh {tag}; tools/fold_profiles.py)",
       "_csrc_sha16": summ.get("_csrc_sha16")}
names = {"r_draw<0>": "r_draw<MODE_MIN64>", "r_draw<1>": "r_draw<MODE_DEPTH>", "r_draw<2>": "r_draw<MODE_COLOR>",
         "r_overflow<0>": "r_overflow<MODE_MIN64>", "r_overflow<1>": "r_overflow<MODE_DEPTH>", "r_overflow<2>": "r_overflow<MODE_COLOR>"}
json.dump({"_csrc_sha16": summ.get("_csrc_sha16"), "what": "simlod_amd.fingerprint.csrc_sha16() of the sources the files of this directory were measured on"}, open(os.path.join(dst, "fingerprint.json"), "w"))
for k, v in summ["hbm_traffic"].items():
    total = v["fetch_bytes_x2"] + v["write_bytes"]
    base = k.split("<")[0]
    if k.startswith("k_") and base not in ("k_begin", "k_finish", "k_stats", "k_parents", "k_rebuild", "k_voxroot", "k_paths", "k_reset", "k_end", "k_voxdone"):
        out[base] = total / (n_ingests * n_batches * per_batch.get(base, 1))
    elif k in names:
        out[names[k]] = total / frames
    elif k in ("r_output<true>", "r_output<false>", "r_resolve"):      # one mode's frames only (<true>: HQS, resolve fused in)
        out[k] = total / frames
    elif k == "r_visible":                                              # both modes' frames (it clears 8 B/px for a plain frame, 36 B/px for an HQS one: the mean)
        out[k] = total / (frames * 2)
for k, v in summ.get("hbm_traffic_close", {}).items():                  # the close-up preset's frames (bench.py --raster-presets close --frames 4: 2 + 4 frames per mode)
    total = v["fetch_bytes_x2"] + v["write_bytes"]
    if k in names:
        out["close/" + names[k]] = total / frames
    elif k in ("r_output<true>", "r_output<false>", "r_resolve"):
        out["close/" + k] = total / frames
    elif k == "r_visible":
        out["close/" + k] = total / (frames * 2)
json.dump(out, open(os.path.join(ROOT, "profiles", f"traffic_{tag}.json"), "w"), indent=1)
print(json.dumps({k: round(v / 1e6, 2) for k, v in out.items() if not k.startswith("_")}), "MB")
