"""A fingerprint of the kernel sources: what a kept measurement (profiles/*.json) was taken on.  bench.py quotes numbers from profiles/
only while this still matches, so that a stale profile cannot dress up a fresh run."""
import glob
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha16(root=ROOT):
    files = sorted(glob.glob(os.path.join(root, "simlod_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "simlod_amd", "csrc", "*.hpp")) +
                   glob.glob(os.path.join(root, "simlod_amd", "csrc", "*.cpp")) + glob.glob(os.path.join(root, "simlod_amd", "csrc", "*.inc")) + glob.glob(os.path.join(root, "include", "*.h")))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def kept_profiles(requested=None, root=ROOT):
    """Which profiles/<round>/ directory bench.py may quote from: the requested one, else the newest profiles/r*/ — and only if its
    fingerprint.json names the kernel sources of this tree.  Returns (directory or None, traffic file or None, note, sha of this tree)."""
    sha_now = csrc_sha16(root)
    dirs = [os.path.join(root, "profiles", requested)] if requested else sorted(glob.glob(os.path.join(root, "profiles", "r[0-9]*")), reverse=True)
    pdir, note = None, "no profiles/r*/ directory"
    for d in dirs[:1]:                                   # the newest (or the requested) one decides: an older matching one is not dug out
        fp = os.path.join(d, "fingerprint.json")
        sha = json.load(open(fp)).get("_csrc_sha16") if os.path.exists(fp) else None
        if sha == sha_now:
            pdir, note = d, f"{os.path.relpath(d, root)}: measured on these kernel sources (csrc sha {sha_now})"
        else:
            note = f"{os.path.relpath(d, root)} was measured on other kernel sources (csrc sha {sha} != {sha_now}): nothing quoted from it"
    tfile = os.path.join(root, "profiles", "traffic_" + os.path.basename(pdir) + ".json") if pdir else None
    if tfile and not (os.path.exists(tfile) and json.load(open(tfile)).get("_csrc_sha16") == sha_now):
        tfile = None
    return pdir, tfile, note, sha_now
