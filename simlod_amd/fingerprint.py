"""A fingerprint of the kernel sources: what a kept measurement (profiles/*.json) was taken on.  bench.py quotes numbers from profiles/
only while this still matches, so that a stale profile cannot dress up a fresh run."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha16():
    files = sorted(glob.glob(os.path.join(ROOT, "simlod_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "simlod_amd", "csrc", "*.hpp")) +
                   glob.glob(os.path.join(ROOT, "simlod_amd", "csrc", "*.cpp")) + glob.glob(os.path.join(ROOT, "include", "*.h")))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
