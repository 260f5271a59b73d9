#!/usr/bin/env python3
"""Generates tests/golden/*.npz from oracle/_ref — the reference's OWN sources (modules/progressive_octree/{reset,
progressive_octree_voxels,render}.cu) compiled in place as host code (oracle/Makefile, `make -C oracle ref`).
Needs /root/reference at build time; run from the repo root:  python tests/golden/make_golden.py

Each fixture holds, for one seeded case of tests/cases.py: the canonical octree dump (oracle.dump_dtype, sorted by
(level,X,Y,Z)), the deterministic Stats fields after construct and after render, and SHA-256 + non-background pixel
count of the pre-EDL uint64 framebuffer for plain and HQS rendering."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import oracle  # noqa: E402
from cases import CASES, batches_of, case, uniforms_for  # noqa: E402
from simlod_amd import abi  # noqa: E402

for name in CASES:
    pts, box, batch, T = case(name)
    u = uniforms_for(box, T)
    o = oracle.HostOctree("ref", persistent_bytes=1 << 30, ring_slots=8)
    o.reset(u)
    for b in batches_of(name, pts, batch):
        o.upload(b)
    while int(o.stats["batchletIndex"][0]) < int(o.num_uploaded[0]):
        o.construct(u)
    build_stats = o.stats.copy()
    dump = o.dump()
    out = {"dump": dump, "build_stats": build_stats}
    for mode, hqs in (("plain", False), ("hqs", True)):
        uu = uniforms_for(box, T, hqs=hqs)
        fb, _ = o.render(uu)
        out[f"fb_sha256_{mode}"] = np.frombuffer(hashlib.sha256(fb.tobytes()).digest(), dtype=np.uint8)
        out[f"fb_nonbg_{mode}"] = np.array([int((fb != abi.CLEAR_PIXEL).sum())])
        out[f"render_stats_{mode}"] = o.stats.copy()
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    s = build_stats[0]
    print(name, {k: int(s[k]) for k in ("numNodes", "numPoints", "numVoxels", "allocatedBytes_persistent", "chunkPoolSize")},
          "nonbg", int(out["fb_nonbg_plain"][0]), int(out["fb_nonbg_hqs"][0]))
