"""Mints the loader-row fixtures FROM THE REFERENCE ITSELF.  Runs only where /root/reference exists:

  * las_decode.npz  — for LAS point formats 0,1,2,3,5,6,7,8: raw records + header numbers of a small synthetic file and the
                      Points the reference's own loadLasNative (LasLoader.cpp:169-227, built in place as
                      oracle/_ref/libref_las.so) decodes from it, plus the reference's loadHeader fields.
  * tiny_f2.las     — 3000-point synthetic terrain, LAS 1.2 format 2 (written by simlod_amd.lasio.points_to_las)
  * tiny_f2.simlod  — what the reference's tools/las2simlod.mjs makes of tiny_f2.las, run with node; its two hard-coded
                      path lines are substituted in a temporary copy at run time, the tool itself is not stored here.

    python tests/golden/make_golden_las.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle                                   # noqa: E402
from simlod_amd import lasio, synthetic         # noqa: E402

REF_TOOL = "/root/reference/tools/las2simlod.mjs"
CASES = [(0, (1, 2), None), (1, (1, 2), None), (2, (1, 2), None), (3, (1, 3), None), (5, (1, 3), None), (6, (1, 4), None),
         (7, (1, 4), None), (8, (1, 4), None), (2, (1, 2), 29), (3, (1, 4), 41)]     # last two: extra bytes, odd strides


def main():
    oracle.build(ref=True)
    out = {}
    tmp = tempfile.mkdtemp()
    rs = np.random.RandomState(2024)
    for ci, (fmt, ver, bpp) in enumerate(CASES):
        n = 400
        xyz = rs.randint(-2 ** 31, 2 ** 31 - 1, size=(n, 3), dtype=np.int64).astype(np.int32)
        xyz[:4] = [[0, 0, 0], [2 ** 31 - 1] * 3, [-2 ** 31] * 3, [1, -1, 255]]
        rgb = rs.randint(0, 65536, size=(n, 3)).astype(np.uint16)
        rgb[:100] = rs.randint(0, 300, size=(100, 3))           # around the c > 255 switch
        rgb[100:103] = [[255, 256, 257], [0, 65535, 511], [512, 255, 254]]
        rec = lasio.las_records(xyz, rgb, fmt, bytes_per_point=bpp, seed=100 + ci)
        scale = (0.001, 0.01, 0.00025)
        offset = (123456.789, -98765.4321, 12.5)
        mins, maxs = (100000.0, -100000.0, 0.0), (200000.0, 0.0, 100.0)
        path = os.path.join(tmp, f"c{ci}.las")
        lasio.write_las(path, rec, fmt, scale, offset, mins, maxs, version=ver, vlr_bytes=54 + ci)
        rh = oracle.ref_las_header(path)
        first, count = 7, n - 20
        tr = tuple(-m for m in rh["min"])
        pts = oracle.ref_las_load(path, first, count, tr)
        out[f"c{ci}_fmt"] = np.array([fmt, ver[0], ver[1], rec.shape[1], first, count], dtype=np.int64)
        out[f"c{ci}_file"] = np.fromfile(path, dtype=np.uint8)
        out[f"c{ci}_header"] = np.array([rh["versionMajor"], rh["versionMinor"], rh["headerSize"], rh["offsetToPointData"], rh["format"],
                                         rh["bytesPerPoint"], rh["numPoints"]], dtype=np.int64)
        out[f"c{ci}_header_f"] = np.array(rh["scale"] + rh["offset"] + rh["min"] + rh["max"], dtype=np.float64)
        out[f"c{ci}_points"] = pts
    np.savez_compressed(os.path.join(HERE, "las_decode.npz"), **out)

    pts, box = synthetic.terrain(3000, seed=5, box=(60.0, 40.0, 4.0), tile=10.0)
    las = os.path.join(HERE, "tiny_f2.las")
    lasio.points_to_las(las, pts, box, fmt=2, scale=0.001, world_min=(694000.0, 3915000.0, -3.0))
    src = open(REF_TOOL).read().splitlines()
    patched = [("let file = process.argv[2];" if l.startswith("let file =") else "let outPath = process.argv[3];" if l.startswith("let outPath =") else l) for l in src]
    tool = os.path.join(tmp, "las2simlod_run.mjs")
    open(tool, "w").write("\n".join(patched))
    subprocess.check_call(["node", tool, las, os.path.join(HERE, "tiny_f2.simlod")], stdout=subprocess.DEVNULL)
    print("wrote las_decode.npz, tiny_f2.las, tiny_f2.simlod")


if __name__ == "__main__":
    main()
