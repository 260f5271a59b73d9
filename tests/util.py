"""Comparison helpers shared by the parity tests (H6-aware: see SURVEY.md §2.5)."""
import numpy as np

import oracle
from simlod_amd import abi

# fields of oracle.dump_dtype that every correct implementation must reproduce exactly
EXACT_FIELDS = ["key", "level", "X", "Y", "Z", "isLeaf", "childMask", "counter", "numPoints", "numVoxels", "numVoxelsStored",
                "countIteration", "hasGrid", "gridPopcount", "gridHash", "pointsSum", "pointsXor", "voxelPosSum", "voxelPosXor",
                "pointChunks", "voxelChunks", "name"]

# Stats fields that are deterministic for a fixed batch sequence (allocatedBytes_momentary is implementation defined)
STATS_BUILD_FIELDS = ["numNodes", "numInner", "numLeaves", "numNonemptyLeaves", "numPoints", "numVoxels", "allocatedBytes_persistent",
                      "numChunksPoints", "numChunksVoxels", "batchletIndex", "numPointsProcessed", "numAllocatedChunks", "chunkPoolSize",
                      "memCapacityReached"]
STATS_RENDER_FIELDS = ["numVisibleNodes", "numVisibleInner", "numVisibleLeaves", "numVisiblePoints", "numVisibleVoxels"]


def assert_dumps_equal(a, b, what=""):
    assert len(a) == len(b), f"{what}: node count {len(a)} != {len(b)}"
    for f in EXACT_FIELDS:
        if not np.array_equal(a[f], b[f]):
            bad = np.nonzero(np.any(np.atleast_2d((a[f] != b[f]).reshape(len(a), -1)), axis=1))[0]
            i = int(bad[0])
            raise AssertionError(f"{what}: field {f} differs at {len(bad)} nodes; first: level={a['level'][i]} "
                                 f"XYZ=({a['X'][i]},{a['Y'][i]},{a['Z'][i]}) {a[f][i]} != {b[f][i]}")


def assert_stats_equal(a, b, fields, what=""):
    for f in fields:
        assert int(a[f]) == int(b[f]), f"{what}: Stats.{f} {int(a[f])} != {int(b[f])}"


def host_image_of(dev):
    """Download a DeviceOctree's image and rebase its pointers to the host copies."""
    nodes, pers, n, nodes_base, pers_base = dev.download_image()
    oracle.rebase_image(nodes, n, pers, nodes_base, pers_base)
    return nodes, pers, n


def voxel_colors_are_member(nodes, n, points, box_size, max_level=20):
    """Every voxel's colour must be the colour of SOME input point that falls into the voxel's cell (first-writer-wins is
    scheduling dependent).  Returns the number of voxels checked.  Nodes deeper than `max_level` are skipped:
    their cells are smaller than an fp32 position can resolve, so the cell cannot be recovered from the stored voxel position."""
    size = np.float32(max(box_size))
    X = (np.float32(2 ** 20) * points["x"] / size).astype(np.uint32)
    Y = (np.float32(2 ** 20) * points["y"] / size).astype(np.uint32)
    Z = (np.float32(2 ** 20) * points["z"] / size).astype(np.uint32)
    pX = (np.float32(2 ** 28) * points["x"] / size).astype(np.uint32)
    pY = (np.float32(2 ** 28) * points["y"] / size).astype(np.uint32)
    pZ = (np.float32(2 ** 28) * points["z"] / size).astype(np.uint32)
    checked = 0
    for i in range(n):
        nd = nodes[i]
        nv = int(nd["numVoxelsStored"])
        if nv == 0:
            continue
        lvl = int(nd["level"])
        if lvl > max_level:
            continue
        vox = oracle.gather_samples(int(nd["voxelChunks"]), nv)
        sel = ((X >> (20 - lvl)) == nd["X"]) & ((Y >> (20 - lvl)) == nd["Y"]) & ((Z >> (20 - lvl)) == nd["Z"]) if lvl > 0 else np.ones(len(points), bool)
        sh = 21 - lvl
        cell = ((pX[sel] >> sh) & 127).astype(np.uint64) | (((pY[sel] >> sh) & 127).astype(np.uint64) << 7) | (((pZ[sel] >> sh) & 127).astype(np.uint64) << 14)
        have = np.unique((cell << np.uint64(32)) | points["color"][sel].astype(np.uint64))
        node_size = size / np.float32(2.0 ** lvl)
        mn = np.array([nd["X"], nd["Y"], nd["Z"]], dtype=np.float32) * node_size
        vc = [np.floor((vox[a] - mn[k]) / node_size * np.float32(128.0)).astype(np.int64).clip(0, 127).astype(np.uint64) for k, a in enumerate("xyz")]
        vkey = ((vc[0] | (vc[1] << np.uint64(7)) | (vc[2] << np.uint64(14))) << np.uint64(32)) | vox["color"].astype(np.uint64)
        pos = np.searchsorted(have, vkey)
        ok = (pos < len(have)) & (have[np.minimum(pos, len(have) - 1)] == vkey)
        assert ok.all(), f"node level={lvl} XYZ=({nd['X']},{nd['Y']},{nd['Z']}): {int((~ok).sum())} voxels carry a colour no point of their cell has"
        checked += nv
    return checked


def points_multiset_hash(pts):
    """(sum, xor) order-independent hash of a set of 16-byte points — the same mixer as oracle_dump's pointsSum / pointsXor."""
    w = pts.view(np.uint32).reshape(-1, 4).astype(np.uint64)

    def mix(x):
        x = x ^ (x >> np.uint64(30)); x = x * np.uint64(0xbf58476d1ce4e5b9); x = x ^ (x >> np.uint64(27)); x = x * np.uint64(0x94d049bb133111eb)
        return x ^ (x >> np.uint64(31))
    with np.errstate(over="ignore"):
        a = (w[:, 0] << np.uint64(32)) | w[:, 1]
        b = (w[:, 2] << np.uint64(32)) | w[:, 3]
        h = mix(a ^ mix(b + np.uint64(0x9e3779b97f4a7c15)))
        return np.uint64(h.sum()), np.bitwise_xor.reduce(mix(h + np.uint64(1)))
