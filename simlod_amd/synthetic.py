"""Synthetic XYZRGBA point sets shaped like BASELINE.json's configs (there is no network for the real files).

The record layout is the .simlod one (tools/las2simlod.mjs:141-147): 3 x float32 position + RGBA8, 16 bytes,
positions already translated so that the bounding box starts at the origin (main_progressive_octree.cpp:312-313,
:868).  Every generator returns (points[abi.point_dtype], box_size(3,) float32).
"""
import numpy as np

from . import abi


def _pack(x, y, z, r, g, b):
    pts = np.empty(len(x), dtype=abi.point_dtype)
    pts["x"], pts["y"], pts["z"] = x, y, z
    pts["color"] = (r.astype(np.uint32) | (g.astype(np.uint32) << 8) | (b.astype(np.uint32) << 16) | np.uint32(255 << 24))
    return pts


def uniform_cube(n=1_000_000, seed=1234):
    """BASELINE config 1: x,y,z ~ U[0,1) drawn in that order per point from std::mt19937(seed)
    through std::uniform_real_distribution<float> (= float(u32) / 2^32, clamped below 1); colour = floor(255*xyz)."""
    rs = np.random.RandomState(seed)
    raw = np.frombuffer(rs.bytes(4 * 3 * n), dtype=np.uint32)
    v = raw.astype(np.float32) / np.float32(4294967296.0)
    v = np.minimum(v, np.nextafter(np.float32(1.0), np.float32(0.0))).reshape(n, 3)
    c = np.floor(v * np.float32(255.0)).astype(np.uint32)
    return _pack(v[:, 0], v[:, 1], v[:, 2], c[:, 0], c[:, 1], c[:, 2]), np.array([1, 1, 1], dtype=np.float32)


def _height(x, y, rng, box):
    """Band-limited fractal height field in [0, box.z): 7 octaves of rotated sine products."""
    h = np.zeros_like(x, dtype=np.float32)
    amp, norm = 1.0, 0.0
    for o in range(7):
        f = (2.0 ** o) * 2.0 * np.pi / box[0]
        a = rng.uniform(0, 2 * np.pi)
        ph1, ph2 = rng.uniform(0, 2 * np.pi, size=2)
        u = (np.cos(a) * x + np.sin(a) * y).astype(np.float32)
        v = (-np.sin(a) * x + np.cos(a) * y).astype(np.float32)
        h += np.float32(amp) * np.sin(np.float32(f) * u + np.float32(ph1)) * np.cos(np.float32(0.8 * f) * v + np.float32(ph2))
        norm += amp
        amp *= 0.55
    h = (h / np.float32(norm)) * np.float32(0.5) + np.float32(0.5)
    return h * np.float32(box[2] * 0.9) + np.float32(box[2] * 0.02)


def terrain_height(x, y, seed=7, box=(6000.0, 4000.0, 400.0)):
    """Height of terrain()/terrain_scan()'s surface (same seed) at one position: where a camera preset should look."""
    return float(_height(np.asarray([x], dtype=np.float32), np.asarray([y], dtype=np.float32), np.random.RandomState(seed + 1), np.asarray(box, dtype=np.float64))[0])


def terrain(n=36_000_000, seed=7, box=(6000.0, 4000.0, 400.0), tile=250.0, chunk=4_000_000):
    """Stand-in for Morro Bay (BASELINE config 2/3): a fractal height field over a 6 km x 4 km x 0.4 km box,
    emitted swath by swath (serpentine tiles of `tile` metres with uneven density) so that a 1 M-point batch is
    spatially compact, as consecutive records of an aerial LiDAR file are."""
    rng = np.random.RandomState(seed)
    box = np.asarray(box, dtype=np.float64)
    tx, ty = int(np.ceil(box[0] / tile)), int(np.ceil(box[1] / tile))
    dens = rng.gamma(4.0, 1.0, size=(ty, tx))
    counts = np.floor(dens / dens.sum() * n).astype(np.int64)
    counts.flat[: n - counts.sum()] += 1
    order = [(j, i if j % 2 == 0 else tx - 1 - i) for j in range(ty) for i in range(tx)]
    hrng = np.random.RandomState(seed + 1)
    hstate = hrng.get_state()
    out = np.empty(n, dtype=abi.point_dtype)
    pos = 0
    buf_j, buf_i, buf_c = [], [], []

    def flush():
        nonlocal pos
        if not buf_c:
            return
        cs = np.asarray(buf_c)
        m = int(cs.sum())
        ox = np.repeat(np.asarray(buf_i, dtype=np.float32) * np.float32(tile), cs)
        oy = np.repeat(np.asarray(buf_j, dtype=np.float32) * np.float32(tile), cs)
        x = np.minimum(ox + rng.random_sample(m).astype(np.float32) * np.float32(tile), np.float32(box[0] * 0.999999))
        y = np.minimum(oy + rng.random_sample(m).astype(np.float32) * np.float32(tile), np.float32(box[1] * 0.999999))
        hrng.set_state(hstate)
        z = _height(x, y, hrng, box) + rng.random_sample(m).astype(np.float32) * np.float32(0.15)
        t = np.clip(z / np.float32(box[2]), 0, 1)
        r = (60 + 180 * t).astype(np.uint32)
        g = (90 + 120 * (1 - np.abs(t - 0.5) * 2)).astype(np.uint32)
        b = (50 + 100 * (1 - t)).astype(np.uint32)
        out[pos:pos + m] = _pack(x, y, z, r, g, b)
        pos += m
        buf_j.clear(), buf_i.clear(), buf_c.clear()

    acc = 0
    for j, i in order:
        c = int(counts[j, i])
        if c == 0:
            continue
        buf_j.append(j), buf_i.append(i), buf_c.append(c)
        acc += c
        if acc >= chunk:
            flush()
            acc = 0
    flush()
    assert pos == n
    return out, box.astype(np.float32)


def terrain_scan(n=36_000_000, seed=7, box=(6000.0, 4000.0, 400.0), swath=250.0, chunk=4_000_000):
    """The same fractal terrain as terrain(), but emitted in ACQUISITION order, the order an airborne LiDAR file (and hence
    a .simlod converted from it, tools/las2simlod.mjs keeps the record order) stores its points: parallel flight swaths of
    `swath` metres, each covered by zig-zag scan lines across the swath, consecutive records one point spacing apart.
    Consecutive records are therefore spatial neighbours — unlike terrain(), whose records are shuffled inside 250 m tiles."""
    rng = np.random.RandomState(seed)
    box = np.asarray(box, dtype=np.float64)
    spacing = float(np.sqrt(box[0] * box[1] / n))                  # mean point spacing for n points over the footprint
    nsw = int(np.ceil(box[1] / swath))
    per_line = max(2, int(round(swath / spacing)))
    hrng = np.random.RandomState(seed + 1)
    hstate = hrng.get_state()
    out = np.empty(n, dtype=abi.point_dtype)
    counts = np.full(nsw, n // nsw, dtype=np.int64)
    counts[: n - counts.sum()] += 1
    pos = 0
    for s in range(nsw):
        m_total = int(counts[s])
        lines = int(np.ceil(m_total / per_line))
        dx = box[0] / lines
        for c0 in range(0, m_total, chunk):
            m = min(chunk, m_total - c0)
            k = np.arange(c0, c0 + m, dtype=np.int64)
            line, i = k // per_line, k % per_line
            i = np.where(line % 2 == 0, i, per_line - 1 - i)        # zig-zag mirror
            along = (line.astype(np.float64) + rng.random_sample(m) * 0.6) * dx
            if s % 2 == 1:
                along = box[0] - along                               # the aircraft turns around for the next swath
            across = s * swath + (i.astype(np.float64) + rng.random_sample(m) * 0.6) * (swath / per_line)
            x = np.clip(along, 0, box[0] * 0.999999).astype(np.float32)
            y = np.clip(across, 0, box[1] * 0.999999).astype(np.float32)
            hrng.set_state(hstate)
            z = _height(x, y, hrng, box) + rng.random_sample(m).astype(np.float32) * np.float32(0.15)
            t = np.clip(z / np.float32(box[2]), 0, 1)
            r = (60 + 180 * t).astype(np.uint32)
            g = (90 + 120 * (1 - np.abs(t - 0.5) * 2)).astype(np.uint32)
            b = (50 + 100 * (1 - t)).astype(np.uint32)
            out[pos:pos + m] = _pack(x, y, z, r, g, b)
            pos += m
    assert pos == n
    return out, box.astype(np.float32)


def hotspot(n=1_000_000, seed=11, level=6, cell=(21, 40, 13), box=(1.0, 1.0, 1.0)):
    """BASELINE config 5: every point inside ONE level-`level` octree cell of the unit cube (uniform inside it), so
    the first batch forces `level`+ split rounds and a camera aimed at the cell piles all samples on few pixels."""
    rs = np.random.RandomState(seed)
    s = np.float32(1.0 / (1 << level))
    v = rs.random_sample((n, 3)).astype(np.float32)
    v = np.minimum(v, np.float32(0.999999))
    base = np.asarray(cell, dtype=np.float32) * s
    p = base + v * s
    c = np.floor(v * np.float32(255.0)).astype(np.uint32)
    return _pack(p[:, 0], p[:, 1], p[:, 2], c[:, 0], c[:, 1], c[:, 2]), np.asarray(box, dtype=np.float32)


def write_simlod(path, points, box_size):
    """.simlod = 24-byte header (bbox min xyz, max xyz as float32... the converter writes 6 floats,
    tools/las2simlod.mjs:96-101) followed by the 16-byte records."""
    with open(path, "wb") as f:
        hdr = np.zeros(6, dtype=np.float32)
        hdr[3:] = box_size
        f.write(hdr.tobytes())
        f.write(points.tobytes())
