"""Host side of the loader row: LAS / .simlod files -> batches for the ring.

Mirrors the reference's loader interface — `LasHeader` + `load_header` (modules/progressive_octree/LasLoader.h:8-55),
the byte range a loader thread reads for a batch (LasLoader.cpp:147-166: offsetToPointData + bytesPerPoint * firstPoint,
bytesPerPoint * numPoints), the .simlod layout (24-byte header + 16-byte records, main_progressive_octree.cpp:925-929,
tools/las2simlod.mjs:96-147) — but does not parse points on the CPU: the raw records go to the device and
`simlod_decode_las` (simlod_amd/csrc/loader.hip) writes the XYZRGBA points into the ring.  LAZ (laszip) is out of scope.

`write_las` / `las_records` build synthetic LAS files for tests and benchmarks (there is no network for real ones).
"""
import dataclasses
import struct

import numpy as np

from . import abi

# RGB byte offset inside a point record, for the formats the reference takes colour from (LasLoader.cpp:177-185)
RGB_OFFSET = {2: 20, 3: 28, 5: 28, 7: 30}
# minimum record length of the LAS point data record formats 0-10 (ASPRS LAS 1.4 R15, tables 7-17)
FORMAT_BYTES = {0: 20, 1: 28, 2: 26, 3: 34, 4: 57, 5: 63, 6: 30, 7: 36, 8: 38, 9: 59, 10: 67}


@dataclasses.dataclass
class LasHeader:
    """Same fields as the reference's LasHeader (LasLoader.h:8-19)."""
    versionMajor: int = 0
    versionMinor: int = 0
    headerSize: int = 0
    offsetToPointData: int = 0
    format: int = 0
    bytesPerPoint: int = 0
    numPoints: int = 0
    scale: tuple = (0.0, 0.0, 0.0)
    offset: tuple = (0.0, 0.0, 0.0)
    min: tuple = (0.0, 0.0, 0.0)
    max: tuple = (0.0, 0.0, 0.0)


def load_header(path):
    """loadHeader, LasLoader.h:21-55: fixed byte offsets into the first 375 bytes; the legacy 32-bit point count for
    LAS <= 1.3, the 64-bit one at byte 247 otherwise."""
    with open(path, "rb") as f:
        b = f.read(375).ljust(375, b"\0")
    h = LasHeader()
    h.versionMajor, h.versionMinor = b[24], b[25]
    h.headerSize = struct.unpack_from("<H", b, 94)[0]
    h.offsetToPointData = struct.unpack_from("<I", b, 96)[0]
    h.format = b[104]
    h.bytesPerPoint = struct.unpack_from("<H", b, 105)[0]
    if h.versionMajor == 1 and h.versionMinor <= 3:
        h.numPoints = struct.unpack_from("<I", b, 107)[0]
    else:
        h.numPoints = struct.unpack_from("<Q", b, 247)[0]
    d = lambda o: struct.unpack_from("<d", b, o)[0]
    h.scale = (d(131), d(139), d(147))
    h.offset = (d(155), d(163), d(171))
    h.max = (d(179), d(195), d(211))
    h.min = (d(187), d(203), d(219))
    return h


def read_records(path, header, first, count):
    """The bytes a loader thread reads for one batch (LasLoader.cpp:147-166), as a uint8 array."""
    count = max(0, min(count, header.numPoints - first))
    with open(path, "rb") as f:
        f.seek(header.offsetToPointData + header.bytesPerPoint * first)
        return np.fromfile(f, dtype=np.uint8, count=header.bytesPerPoint * count)


def decode_offset(header, translation):
    """offset_x = header.offset[0] + translation[0] ... (LasLoader.cpp:197-199), in fp64."""
    return tuple(float(np.float64(o) + np.float64(t)) for o, t in zip(header.offset, translation))


def batches(header, batch=abi.MAX_BATCH_SIZE):
    """(first, count) of every batch of a file, as main_progressive_octree.cpp:893-910 queues them."""
    return [(f, min(batch, header.numPoints - f)) for f in range(0, header.numPoints, batch)]


def read_simlod(path):
    """(points, box_size): the 24-byte header holds min (zeros) and max = box size as 6 float32."""
    with open(path, "rb") as f:
        hdr = np.frombuffer(f.read(24), dtype=np.float32)
        pts = np.frombuffer(f.read(), dtype=abi.point_dtype)
    return pts, (hdr[3:6] - hdr[0:3]).astype(np.float32)


def las_records(xyz_int, rgb16, fmt, bytes_per_point=None, seed=0):
    """Raw point records (uint8 [n, bytesPerPoint]) of LAS point format `fmt`: int32 X,Y,Z at bytes 0-11, RGB16 where the
    format has it, every other byte pseudo-random (intensity, flags, gps time, extra bytes ... — a decoder must ignore them)."""
    n = len(xyz_int)
    bpp = FORMAT_BYTES[fmt] if bytes_per_point is None else bytes_per_point
    rec = np.random.RandomState(seed).randint(0, 256, size=(n, bpp), dtype=np.uint8)
    rec[:, 0:12] = np.ascontiguousarray(xyz_int.astype("<i4")).view(np.uint8).reshape(n, 12)
    off = {2: 20, 3: 28, 5: 28, 7: 30, 8: 30, 10: 30}.get(fmt)
    if off is not None and rgb16 is not None:
        rec[:, off:off + 6] = np.ascontiguousarray(rgb16.astype("<u2")).view(np.uint8).reshape(n, 6)
    return rec


def write_las(path, records, fmt, scale, offset, mins, maxs, version=(1, 2), header_size=None, vlr_bytes=0, num_points=None):
    """Minimal LAS file: public header block (227 B for 1.2, 375 B for 1.4), `vlr_bytes` of filler, the point records
    (`num_points`: the count the header announces when the records are appended later)."""
    n, bpp = records.shape
    if num_points is not None:
        n = num_points
    major, minor = version
    hs = header_size if header_size is not None else (375 if minor >= 4 else 227)
    b = bytearray(hs)
    b[0:4] = b"LASF"
    b[24], b[25] = major, minor
    struct.pack_into("<H", b, 94, hs)
    struct.pack_into("<I", b, 96, hs + vlr_bytes)
    b[104] = fmt
    struct.pack_into("<H", b, 105, bpp)
    struct.pack_into("<I", b, 107, n if minor <= 3 else min(n, 0xffffffff))
    struct.pack_into("<3d", b, 131, *scale)
    struct.pack_into("<3d", b, 155, *offset)
    for k in range(3):
        struct.pack_into("<d", b, 179 + 16 * k, maxs[k])
        struct.pack_into("<d", b, 187 + 16 * k, mins[k])
    if minor >= 4:
        struct.pack_into("<Q", b, 247, n)
    with open(path, "wb") as f:
        f.write(bytes(b))
        f.write(bytes(vlr_bytes))
        f.write(records.tobytes())


def points_to_las(path, points, box_size, fmt=2, scale=0.001, world_min=(0.0, 0.0, 0.0), version=(1, 2), seed=0, chunk=4_000_000):
    """Synthetic LAS file whose decoded (translated) positions are close to `points`: X = round((x + world_min) / scale).
    Written `chunk` records at a time (a 350 M-point file is 9 GB; its records never exist in memory all at once)."""
    wm = np.asarray(world_min, dtype=np.float64)
    n = len(points)
    write_las(path, np.zeros((0, FORMAT_BYTES[fmt]), dtype=np.uint8), fmt, (scale,) * 3, (0.0, 0.0, 0.0), wm, wm + np.asarray(box_size, dtype=np.float64), version=version, num_points=n)
    with open(path, "ab") as f:
        for i in range(0, n, chunk):
            p = points[i:i + chunk]
            xyz = np.stack([p["x"], p["y"], p["z"]], axis=1).astype(np.float64) + wm
            xyz_int = np.rint(xyz / scale).astype(np.int64).astype(np.int32)
            c = p["color"]
            rgb8 = np.stack([c & 255, (c >> 8) & 255, (c >> 16) & 255], axis=1).astype(np.uint16)
            rgb16 = rgb8 * np.uint16(256) + rgb8                   # 16-bit colour as scanners write it; decodes back to rgb8
            f.write(las_records(xyz_int, rgb16, fmt, seed=seed + i // chunk).tobytes())
    return load_header(path)
