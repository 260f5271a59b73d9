"""Loader row, CPU side: the numpy restatement of the reference's LAS parse loop and the host-side header /
.simlod readers, pinned to fixtures minted from the reference itself (tests/golden/make_golden_las.py: loadLasNative built in
place, tools/las2simlod.mjs run with node), and — where oracle/_ref/libref_las.so is present — to the reference live."""
import os
import tempfile

import numpy as np
import pytest

import oracle
from simlod_amd import abi, lasio

GOLD = os.path.join(os.path.dirname(__file__), "golden")
G = np.load(os.path.join(GOLD, "las_decode.npz"))
NUM_CASES = len([k for k in G.files if k.endswith("_fmt")])


def case(ci, tmpdir):
    fmt, vmaj, vmin, bpp, first, count = (int(v) for v in G[f"c{ci}_fmt"])
    path = os.path.join(tmpdir, f"c{ci}.las")
    G[f"c{ci}_file"].tofile(path)
    return fmt, bpp, first, count, path


@pytest.mark.parametrize("ci", range(NUM_CASES))
def test_header_and_decode_match_the_reference_fixture(ci, tmp_path):
    fmt, bpp, first, count, path = case(ci, str(tmp_path))
    h = lasio.load_header(path)
    hi, hf = G[f"c{ci}_header"], G[f"c{ci}_header_f"]
    assert [h.versionMajor, h.versionMinor, h.headerSize, h.offsetToPointData, h.format, h.bytesPerPoint, h.numPoints] == [int(v) for v in hi]
    assert np.array_equal(np.array(h.scale + h.offset + h.min + h.max, dtype=np.float64).view(np.uint64), hf.view(np.uint64))
    assert h.bytesPerPoint == bpp and h.format == fmt
    tr = tuple(-m for m in h.min)
    raw = lasio.read_records(path, h, first, count)
    assert raw.size == count * bpp
    got = oracle.decode_las_port(raw, bpp, fmt, h.scale, lasio.decode_offset(h, tr))
    want = G[f"c{ci}_points"]
    for k in "xyz":                                                  # bit-exact fp32 positions
        assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), k
    if fmt in lasio.RGB_OFFSET:                                      # r, g, b where the reference defines them
        assert np.array_equal(got["color"] & 0xffffff, want["color"] & 0xffffff)
    else:
        assert np.all(got["color"] == 0xff000000)
    assert np.all(got["color"] >> 24 == 255)


def test_simlod_reader_reads_what_the_reference_converter_wrote():
    pts, box = lasio.read_simlod(os.path.join(GOLD, "tiny_f2.simlod"))
    h = lasio.load_header(os.path.join(GOLD, "tiny_f2.las"))
    assert len(pts) == h.numPoints == 3000
    assert np.array_equal(box, (np.array(h.max) - np.array(h.min)).astype(np.float32))
    raw = lasio.read_records(os.path.join(GOLD, "tiny_f2.las"), h, 0, h.numPoints)
    dec = oracle.decode_las_port(raw, h.bytesPerPoint, h.format, h.scale, lasio.decode_offset(h, tuple(-m for m in h.min)))
    # tools/las2simlod.mjs:131-147 forms X*scale + offset - min (two roundings in fp64) where the loader forms
    # X*scale + (offset - min): identical after the fp32 conversion on this file, alpha 255 in both
    assert np.array_equal(dec.view(np.uint8), pts.view(np.uint8))


def test_batches_and_ragged_reads(tmp_path):
    h = lasio.load_header(os.path.join(GOLD, "tiny_f2.las"))
    b = lasio.batches(h, batch=1000)
    assert b == [(0, 1000), (1000, 1000), (2000, 1000)]
    assert lasio.batches(h, batch=1024)[-1] == (2048, 952)
    assert lasio.read_records(os.path.join(GOLD, "tiny_f2.las"), h, 2990, 100).size == 10 * h.bytesPerPoint   # clamped like unsuck.hpp:478-498
    assert lasio.read_records(os.path.join(GOLD, "tiny_f2.las"), h, 3000, 5).size == 0


def test_writer_roundtrip_las14_point_count(tmp_path):
    rec = lasio.las_records(np.zeros((5, 3), dtype=np.int32), np.zeros((5, 3), dtype=np.uint16), 7)
    p = str(tmp_path / "v14.las")
    lasio.write_las(p, rec, 7, (1, 1, 1), (0, 0, 0), (0, 0, 0), (1, 1, 1), version=(1, 4))
    h = lasio.load_header(p)
    assert (h.versionMinor, h.numPoints, h.bytesPerPoint, h.headerSize, h.offsetToPointData) == (4, 5, 36, 375, 375)


@pytest.mark.skipif(not oracle.have_ref_las(), reason="oracle/_ref/libref_las.so not built (needs /root/reference at build time)")
def test_port_equals_reference_live(tmp_path):
    rs = np.random.RandomState(77)
    for fmt, ver in [(2, (1, 2)), (3, (1, 2)), (7, (1, 4)), (1, (1, 1))]:
        n = 20_000
        xyz = rs.randint(-2 ** 31, 2 ** 31 - 1, size=(n, 3), dtype=np.int64).astype(np.int32)
        rgb = rs.randint(0, 65536, size=(n, 3)).astype(np.uint16)
        p = str(tmp_path / f"l{fmt}.las")
        lasio.write_las(p, lasio.las_records(xyz, rgb, fmt, seed=fmt), fmt, (1e-3, 1e-2, 1e-7), (5e5, -4e6, 1e3), (1e5, -5e6, 0), (9e5, 0, 5e3), version=ver)
        h = lasio.load_header(p)
        rh = oracle.ref_las_header(p)
        assert rh["numPoints"] == h.numPoints and rh["offsetToPointData"] == h.offsetToPointData
        tr = tuple(-m for m in h.min)
        ref = oracle.ref_las_load(p, 123, 15_000, tr)
        port = oracle.decode_las_port(lasio.read_records(p, h, 123, 15_000), h.bytesPerPoint, h.format, h.scale, lasio.decode_offset(h, tr))
        for k in "xyz":
            assert np.array_equal(ref[k].view(np.uint32), port[k].view(np.uint32))
        if fmt in lasio.RGB_OFFSET:
            assert np.array_equal(ref["color"] & 0xffffff, port["color"] & 0xffffff)
