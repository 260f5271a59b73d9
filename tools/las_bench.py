"""k_decode_las alone: a rotating set of 42 raw 1 M-point batches (LAS format 2, 26-byte records) into 42 output slots — 1.76 GB per
round, beyond the 256 MiB Infinity Cache —, HIP events around 3 rounds; and the same bytes as one device-to-device copy for the denominator."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from simlod_amd import lasio
from simlod_amd.runtime import lib

L = lib()
batch, NSET = 1_000_000, 42
for fmt, bpp in ((2, 26), (3, 34), (7, 36)):
    rs = np.random.RandomState(5)
    rec = lasio.las_records(rs.randint(0, 6_000_000, size=(batch, 3)).astype(np.int32), rs.randint(0, 65536, size=(batch, 3)).astype(np.uint16), fmt)
    assert rec.shape[1] == bpp, rec.shape
    d_raw = torch.from_numpy(rec.reshape(-1)).to("cuda:0").repeat(NSET).reshape(NSET, -1)
    d_out = torch.empty((NSET, batch * 16), dtype=torch.uint8, device="cuda:0")
    scale3, off3 = (ctypes.c_double * 3)(1e-3, 1e-3, 1e-3), (ctypes.c_double * 3)(0.0, 0.0, 0.0)
    call = lambda k: L.simlod_decode_las(ctypes.c_void_p(d_raw[k].data_ptr()), ctypes.c_uint64(batch), ctypes.c_uint32(bpp), ctypes.c_uint32(fmt), scale3, off3,
                                         ctypes.c_void_p(d_out[k].data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    for k in range(NSET):
        call(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for rep in range(3):
        for k in range(NSET):
            call(k)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (3 * NSET)
    print(f"format {fmt} ({bpp} B records): {us:6.2f} us per 1 M-point launch (back to back) = {batch / us:7.0f} M points/s, {(bpp + 16) * batch / us / 1e6:6.2f} TB/s of {bpp} + 16 B per point", flush=True)
    # the same bytes in launches of 6 M points (six batches that lie back to back): what the kernel does when the ~3 us between two launches of a
    # stream are a twentieth of a launch instead of a quarter
    G = 6
    big = lambda k: L.simlod_decode_las(ctypes.c_void_p(d_raw[k].data_ptr()), ctypes.c_uint64(G * batch), ctypes.c_uint32(bpp), ctypes.c_uint32(fmt), scale3, off3,
                                        ctypes.c_void_p(d_out[k].data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    for k in range(0, NSET, G):
        big(k)
    torch.cuda.synchronize()
    e0.record()
    for rep in range(3):
        for k in range(0, NSET, G):
            big(k)
    e1.record(); torch.cuda.synchronize()
    usb = e0.elapsed_time(e1) * 1e3 / (3 * NSET // G)
    print(f"format {fmt} ({bpp} B records): {usb:6.2f} us per {G} M-point launch               = {G * batch / usb:7.0f} M points/s, {(bpp + 16) * G * batch / usb / 1e6:6.2f} TB/s", flush=True)
    del d_raw, d_out
src = torch.empty(1 << 30, dtype=torch.uint8, device="cuda:0"); dst = torch.empty_like(src)
for _ in range(2):
    dst.copy_(src)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    dst.copy_(src)
e1.record(); torch.cuda.synchronize()
print(f"1 GiB device-to-device copy: {2.0 * src.numel() / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e12:5.2f} TB/s (read + write)")
