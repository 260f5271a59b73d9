import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from simlod_amd import abi, camera, synthetic
from simlod_amd.runtime import DeviceOctree
from util import host_image_of
from test_gpu_parity import _oracle_render, _ingest
pts, box = synthetic.uniform_cube(1_000_000, seed=1234)
Wd = Hd = 512
T = camera.lookat_transform((1.8, -1.2, 1.4), (0.5, 0.5, 0.3), Wd, Hd)
dev = DeviceOctree("cuda:0", persistent_bytes=1 << 30, ring_slots=2, max_pixels=Wd * Hd)
u = dev.uniforms(Wd, Hd, T, box)
_ingest(dev, u, [pts])
u["showBoundingBox"] = 1
dev.render(u)
fd = dev.framebuffer(Wd, Hd)
nodes, pers, nn = host_image_of(dev)
fo, _, _ = _oracle_render(nodes, nn, u)
d = np.nonzero(fd != fo)[0]
print(len(d), "diff")
for i in d[:25]:
    print(i % Wd, i // Wd, "dev %016x" % fd[i], "ora %016x" % fo[i])
u["showBoundingBox"] = 0
dev.render(u); f0 = dev.framebuffer(Wd, Hd)
ld = np.nonzero(fd != f0)[0]; lo = np.nonzero(fo != f0)[0]
print("line pixels dev", len(ld), "oracle", len(lo), "only dev", len(np.setdiff1d(ld, lo)), "only oracle", len(np.setdiff1d(lo, ld)))

import torch, oracle
off_lines = 100000*152 + 7*16
cnt = int(dev.render_buffer[off_lines:off_lines+4].cpu().numpy().view(np.uint32)[0])
# re-render with lines to refill vertices (last render was without)
u["showBoundingBox"] = 1; dev.render(u); torch.cuda.synchronize()
cnt = int(dev.render_buffer[off_lines:off_lines+4].cpu().numpy().view(np.uint32)[0])
verts = dev.render_buffer[off_lines+32: off_lines+32+cnt*16].cpu().numpy().copy()
fb2 = f0.copy()
uu = np.ascontiguousarray(u).reshape(1)
oracle.port_lib().oracle_rasterize_lines(ctypes.c_void_p(uu.ctypes.data), ctypes.c_void_p(verts.ctypes.data), cnt, ctypes.c_void_p(fb2.ctypes.data))
print("vertices", cnt, "host raster of DEVICE vertices vs device fb: diff", int((fb2 != fd).sum()), " vs oracle fb: diff", int((fb2 != fo).sum()))
