import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from simlod_amd import abi, camera, synthetic
from simlod_amd.runtime import DeviceOctree
import oracle
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import *

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "terrain"
t = time.time()
pts, box = (synthetic.terrain(n, seed=7) if kind == "terrain" else synthetic.hotspot(n) if kind == "hotspot" else synthetic.uniform_cube(n))
print("gen %.1fs" % (time.time() - t), flush=True)
W, H = 1920, 1080
T = camera.world_view_proj(camera.orbit_view(-0.207, -0.797, 3866.886 * box[0] / 6000, (box[0] / 2, box[1] / 2, 0.35 * box[2])), camera.perspective(aspect=W / H))
dev = DeviceOctree("cuda:0", persistent_bytes=8 << 30, max_pixels=W * H)
u = dev.uniforms(W, H, T, box)
dev.reset(u)
ref = oracle.HostOctree("port", persistent_bytes=8 << 30, ring_slots=50)
ref.reset(u)
B = abi.MAX_BATCH_SIZE
for i in range(0, n, B):
    dev.upload(pts[i:i + B]); ref.upload(pts[i:i + B])
    if ((i // B) + 1) % 20 == 0 or i + B >= n:
        torch.cuda.synchronize(); t = time.time()
        nl = dev.drain(u)
        torch.cuda.synchronize(); dt = time.time() - t
        ds = dev.read_stats()
        t = time.time()
        while int(ref.stats['batchletIndex'][0]) < int(ref.num_uploaded[0]): ref.construct(u)
        dtr = time.time() - t
        rs = ref.stats[0]
        print("launches %d: dev %.1f ms  oracle %.1f s" % (nl, dt * 1e3, dtr), "dbg=%#x" % int(ds["dbg"]), "err", ref.last_error())
        for f in STATS_BUILD_FIELDS:
            flag = "" if int(ds[f]) == int(rs[f]) else "   <<<<<< MISMATCH"
            print("   %-28s %14d %14d%s" % (f, int(ds[f]), int(rs[f]), flag))
nodes, pers, nn = host_image_of(dev)
try:
    assert_dumps_equal(oracle.dump_image(nodes, nn), ref.dump(), "construct")
    print("DUMPS EQUAL")
except AssertionError as e:
    print("DUMP MISMATCH:", e)
