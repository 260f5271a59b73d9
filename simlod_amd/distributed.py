"""Multi-GPU layer (new design — the reference is single-GPU): one process per GPU, every rank owns
the sub-octrees of a set of top-level cells of the SAME global cube, so node coordinates, names, voxel grids and the
LOD maths are those of the single-GPU octree.

  ingest   a point belongs to the rank that owns its level-`level` cell of the global cube.  Cells are dealt to ranks BY POINT COUNT
           (balanced_owners: every rank histograms its share of the input over the 8^level cells, one all-reduce(SUM) makes the
           global histogram, the same greedy assignment runs on every rank), and a mixed input is routed with ONE all-to-all
           (route_points: all_to_all_single of the 16-byte records, split sizes from an all-to-all of the counts) — every rank
           receives exactly what it owns, nothing else.  Pre-partitioned inputs (BASELINE config 4) send most records to
           themselves.
  render   every rank rasterises its own visible nodes; the frame is the element-wise MIN of the uint64 framebuffers
           (depth bits << 32 | colour, exactly what atomicMin builds on one GPU, render.cu:95-100) — one all-reduce —
           and the visible-node records are all-gathered so every rank holds the merged list (stats, LOD bookkeeping).
           HQS frames reduce the depth plane (MIN) and the colour sums (SUM) between the passes (render_frame).

All functions take torch tensors living wherever the process group's backend wants them (CUDA for nccl/RCCL, CPU for
gloo in the tests); nothing here launches kernels.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import abi


def cell_of(x, y, z, box_size, level=1):
    """Level-`level` octree cell of every point (the reference's quantisation, progressive_octree_voxels.cu:148-150:
    X = uint32(2^20 * (p - min) / size) with boxMin = 0) as a Morton-like code x<<2|y<<1|z per level."""
    size = np.float32(max(box_size))
    q = [np.minimum((np.float32(2 ** 20) * np.asarray(v, dtype=np.float32) / size).astype(np.uint32), np.uint32(2 ** 20 - 1)) for v in (x, y, z)]
    code = np.zeros(len(q[0]), dtype=np.uint32)
    for lv in range(level):
        s = np.uint32(19 - lv)
        code = (code << np.uint32(3)) | (((q[0] >> s) & 1) << np.uint32(2)) | (((q[1] >> s) & 1) << np.uint32(1)) | ((q[2] >> s) & 1)
    return code


def owner_of(points, box_size, world, level=1):
    """Rank that owns each point: cells are dealt round-robin (world 8, level 1: octant k -> rank k)."""
    return (cell_of(points["x"], points["y"], points["z"], box_size, level) % np.uint32(world)).astype(np.int64)


def exchange_points(points, owners, group=None):
    """Route one mixed batch: returns the records (from every rank, rank order, original order inside a rank) that THIS rank
    owns.  Implemented as an all-gather of size-padded buffers so that it runs on gloo as well as on RCCL."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = points.device if isinstance(points, torch.Tensor) else torch.device("cpu")
    rec = points if isinstance(points, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(points).view(np.uint8).reshape(-1, 16))
    own = owners if isinstance(owners, torch.Tensor) else torch.from_numpy(np.asarray(owners, dtype=np.int64))
    counts = torch.tensor([rec.shape[0]], dtype=torch.int64, device=dev)
    all_counts = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts, group=group)
    cap = int(max(int(c.item()) for c in all_counts))
    pad_r = torch.zeros((cap, 16), dtype=torch.uint8, device=dev); pad_r[: rec.shape[0]] = rec
    pad_o = torch.full((cap,), -1, dtype=torch.int64, device=dev); pad_o[: rec.shape[0]] = own
    gr = [torch.empty_like(pad_r) for _ in range(world)]
    go = [torch.empty_like(pad_o) for _ in range(world)]
    dist.all_gather(gr, pad_r, group=group)
    dist.all_gather(go, pad_o, group=group)
    mine = [gr[r][go[r] == rank] for r in range(world)]
    return torch.cat(mine, dim=0)


def cell_codes(points, box_size, level):
    """Level-`level` cell code (x<<2|y<<1|z per level, as cell_of) of every 16-byte record of a torch tensor, on the tensor's
    device (CUDA for RCCL jobs, CPU under gloo): the builder's quantisation X = uint32((2^20 * p) / size), boxMin = 0, bits 19..0 —
    so that every rank receives exactly the points the builder files under the cells it owns (also on cell boundaries and the max faces)."""
    xyz = points.reshape(-1, 16)[:, :12].contiguous().view(torch.float32).reshape(-1, 3)
    size = torch.tensor(float(np.float32(max(box_size))), dtype=torch.float32, device=points.device)
    # exactly the builder's arithmetic (simlod_device.hpp quantize / child_index): (2^20 * p) / size in fp32, left to right, truncated
    # (negative -> 0 as v_cvt_u32_f32 saturates), and only bits 19..0 are looked at — a coordinate on the max face wraps to cell 0
    q = ((xyz * torch.tensor(float(2 ** 20), dtype=torch.float32, device=points.device)) / size).clamp_(min=0.0).to(torch.int64) & (2 ** 20 - 1)
    code = torch.zeros(q.shape[0], dtype=torch.int64, device=points.device)
    for lv in range(level):
        s = 19 - lv
        code = (code << 3) | (((q[:, 0] >> s) & 1) << 2) | (((q[:, 1] >> s) & 1) << 1) | ((q[:, 2] >> s) & 1)
    return code


def balanced_owners(codes, world, level, group=None):
    """Owner rank of each of the 8^level cells, the same table on every rank: global point count per cell (all-reduce of the local
    histograms), cells taken heaviest first, each to the rank that is lightest so far (ties: lowest cell, lowest rank).
    Returns (owner table int64[8^level] on the codes' device, global counts int64[8^level] on the host)."""
    ncell = 8 ** level
    hist = torch.bincount(codes, minlength=ncell).to(torch.int64)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(hist, op=dist.ReduceOp.SUM, group=group)
    counts = hist.cpu().numpy()
    load = np.zeros(world, dtype=np.int64)
    owner = np.zeros(ncell, dtype=np.int64)
    for c in sorted(range(ncell), key=lambda c: (-int(counts[c]), c)):
        r = int(np.argmin(load))                   # first minimum = lowest rank
        owner[c] = r
        load[r] += counts[c]
    return torch.from_numpy(owner).to(codes.device), counts


def route_points(points, codes, owner_table, group=None):
    """Send every record to the rank that owns its cell: ONE all-to-all of the records (plus one of the counts).  Returns the records
    this rank owns — from rank 0 first, original order inside a source rank — and the number it received from each rank."""
    world = dist.get_world_size(group)
    rec = points.reshape(-1, 16)
    dest = owner_table[codes]
    order = torch.argsort(dest, stable=True)
    send = rec[order].contiguous()
    send_counts = torch.bincount(dest, minlength=world).to(torch.int64)
    recv_counts = torch.zeros_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    rs, ss = [int(v) for v in recv_counts.cpu()], [int(v) for v in send_counts.cpu()]
    out = torch.empty((sum(rs), 16), dtype=torch.uint8, device=points.device)
    dist.all_to_all_single(out, send, rs, ss, group=group)
    return out, rs


def compose_min(framebuffer_u64_as_i64, group=None):
    """In-place all-reduce(MIN) over the 64-bit depth|colour words.  The sign bit of a stored word is never set (a sample
    with negative depth bits never beats the +inf clear value, render.cu:95-100), so signed MIN == unsigned MIN."""
    dist.all_reduce(framebuffer_u64_as_i64, op=dist.ReduceOp.MIN, group=group)
    return framebuffer_u64_as_i64


def render_frame(renderer, uniforms, group=None, gather_capacity=4096):
    """One frame over all ranks, EXACT for plain and for HQS shading: every rank rasterises its own sub-octrees
    and the ranks reduce the planes between the parts of kernel_render (include/simlod_hip.h, simlod_launch_render_part):
        part 0 | HQS: all-reduce(MIN) depth | part 1 | HQS: all-reduce(SUM) {R,G,B,count} | part 2 | all-reduce(MIN) framebuffer | part 3
    so a pixel's average is taken over the samples of ALL ranks within 1 % of the GLOBAL nearest depth, exactly as one GPU holding
    every sample would.  `renderer` is a runtime.DeviceOctree (RCCL) or, in the CPU tests, an oracle.HostOctree (gloo): anything
    with render_part / depth_plane / sum_planes / framebuffer_words / visible_records.  Returns (visible records, counts)."""
    u = np.ascontiguousarray(uniforms).reshape(1)
    hqs = bool(u["useHighQualityShading"][0])
    boxes = bool(u["showBoundingBox"][0])
    renderer.render_part(uniforms, 0)
    # the visible-node lists are final after part 0: their all-gather is issued now, asynchronously, and runs beside the passes and
    # reductions that follow (RCCL puts it on its own stream behind part 0; the frame waits for it at the very end)
    pending = None
    if hasattr(renderer, "visible_records_early"):
        vis, n = renderer.visible_records_early()
        pending = gather_visible(vis, n, group=group, capacity=gather_capacity, async_op=True)
    if hqs:
        dist.all_reduce(renderer.depth_plane(), op=dist.ReduceOp.MIN, group=group)
        renderer.render_part(uniforms, 1)
        dist.all_reduce(renderer.sum_planes(), op=dist.ReduceOp.SUM, group=group)
        renderer.render_part(uniforms, 2)
    if not hqs or boxes:                       # resolved HQS frames are identical on every rank; only rank-local debug lines differ
        compose_min(renderer.framebuffer_words(), group=group)
    renderer.render_part(uniforms, 3)
    if pending is not None:
        return pending()
    vis, n = renderer.visible_records()
    return gather_visible(vis, n, group=group, capacity=gather_capacity)


def gather_visible(visible_bytes, count, group=None, capacity=4096, async_op=False):
    """All-gather the first `count` visible-node records (152 B each) of every rank; returns (records[world, capacity, 152], counts).
    `count` may be a host int or a one-element tensor on the records' device (then nothing synchronises with the host: the
    first `capacity` records travel as they are and the gathered counts say how many of them are valid).  async_op: returns a
    function that waits for the two collectives and hands out the result."""
    world = dist.get_world_size(group)
    dev = visible_bytes.device
    if isinstance(count, torch.Tensor):
        buf = visible_bytes[: capacity * 152].view(capacity, 152).contiguous()
        cnt = torch.clamp(count.to(torch.int64).reshape(1), max=capacity)
    else:
        n = min(int(count), capacity)
        buf = torch.zeros((capacity, 152), dtype=torch.uint8, device=dev)
        buf[:n] = visible_bytes[: n * 152].view(n, 152)
        cnt = torch.tensor([n], dtype=torch.int64, device=dev)
    bufs = [torch.empty_like(buf) for _ in range(world)]
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    if async_op:
        works = [dist.all_gather(bufs, buf, group=group, async_op=True), dist.all_gather(cnts, cnt, group=group, async_op=True)]

        def finish():
            for w in works:
                w.wait()
            return torch.stack(bufs), torch.cat(cnts)
        return finish
    dist.all_gather(bufs, buf, group=group)
    dist.all_gather(cnts, cnt, group=group)
    return torch.stack(bufs), torch.cat(cnts)
