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


def partition_and_route(points, box_size, world, level=3, slice_points=32_000_000, group=None):
    """cell_codes -> balanced_owners -> route_points for inputs of any size with BOUNDED transient memory: the records are looked at in
    slices of `slice_points` (codes, destinations and the sort permutation of one slice at a time: 56 B per slice point = 1.8 GB for the
    default slice, whatever the input's size), the result is allocated ONCE from the exchanged totals, and every slice is one
    all_to_all_single into a slice-sized staging buffer whose per-source parts are appended to the result (records from rank 0 first,
    original order inside a source rank — the order route_points leaves).  Peak per rank on top of the input: the result (what the rank
    owns) + 2 slices of records + the slice's index arrays; a rank may free its input afterwards.
    Returns (records this rank owns [n, 16] uint8, owner table, global counts per cell, records received from each rank)."""
    assert dist.is_initialized() or world == 1, "partition_and_route exchanges records over torch.distributed: initialise a process group first (one rank needs none)"
    rec = points.reshape(-1, 16)
    n = rec.shape[0]
    ncell = 8 ** level
    hist = torch.zeros(ncell, dtype=torch.int64, device=points.device)
    for first in range(0, n, slice_points):
        hist += torch.bincount(cell_codes(rec[first: first + slice_points], box_size, level), minlength=ncell)
    local = hist.clone()
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(hist, op=dist.ReduceOp.SUM, group=group)
    counts = hist.cpu().numpy()
    load = np.zeros(world, dtype=np.int64)
    owner = np.zeros(ncell, dtype=np.int64)
    for c in sorted(range(ncell), key=lambda c: (-int(counts[c]), c)):             # (balanced_owners' rule)
        r = int(np.argmin(load))
        owner[c] = r
        load[r] += counts[c]
    owner_table = torch.from_numpy(owner).to(points.device)
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):      # one rank owns every cell: nothing to exchange
        return rec, owner_table, counts, [n]
    # what this rank sends to / receives from everybody, in all: from its own histogram
    send_total = torch.zeros(world, dtype=torch.int64, device=points.device).index_add_(0, owner_table, local)
    recv_total = torch.zeros_like(send_total)
    dist.all_to_all_single(recv_total, send_total, group=group)
    rt = [int(v) for v in recv_total.cpu()]
    out = torch.empty((sum(rt), 16), dtype=torch.uint8, device=points.device)
    offset = np.concatenate([[0], np.cumsum(rt)])[:-1].astype(np.int64)             # where source rank r's records start in `out`
    filled = np.zeros(world, dtype=np.int64)
    nslices = torch.tensor([(n + slice_points - 1) // slice_points], dtype=torch.int64, device=points.device)
    dist.all_reduce(nslices, op=dist.ReduceOp.MAX, group=group)                     # every rank takes part in every exchange, with an empty slice if it has run out
    for k in range(int(nslices.item())):
        sl = rec[k * slice_points: (k + 1) * slice_points]
        dest = owner_table[cell_codes(sl, box_size, level)] if sl.shape[0] else torch.zeros(0, dtype=torch.int64, device=points.device)
        send = sl[torch.argsort(dest, stable=True)].contiguous()
        sc = torch.bincount(dest, minlength=world).to(torch.int64)
        rc = torch.zeros_like(sc)
        dist.all_to_all_single(rc, sc, group=group)
        rs, ss = [int(v) for v in rc.cpu()], [int(v) for v in sc.cpu()]
        stage = torch.empty((sum(rs), 16), dtype=torch.uint8, device=points.device)
        dist.all_to_all_single(stage, send, rs, ss, group=group)
        at = 0
        for r in range(world):
            out[offset[r] + filled[r]: offset[r] + filled[r] + rs[r]] = stage[at: at + rs[r]]
            filled[r] += rs[r]; at += rs[r]
        del dest, send, stage
    assert [int(v) for v in filled] == rt
    return out, owner_table, counts, rt


MAX_POINTS_PER_NODE = 50_000          # structures.cuh:21: a leaf that holds more splits (progressive_octree_voxels.cu:209-217)
TRUNK_LEVELS = 3                      # ranks own level-3 cells: the nodes of levels 0..2 are shared by all of them


def trunk_mask(counts):
    """Which nodes of the shared upper levels (0, 1, 2) are INNER nodes of the single-GPU octree of the whole data set: those whose GLOBAL
    point count exceeds 50 000 (a leaf splits iff its count crosses the limit, progressive_octree_voxels.cu:209-217; the final topology does
    not depend on the order of arrival).  `counts`: the all-reduced histogram over the 512 level-3 cells, indexed by cell code (cell_codes;
    partition_and_route and balanced_owners return it).  Returns (lo, hi), the 73-bit mask of simlod_context_set_trunk_mask: bit 0 the
    root, bit 1 + c the level-1 node with cell code c, bit 9 + c the level-2 node with code c.

    A rank that refined these nodes from ITS points alone would keep one as a leaf (and draw its points) where the single GPU has an inner
    node (and draws its voxels) — the composed frame would show a different LOD cut there.  With the mask every rank's upper levels have
    the single-GPU octree's topology; a voxel cell of an upper node lies inside ONE level-3 cell (a level-l node's grid has 128 cells per
    axis, a level-3 cell spans 128 / 2^(3 - l) >= 16 of them), so the ranks' voxel sets are disjoint, their union is the single GPU's, and
    the MIN / SUM composition of render_frame yields the single-GPU frame."""
    c = np.asarray(counts, dtype=np.int64).reshape(-1)
    assert c.size == 8 ** TRUNK_LEVELS, "trunk_mask wants the histogram over the 512 level-3 cells"
    mask = 0
    if int(c.sum()) > MAX_POINTS_PER_NODE:
        mask |= 1
    for code, n in enumerate(c.reshape(8, 64).sum(axis=1)):
        if int(n) > MAX_POINTS_PER_NODE:
            mask |= 1 << (1 + code)
    for code, n in enumerate(c.reshape(64, 8).sum(axis=1)):
        if int(n) > MAX_POINTS_PER_NODE:
            mask |= 1 << (9 + code)
    return mask & ((1 << 64) - 1), mask >> 64


def global_trunk_mask(points, box_size, group=None, slice_points=32_000_000):
    """trunk_mask of the records the ranks hold between them (any partition, any order): level-3 histogram of this rank's records, one
    all-reduce(SUM) of 512 counters.  For jobs that did not come through partition_and_route (pre-partitioned inputs, BASELINE config 4)."""
    rec = points.reshape(-1, 16)
    hist = torch.zeros(8 ** TRUNK_LEVELS, dtype=torch.int64, device=points.device)
    for first in range(0, rec.shape[0], slice_points):
        hist += torch.bincount(cell_codes(rec[first: first + slice_points], box_size, TRUNK_LEVELS), minlength=8 ** TRUNK_LEVELS)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(hist, op=dist.ReduceOp.SUM, group=group)
    return trunk_mask(hist.cpu().numpy())


def compose_min(framebuffer_u64_as_i64, group=None):
    """In-place all-reduce(MIN) over the 64-bit depth|colour words.  The sign bit of a stored word is never set (a sample
    with negative depth bits never beats the +inf clear value, render.cu:95-100), so signed MIN == unsigned MIN."""
    dist.all_reduce(framebuffer_u64_as_i64, op=dist.ReduceOp.MIN, group=group)
    return framebuffer_u64_as_i64


def render_frame(renderer, uniforms, group=None, gather_capacity=4096, check_overflow=True):
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
        pending = gather_visible(vis, n, group=group, capacity=gather_capacity, async_op=True, check_overflow=check_overflow)
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
    return gather_visible(vis, n, group=group, capacity=gather_capacity, check_overflow=check_overflow)


def render_frames_pipelined(renderer, frames, group=None, gather_capacity=4096, on_frame=None, check_overflow=True):
    """A sequence of frames (`frames`: one Uniforms record each) composed across ranks exactly as render_frame composes one, with TWO frames
    in flight: while the planes of frame f are on the wire, the ranks rasterise the next part of frame f + 1 — on an 8-GPU ring over xGMI the
    reductions of an HQS frame (8.3 + 33.2 MB at 1080p) take about as long as its passes, and inside ONE frame each reduction feeds the very
    next pass, so only a second frame gives the links something to overlap with.  Needs a second set of planes: `renderer.select_frame(k)`
    (runtime.DeviceOctree, oracle.HostOctree) switches between two render buffers; the octree, and Stats, are shared — Stats holds the
    counters of the frame whose part 3 ran last.  `on_frame(index, renderer, records, counts)` is called as each frame completes (frames of
    one kind complete in order; a plain frame has fewer stages than an HQS one and can overtake it), with that frame's buffers selected."""
    todo = list(enumerate(frames))
    active = []                                              # frames in flight, oldest first: dicts with slot, uniforms, next stage, pending collectives

    def issue(f, stage):
        """run stage `stage` of frame f (its inputs are complete) and put what it produced on the wire; returns False when the frame is done"""
        renderer.select_frame(f["slot"])
        u = f["u"]
        hqs, boxes = f["hqs"], f["boxes"]
        f["works"] = []
        if stage == 0:
            renderer.render_part(u, 0)
            if hasattr(renderer, "visible_records_early"):
                vis, n = renderer.visible_records_early()
                f["vis"] = gather_visible(vis, n, group=group, capacity=gather_capacity, async_op=True, check_overflow=check_overflow)
            if hqs:
                f["works"].append(dist.all_reduce(renderer.depth_plane(), op=dist.ReduceOp.MIN, group=group, async_op=True)); f["next"] = 1
            else:
                f["works"].append(dist.all_reduce(renderer.framebuffer_words(), op=dist.ReduceOp.MIN, group=group, async_op=True)); f["next"] = 3
        elif stage == 1:
            renderer.render_part(u, 1)
            f["works"].append(dist.all_reduce(renderer.sum_planes(), op=dist.ReduceOp.SUM, group=group, async_op=True)); f["next"] = 2
        elif stage == 2:
            renderer.render_part(u, 2)
            if boxes:
                f["works"].append(dist.all_reduce(renderer.framebuffer_words(), op=dist.ReduceOp.MIN, group=group, async_op=True))
            f["next"] = 3
        else:
            renderer.render_part(u, 3)
            if f.get("vis") is not None:
                recs, cnts = f["vis"]()
            else:
                vis, n = renderer.visible_records()
                recs, cnts = gather_visible(vis, n, group=group, capacity=gather_capacity, check_overflow=check_overflow)
            if on_frame is not None:
                on_frame(f["index"], renderer, recs, cnts)
            return False
        return True

    while todo or active:
        if todo and len(active) < 2:                         # admit a frame: its part 0 runs beside the older frame's reduction
            index, u = todo.pop(0)
            uu = np.ascontiguousarray(u).reshape(1)
            slot = 1 if any(g["slot"] == 0 for g in active) else 0
            f = {"index": index, "u": u, "slot": slot, "hqs": bool(uu["useHighQualityShading"][0]), "boxes": bool(uu["showBoundingBox"][0]), "vis": None}
            issue(f, 0)
            active.append(f)
            if todo and len(active) < 2:
                continue
        f = active[0]                                        # the older frame: its reduction has had a whole part of the other frame to complete
        for w in f["works"]:
            w.wait()
        if issue(f, f["next"]):
            active.append(active.pop(0))                     # it is on the wire again: the other frame's turn
        else:
            active.pop(0)


def visible_overflowed(counts, capacity):
    """The one look from the host at the counts a gather_visible(..., check_overflow=False) returned: did a rank have more visible nodes than travelled?"""
    most = int(counts.max().item())
    if most > abi.MAX_VISIBLE_NODES:
        raise VisibleOverflow(f"a rank reports {most} visible nodes; the visible-node array holds {abi.MAX_VISIBLE_NODES}")
    return most > capacity


class VisibleOverflow(RuntimeError):
    """A rank has more visible nodes than the visible-node array holds (render.cu:1108: 100 000) — the multi-rank counterpart of
    SIMLOD_ERR_VISIBLE_OVERFLOW in Stats.dbg."""


def gather_visible(visible_bytes, count, group=None, capacity=4096, async_op=False, check_overflow=True):
    """All-gather the first `count` visible-node records (152 B each) of every rank; returns (records[world, cap, 152], counts) with
    cap >= every rank's count: NOTHING is dropped.  `capacity` records per rank travel at once (no host synchronisation: `count` may be a
    one-element tensor on the records' device); the gathered counts are the TRUE counts, and should one of them exceed `capacity` — BASELINE
    config 5 has 4 097 visible nodes on one GPU — the records are gathered once more with room for the largest (the power of two above it).
    More than the visible-node array can hold (abi.MAX_VISIBLE_NODES) raises VisibleOverflow.  async_op: returns a function that waits for
    the collectives and hands out the result.  check_overflow=False: no look at the counts from the host — the frame loop stays free of host
    synchronisation; the caller checks when it suits it (visible_overflowed(counts, capacity): True = the records of that frame were cut at
    `capacity` per rank and the next gathers want more room)."""
    world = dist.get_world_size(group)
    dev = visible_bytes.device

    def buffers(cap):
        if isinstance(count, torch.Tensor):
            have = visible_bytes.numel() // 152
            if have >= cap:
                buf = visible_bytes[: cap * 152].view(cap, 152).contiguous()
            else:
                buf = torch.zeros((cap, 152), dtype=torch.uint8, device=dev)
                buf[:have] = visible_bytes[: have * 152].view(have, 152)
            cnt = count.to(torch.int64).reshape(1).clone()
        else:
            n = min(int(count), cap)
            buf = torch.zeros((cap, 152), dtype=torch.uint8, device=dev)
            buf[:n] = visible_bytes[: n * 152].view(n, 152)
            cnt = torch.tensor([int(count)], dtype=torch.int64, device=dev)
        return buf, cnt

    buf, cnt = buffers(capacity)
    bufs = [torch.empty_like(buf) for _ in range(world)]
    cnts = [torch.zeros_like(cnt) for _ in range(world)]

    def complete():
        counts = torch.cat(cnts)
        if not check_overflow:
            return torch.stack(bufs), counts
        most = int(counts.max().item())                       # (the frame is over by now: this is the one look at the counts from the host)
        if most > abi.MAX_VISIBLE_NODES:
            raise VisibleOverflow(f"a rank reports {most} visible nodes; the visible-node array holds {abi.MAX_VISIBLE_NODES}")
        if most <= capacity:
            return torch.stack(bufs), counts
        cap = 1 << (most - 1).bit_length()                    # every rank sees the same counts: every rank gathers again, with the same room
        big, _ = buffers(cap)
        bigs = [torch.empty_like(big) for _ in range(world)]
        dist.all_gather(bigs, big, group=group)
        return torch.stack(bigs), counts

    if async_op:
        works = [dist.all_gather(bufs, buf, group=group, async_op=True), dist.all_gather(cnts, cnt, group=group, async_op=True)]

        def finish():
            for w in works:
                w.wait()
            return complete()
        return finish
    dist.all_gather(bufs, buf, group=group)
    dist.all_gather(cnts, cnt, group=group)
    return complete()
