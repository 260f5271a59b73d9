"""Seeded parity cases shared by the golden generator, the oracle pin tests and the GPU parity tests."""
import numpy as np

from simlod_amd import abi, camera, synthetic

W = H = 256


def _cam(box):
    return camera.lookat_transform((1.8 * box[0], -1.2 * box[1], 1.4 * max(box)), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2]), W, H)


def case(name):
    """-> (points, box, batch, transform)"""
    if name == "uniform_3x40k":            # root splits in batch 2 with 40 000 stored points -> spill copy + chunk recycling
        pts, box = synthetic.uniform_cube(120_000, seed=1)
        return pts, box, 40_000, _cam(box)
    if name == "hotspot_150k":             # every point in one level-3 cell: 4+ expand rounds inside one batch
        pts, box = synthetic.hotspot(150_000, seed=11, level=3, cell=(5, 2, 6))
        return pts, box, 150_000, _cam(box)
    if name == "terrain_4x100k":           # swath-ordered surface, ragged leaf populations, several split generations
        pts, box = synthetic.terrain(400_000, seed=3, box=(600.0, 400.0, 40.0), tile=50.0)
        return pts, box, 100_000, _cam(box)
    if name == "ragged_tiny":              # batches of 1, 7 and 49 999 + an EMPTY batch, then the 50 001st point
        pts, box = synthetic.uniform_cube(50_010, seed=5)
        return pts, box, None, _cam(box)   # batch boundaries: see ragged_batches()
    raise KeyError(name)


CASES = ["uniform_3x40k", "hotspot_150k", "terrain_4x100k", "ragged_tiny"]


def batches_of(name, pts, batch):
    if name == "ragged_tiny":
        cuts = [0, 1, 8, 8, 50_000, 50_001, 50_010]      # 1, 7, 0 (empty), 49 992, 1 (crosses the 50 000 limit), 9
        return [pts[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    return [pts[i:i + batch] for i in range(0, len(pts), batch)]


def uniforms_for(box, T, *, hqs=False, persistent=1 << 30, momentary=300_000_000, point_size=1):
    return abi.make_uniforms(W, H, T, box, persistent_capacity=persistent, momentary_capacity=momentary, hqs=hqs, point_size=point_size)
