"""numpy mirrors of include/simlod_abi.h (the SimLOD Node/Chunk/Point/Uniforms/Stats layout).

Reference definitions: modules/progressive_octree/structures.cuh:21-143 and
modules/progressive_octree/HostDeviceInterface.h:6-71.  Sizes and offsets are asserted at import time
against the numbers pinned in include/simlod_abi.h.
"""
import numpy as np

MAX_POINTS_PER_NODE = 50_000
POINTS_PER_CHUNK = 1000
GRID_SIZE = 128
GRID_NUM_WORDS = GRID_SIZE ** 3 // 32
MAX_DEPTH = 20
BATCH_STREAM_SIZE = 50
MAX_BATCH_SIZE = 1_000_000
MAX_BATCHES_PER_LAUNCH = 20
MAX_VISIBLE_NODES = 100_000
CLEAR_PIXEL = (0x7F800000 << 32) | 0x00332211
NODE_BYTES_PER_SLOT_HOST = 200          # main_progressive_octree.cpp:552 sizes the node array as 200 000 x 200 B
CHUNK_BYTES = 16016
GRID_BYTES = 262144

point_dtype = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("color", "<u4")])

node_dtype = np.dtype({
    "names": ["children", "counter", "numPoints", "level", "X", "Y", "Z", "countIteration", "countFlag",
              "name", "visible", "isFiltered", "isLeaf", "isLarge", "grid", "points", "voxelChunks",
              "numVoxels", "numVoxelsStored"],
    "formats": [("<u8", 8), "<u4", "<u4", "<u4", "<u4", "<u4", "<u4", "<u4", "<u4",
                ("u1", 20), "u1", "u1", "u1", "u1", "<u8", "<u8", "<u8", "<u4", "<u4"],
    "offsets": [0, 64, 68, 72, 76, 80, 84, 88, 92, 96, 116, 117, 118, 119, 120, 128, 136, 144, 148],
    "itemsize": 152,
})

mat4_dtype = np.dtype(("<f4", (4, 4)))   # rows[i] = matrix row i

uniforms_dtype = np.dtype({
    "names": ["width", "height", "time", "fovy_rad", "world", "view", "proj", "transform",
              "transform_updateBound", "transformInv_updateBound", "persistentBufferCapacity",
              "momentaryBufferCapacity", "frameCounter", "boxMin", "boxMax", "showBoundingBox", "showPoints",
              "colorByNode", "colorByLOD", "colorWhite", "doUpdateVisibility", "doProgressive", "LOD",
              "useHighQualityShading", "minNodeSize", "pointSize", "updateStats", "enableEDL", "edlStrength"],
    "formats": ["<f4", "<f4", "<f4", "<f4", mat4_dtype, mat4_dtype, mat4_dtype, mat4_dtype, mat4_dtype, mat4_dtype,
                "<u8", "<u8", "<u8", ("<f4", 3), ("<f4", 3), "u1", "u1", "u1", "u1", "u1", "u1", "u1", "<f4",
                "u1", "<f4", "<i4", "u1", "u1", "<f4"],
    "offsets": [0, 4, 8, 12, 16, 80, 144, 208, 272, 336, 400, 408, 416, 424, 436, 448, 449, 450, 451, 452, 453,
                454, 456, 460, 464, 468, 472, 473, 476],
    "itemsize": 480,
})

stats_dtype = np.dtype({
    "names": ["frameID", "numNodes", "numInner", "numLeaves", "numNonemptyLeaves", "numPoints", "numVoxels",
              "allocatedBytes_momentary", "allocatedBytes_persistent", "numVisibleNodes", "numVisibleInner",
              "numVisibleLeaves", "numVisiblePoints", "numVisibleVoxels", "numChunksPoints", "numChunksVoxels",
              "batchletIndex", "numPointsProcessed", "numAllocatedChunks", "chunkPoolSize", "dbg",
              "memCapacityReached"],
    "formats": ["<u4"] * 7 + ["<u8", "<u8"] + ["<u4"] * 8 + ["<u8", "<u8", "<u8", "<u4", "u1"],
    "offsets": [0, 4, 8, 12, 16, 20, 24, 32, 40, 48, 52, 56, 60, 64, 68, 72, 76, 80, 88, 96, 104, 108],
    "itemsize": 112,
})

assert point_dtype.itemsize == 16 and node_dtype.itemsize == 152
assert uniforms_dtype.itemsize == 480 and stats_dtype.itemsize == 112


def alloc_round(size: int) -> int:
    """AllocatorGlobal::alloc rounding, utils.h.cu:190."""
    return 16 * ((size + 16) // 16)


def make_uniforms(width, height, transform, box_size, *, transform_update_bound=None,
                  persistent_capacity=0, momentary_capacity=0, frame_counter=0, point_size=1,
                  min_node_size=64.0, hqs=False, show_points=True, color_by_node=False, color_by_lod=False,
                  show_bounding_box=False, fovy_deg=60.0):
    """Fill a Uniforms record the way getUniforms() does (main_progressive_octree.cpp:283-331).

    `transform` is the ROW-MAJOR 4x4 world-view-projection matrix (rows[i] = row i, i.e. what the host
    obtains after glm::transpose).  boxMin is always 0 and boxMax the bounding-box size (:312-313).
    """
    u = np.zeros((), dtype=uniforms_dtype)
    t = np.asarray(transform, dtype=np.float32).reshape(4, 4)
    tu = t if transform_update_bound is None else np.asarray(transform_update_bound, np.float32).reshape(4, 4)
    ident = np.eye(4, dtype=np.float32)
    u["width"], u["height"] = float(width), float(height)
    u["fovy_rad"] = np.float32(3.1415) * np.float32(fovy_deg) / np.float32(180.0)
    u["world"], u["view"], u["proj"] = ident, ident, ident
    u["transform"], u["transform_updateBound"] = t, tu
    with np.errstate(all="ignore"):
        try:
            u["transformInv_updateBound"] = np.linalg.inv(tu.astype(np.float64)).astype(np.float32)
        except np.linalg.LinAlgError:
            u["transformInv_updateBound"] = ident
    u["persistentBufferCapacity"] = persistent_capacity
    u["momentaryBufferCapacity"] = momentary_capacity
    u["frameCounter"] = frame_counter
    u["boxMin"] = (0.0, 0.0, 0.0)
    u["boxMax"] = tuple(float(v) for v in box_size)
    u["showBoundingBox"] = show_bounding_box
    u["showPoints"] = show_points
    u["colorByNode"], u["colorByLOD"] = color_by_node, color_by_lod
    u["doUpdateVisibility"] = 1
    u["LOD"] = 0.2
    u["useHighQualityShading"] = hqs
    u["minNodeSize"] = min_node_size
    u["pointSize"] = point_size
    u["enableEDL"] = 1
    u["edlStrength"] = 0.8
    return u
