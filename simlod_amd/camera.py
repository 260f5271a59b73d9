"""Host-side camera maths: the recipe the reference host uses to fill Uniforms.transform.

Mirrors (does not copy) include/OrbitControls.h:140-159 (yaw/pitch/radius/target -> world matrix),
include/GLRenderer.h:156-161 (view = inverse(world), glm::perspective(fovy, aspect, 0.1, 2e6)) and
modules/progressive_octree/main_progressive_octree.cpp:283-298 (double matrices are narrowed to fp32,
multiplied in fp32 and stored transposed, so Uniforms.transform.rows[i] is matrix row i).

All matrices here are ROW-MAJOR numpy arrays acting on column vectors.
"""
import math
import numpy as np

NEAR, FAR, FOVY_DEG = 0.1, 2_000_000.0, 60.0

# presets: main_progressive_octree.cpp:1314-1328 (yaw, pitch, radius, target)
PRESETS = {
    "morro_bay_bird": (-0.207, -0.797, 3866.886, (2398.747, 2167.120, -394.165)),
    "morro_bay_close": (-11.270, -0.225, 93.982, (2750.218, 974.775, 76.230)),
}


def perspective(fovy_deg=FOVY_DEG, aspect=1.0, near=NEAR, far=FAR):
    t = math.tan(math.pi * fovy_deg / 180.0 / 2.0)
    p = np.zeros((4, 4), dtype=np.float64)
    p[0, 0] = 1.0 / (aspect * t)
    p[1, 1] = 1.0 / t
    p[2, 2] = -(far + near) / (far - near)
    p[2, 3] = -(2.0 * far * near) / (far - near)
    p[3, 2] = -1.0
    return p


def look_at(eye, target, up=(0.0, 0.0, 1.0)):
    eye, target, up = (np.asarray(v, dtype=np.float64) for v in (eye, target, up))
    f = target - eye
    f /= np.linalg.norm(f)
    s = np.cross(f, up)
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    v = np.eye(4)
    v[0, :3], v[1, :3], v[2, :3] = s, u, -f
    v[0, 3], v[1, 3], v[2, 3] = -s @ eye, -u @ eye, f @ eye
    return v


def _translate(t):
    m = np.eye(4)
    m[:3, 3] = t
    return m


def _rotate(angle, axis):
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    c, s = math.cos(angle), math.sin(angle)
    x, y, z = a
    r = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s, 0],
                  [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s, 0],
                  [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c), 0],
                  [0, 0, 0, 1]], dtype=np.float64)
    return r


def orbit_view(yaw, pitch, radius, target):
    """OrbitControls::update(): world = T(target) * Rz(yaw) * Rx(pitch) * flip * T(0,0,radius); view = world^-1."""
    flip = np.array([[1, 0, 0, 0], [0, 0, -1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.float64)
    world = _translate(target) @ _rotate(yaw, (0, 0, 1)) @ _rotate(pitch, (1, 0, 0)) @ flip @ _translate((0, 0, radius))
    return np.linalg.inv(world)


def world_view_proj(view, proj):
    """fp32 product of the fp32-narrowed matrices, row-major == the transposed glm matrix the host uploads."""
    v32 = np.asarray(view, dtype=np.float64).astype(np.float32)
    p32 = np.asarray(proj, dtype=np.float64).astype(np.float32)
    out = np.zeros((4, 4), dtype=np.float32)
    for i in range(4):
        for j in range(4):
            acc = np.float32(0.0)
            for k in range(4):
                acc = np.float32(acc + np.float32(p32[i, k] * v32[k, j]))
            out[i, j] = acc
    return out


def preset_transform(name, width, height):
    yaw, pitch, radius, target = PRESETS[name]
    return world_view_proj(orbit_view(yaw, pitch, radius, target), perspective(aspect=width / height))


def lookat_transform(eye, target, width, height, up=(0.0, 0.0, 1.0)):
    return world_view_proj(look_at(eye, target, up), perspective(aspect=width / height))
