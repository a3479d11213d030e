"""Procedural stand-ins for BASELINE.json's configs.

The reference's assets are stripped from the checkout (`.MISSING_LARGE_BLOBS:1-3`:
assets/models/test_scene.glb and two textures), so every benchmark scene is generated here,
deterministically, as float32 arrays that are handed unchanged to the C-ABI (and, in tests, to the
oracle).  All geometry is glTF-convention: y up, right-handed, CCW front faces.
"""
from __future__ import annotations

import math
from typing import List, Tuple

import numpy as np

from .scene import MESH_VERTEX, CameraDesc, InstanceDesc, Material, MeshDesc, SceneDesc

_ID_Q = (1.0, 0.0, 0.0, 0.0)


def _verts(pos, nrm, tan, uv) -> np.ndarray:
    n = len(pos)
    v = np.zeros(n, MESH_VERTEX)
    v["position"] = np.asarray(pos, np.float32)
    v["normal"] = np.asarray(nrm, np.float32)
    t = np.asarray(tan, np.float32)
    if t.shape[-1] == 3:
        t = np.concatenate([t, np.ones((n, 1), np.float32)], axis=1)
    v["tangent"] = t
    v["texCoords"] = np.asarray(uv, np.float32)
    return v


def grid_patch(origin, du, dv, nu: int, nv: int, displace=None) -> Tuple[np.ndarray, np.ndarray]:
    """(nu × nv)-cell parallelogram patch: origin + s·du + t·dv, normal = normalize(du × dv).
    `displace(P[n,3]) -> h[n]` moves vertices along the normal (normals recomputed by finite differences)."""
    o = np.asarray(origin, np.float64)
    du = np.asarray(du, np.float64)
    dv = np.asarray(dv, np.float64)
    s, t = np.meshgrid(np.linspace(0.0, 1.0, nu + 1), np.linspace(0.0, 1.0, nv + 1), indexing="xy")
    s = s.reshape(-1)
    t = t.reshape(-1)
    P = o[None, :] + s[:, None] * du[None, :] + t[:, None] * dv[None, :]
    n = np.cross(du, dv)
    n /= np.linalg.norm(n)
    N = np.repeat(n[None, :], P.shape[0], axis=0)
    tu = du / np.linalg.norm(du)
    if displace is not None:
        h = displace(P)
        eps = 1e-3
        hu = displace(P + eps * tu[None, :])
        tv = dv / np.linalg.norm(dv)
        hv = displace(P + eps * tv[None, :])
        P = P + h[:, None] * n[None, :]
        gu = (hu - h) / eps
        gv = (hv - h) / eps
        N = n[None, :] - gu[:, None] * tu[None, :] - gv[:, None] * tv[None, :]
        N /= np.linalg.norm(N, axis=1, keepdims=True)
    T = np.repeat(tu[None, :], P.shape[0], axis=0)
    uv = np.stack([s, t], axis=1)
    i0 = (np.arange(nv)[:, None] * (nu + 1) + np.arange(nu)[None, :]).reshape(-1)
    i1 = i0 + 1
    i2 = i0 + (nu + 1)
    i3 = i2 + 1
    idx = np.stack([i0, i1, i3, i0, i3, i2], axis=1).reshape(-1).astype(np.uint32)  # CCW seen from +n
    return _verts(P, N, T, uv), idx


def uv_sphere(nu: int, nv: int, radius: float = 1.0) -> Tuple[np.ndarray, np.ndarray]:
    """nu longitudinal × nv latitudinal segments; 2·nu·(nv−1) triangles (poles are fans)."""
    pos, nrm, tan, uv = [], [], [], []
    for j in range(nv + 1):
        th = math.pi * j / nv
        for i in range(nu + 1):
            ph = 2.0 * math.pi * i / nu
            n = (math.sin(th) * math.cos(ph), math.cos(th), math.sin(th) * math.sin(ph))
            pos.append([radius * c for c in n])
            nrm.append(n)
            tan.append((-math.sin(ph), 0.0, math.cos(ph)))
            uv.append((i / nu, j / nv))
    idx = []
    for j in range(nv):
        for i in range(nu):
            a = j * (nu + 1) + i
            b = a + 1
            c = a + (nu + 1)
            d = c + 1
            if j != 0:
                idx += [a, b, c]
            if j != nv - 1:
                idx += [b, d, c]
    return _verts(pos, nrm, tan, uv), np.asarray(idx, np.uint32)


def cylinder(sides: int, segs: int, radius: float, height: float) -> Tuple[np.ndarray, np.ndarray]:
    """Open cylinder around +y from y=0 to y=height; 2·sides·segs triangles, outward normals."""
    ph = 2.0 * np.pi * np.arange(sides + 1) / sides
    y = height * np.arange(segs + 1) / segs
    PH, Y = np.meshgrid(ph, y, indexing="xy")
    PH = PH.reshape(-1)
    Y = Y.reshape(-1)
    N = np.stack([np.cos(PH), np.zeros_like(PH), np.sin(PH)], axis=1)
    P = N * radius
    P[:, 1] = Y
    T = np.stack([-np.sin(PH), np.zeros_like(PH), np.cos(PH)], axis=1)
    uv = np.stack([PH / (2.0 * np.pi), Y / height], axis=1)
    i0 = (np.arange(segs)[:, None] * (sides + 1) + np.arange(sides)[None, :]).reshape(-1)
    i1 = i0 + 1
    i2 = i0 + (sides + 1)
    i3 = i2 + 1
    idx = np.stack([i0, i2, i3, i0, i3, i1], axis=1).reshape(-1).astype(np.uint32)
    return _verts(P, N, T, uv), idx


def half_torus(major: float, minor: float, seg_major: int, seg_minor: int) -> Tuple[np.ndarray, np.ndarray]:
    """Arch: upper half of a torus in the xy-plane (spans x ∈ [−major, major], apex at y = major)."""
    a = np.pi * np.arange(seg_major + 1) / seg_major
    b = 2.0 * np.pi * np.arange(seg_minor + 1) / seg_minor
    A, B = np.meshgrid(a, b, indexing="xy")
    A = A.reshape(-1)
    B = B.reshape(-1)
    cx, cy = np.cos(A), np.sin(A)
    N = np.stack([np.cos(B) * cx, np.cos(B) * cy, np.sin(B)], axis=1)
    P = np.stack([major * cx, major * cy, np.zeros_like(A)], axis=1) + minor * N
    T = np.stack([-cy, cx, np.zeros_like(A)], axis=1)
    uv = np.stack([A / np.pi, B / (2.0 * np.pi)], axis=1)
    i0 = (np.arange(seg_minor)[:, None] * (seg_major + 1) + np.arange(seg_major)[None, :]).reshape(-1)
    i1 = i0 + 1
    i2 = i0 + (seg_major + 1)
    i3 = i2 + 1
    idx = np.stack([i0, i1, i3, i0, i3, i2], axis=1).reshape(-1).astype(np.uint32)
    return _verts(P, N, T, uv), idx


def _quad(a, b, c, d) -> Tuple[np.ndarray, np.ndarray]:
    """Quad a,b,c,d CCW seen from its front; 2 triangles (a,b,c),(a,c,d)."""
    a, b, c, d = (np.asarray(p, np.float64) for p in (a, b, c, d))
    n = np.cross(b - a, c - a)
    n /= np.linalg.norm(n)
    t = (b - a) / np.linalg.norm(b - a)
    v = _verts([a, b, c, d], [n] * 4, [t] * 4, [(0, 0), (1, 0), (1, 1), (0, 1)])
    return v, np.asarray([0, 1, 2, 0, 2, 3], np.uint32)


# ----------------------------------------------------------------------------------------------
def cornell_box() -> SceneDesc:
    """Config 1: 5 Lambert wall quads + 1 ceiling emitter quad = 12 triangles,
    box [−1,1]³ open towards +z, camera on +z looking down −z, fovY 40°."""
    mats = [
        Material((0.73, 0.73, 0.73, 1.0), 0.0, 1.0),
        Material((0.65, 0.05, 0.05, 1.0), 0.0, 1.0),
        Material((0.12, 0.45, 0.15, 1.0), 0.0, 1.0),
        Material((0.0, 0.0, 0.0, 1.0), 0.0, 1.0, (15.0, 15.0, 15.0)),
    ]
    q = []
    q.append((_quad((-1, -1, 1), (1, -1, 1), (1, -1, -1), (-1, -1, -1)), 0))  # floor, normal +y
    q.append((_quad((-1, 1, -1), (1, 1, -1), (1, 1, 1), (-1, 1, 1)), 0))  # ceiling, normal −y
    q.append((_quad((-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1)), 0))  # back, normal +z
    q.append((_quad((-1, -1, 1), (-1, -1, -1), (-1, 1, -1), (-1, 1, 1)), 1))  # left (x=−1), normal +x
    q.append((_quad((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1)), 2))  # right (x=+1), normal −x
    e = 0.25
    q.append((_quad((-e, 0.998, -e), (e, 0.998, -e), (e, 0.998, e), (-e, 0.998, e)), 3))  # light, normal −y
    meshes = [MeshDesc(v, i, m) for (v, i), m in q]
    inst = [InstanceDesc(k, (0.0, 0.0, 0.0), _ID_Q, (1.0, 1.0, 1.0)) for k in range(len(meshes))]
    d = 1.0 / math.tan(math.radians(20.0))
    cam = CameraDesc((0.0, 0.0, 1.0 + d), (0.0, 0.0, 0.0), math.radians(40.0), 1.0)
    return SceneDesc(mats, meshes, inst, cam, "cornell")


def sphere_scene(nu: int = 100, nv: int = 51) -> SceneDesc:
    """Config 2: one ~10 k-triangle UV sphere (metallic 1, roughness 0.3, base (0.9,0.6,0.2)) instanced
    with a rotation and a non-uniform scale (exercises R3's inverse-transpose), a diffuse ground quad
    grid and one emissive quad.  100×51 segments → 2·100·50 = 10 000 sphere triangles."""
    mats = [
        Material((0.9, 0.6, 0.2, 1.0), 1.0, 0.3),
        Material((0.6, 0.6, 0.6, 1.0), 0.0, 1.0),
        Material((0.0, 0.0, 0.0, 1.0), 0.0, 1.0, (12.0, 11.0, 10.0)),
        Material((0.2, 0.3, 0.8, 1.0), 0.0, 0.5),
    ]
    sv, si = uv_sphere(nu, nv, 1.0)
    gv, gi = grid_patch((-10.0, -1.0, 10.0), (20.0, 0.0, 0.0), (0.0, 0.0, -20.0), 8, 8)
    lv, li = _quad((-2.0, 4.0, -2.0), (2.0, 4.0, -2.0), (2.0, 4.0, 2.0), (-2.0, 4.0, 2.0))  # normal −y
    bv, bi = uv_sphere(24, 12, 0.4)
    meshes = [MeshDesc(sv, si, 0), MeshDesc(gv, gi, 1), MeshDesc(lv, li, 2), MeshDesc(bv, bi, 3)]
    a = math.radians(30.0)
    inst = [
        InstanceDesc(0, (0.0, -0.2, 0.0), (math.cos(a / 2), 0.0, math.sin(a / 2), 0.0), (1.0, 0.8, 1.0)),
        InstanceDesc(1, (0.0, 0.0, 0.0), _ID_Q, (1.0, 1.0, 1.0)),
        InstanceDesc(2, (0.0, 0.0, 0.0), _ID_Q, (1.0, 1.0, 1.0)),
        InstanceDesc(3, (1.6, -0.6, 0.8), _ID_Q, (1.0, 1.0, 1.0)),
        InstanceDesc(3, (-1.7, -0.6, 0.6), _ID_Q, (1.0, 1.0, 1.0)),
    ]
    cam = CameraDesc((0.0, 1.2, 4.5), (0.0, -0.1, 0.0), math.radians(45.0), 1.0)
    return SceneDesc(mats, meshes, inst, cam, "sphere10k")


def atrium(scale: float = 1.0) -> SceneDesc:
    """Config 3: Sponza-class procedural atrium, 250 000 ± 1 % triangles at scale=1 (tessellation
    counts scale with `scale` for small test versions).  40×12×20 m box shell (tessellated walls,
    ceiling), displaced floor, an 8×4 grid of 64-sided columns (two meshes, 32 instances), tube arches
    between columns along x, emissive ceiling panels.  6 material classes: Lambert, GGX dielectric at
    roughness 0.6 / 0.3 / 0.15, metal, emitter.  Every mesh has < 65 536 vertices (u16-expressible)."""
    sc = max(scale, 1e-3)

    def n_(x, lo=1):
        return max(lo, int(round(x * math.sqrt(sc))))

    mats = [
        Material((0.70, 0.68, 0.62, 1.0), 0.0, 1.0),  # 0 walls / ceiling: Lambert
        Material((0.50, 0.40, 0.35, 1.0), 0.0, 0.6),  # 1 floor: GGX rough 0.6
        Material((0.80, 0.80, 0.75, 1.0), 0.0, 0.3),  # 2 columns: GGX rough 0.3
        Material((0.60, 0.62, 0.70, 1.0), 0.0, 0.15),  # 3 arches: GGX rough 0.15
        Material((1.00, 0.78, 0.34, 1.0), 1.0, 0.2),  # 4 metal columns
        Material((0.0, 0.0, 0.0, 1.0), 0.0, 1.0, (20.0, 18.0, 15.0)),  # 5 emitter panels
    ]
    X, Y, Z = 20.0, 12.0, 10.0
    meshes: List[MeshDesc] = []
    inst: List[InstanceDesc] = []

    def add(vi, mat, t=(0.0, 0.0, 0.0), q=_ID_Q, s=(1.0, 1.0, 1.0)):
        meshes.append(MeshDesc(vi[0], vi[1], mat))
        inst.append(InstanceDesc(len(meshes) - 1, t, q, s))
        return len(meshes) - 1

    def bump(P):
        return 0.04 * np.sin(3.1 * P[:, 0]) * np.cos(2.3 * P[:, 2]) + 0.02 * np.sin(11.0 * P[:, 0] + 1.7 * P[:, 2])

    # floor (normal +y): du × dv = +y  →  du = +x, dv = −z
    add(grid_patch((-X, 0.0, Z), (2 * X, 0.0, 0.0), (0.0, 0.0, -2 * Z), n_(264), n_(132), bump), 1)
    # ceiling (normal −y)
    add(grid_patch((-X, Y, -Z), (2 * X, 0.0, 0.0), (0.0, 0.0, 2 * Z), n_(64), n_(32)), 0)
    # walls, normals pointing inside
    add(grid_patch((-X, 0.0, -Z), (2 * X, 0.0, 0.0), (0.0, Y, 0.0), n_(64), n_(24)), 0)  # back  z=−Z, n=+z
    add(grid_patch((X, 0.0, Z), (-2 * X, 0.0, 0.0), (0.0, Y, 0.0), n_(64), n_(24)), 0)  # front z=+Z, n=−z
    add(grid_patch((-X, 0.0, Z), (0.0, 0.0, -2 * Z), (0.0, Y, 0.0), n_(32), n_(24)), 0)  # left  x=−X, n=+x
    add(grid_patch((X, 0.0, -Z), (0.0, 0.0, 2 * Z), (0.0, Y, 0.0), n_(32), n_(24)), 0)  # right x=+X, n=−x
    # columns: 8 × 4 grid, two shared meshes (dielectric / metal), instanced with per-column yaw + scale
    col_h = 8.0
    col = cylinder(n_(64, 8), n_(32, 2), 0.45, col_h)
    meshes.append(MeshDesc(col[0], col[1], 2))
    m_col = len(meshes) - 1
    meshes.append(MeshDesc(col[0], col[1], 4))
    m_colm = len(meshes) - 1
    xs = [-17.5 + 5.0 * i for i in range(8)]
    zs = [-7.5 + 5.0 * k for k in range(4)]
    for k, z in enumerate(zs):
        for i, x in enumerate(xs):
            yaw = 0.37 * (i + 8 * k)
            q = (math.cos(yaw / 2), 0.0, math.sin(yaw / 2), 0.0)
            metal = (i + k) % 4 == 0
            sxz = 1.0 + 0.1 * ((i * 3 + k) % 3)
            inst.append(InstanceDesc(m_colm if metal else m_col, (x, 0.0, z), q, (sxz, 1.0, sxz)))
    # arches between neighbouring columns along x: one mesh, 7 × 4 instances
    arch = half_torus(2.5, 0.22, n_(40, 4), n_(16, 4))
    meshes.append(MeshDesc(arch[0], arch[1], 3))
    m_arch = len(meshes) - 1
    for z in zs:
        for i in range(7):
            inst.append(InstanceDesc(m_arch, (xs[i] + 2.5, col_h, z), _ID_Q, (1.0, 1.0, 1.0)))
    # emissive ceiling panels (normal −y), 2 × 4
    for k in range(2):
        for i in range(4):
            cx, cz = -15.0 + 10.0 * i, -5.0 + 10.0 * k
            add(_quad((cx - 1.5, Y - 0.02, cz - 1.0), (cx + 1.5, Y - 0.02, cz - 1.0), (cx + 1.5, Y - 0.02, cz + 1.0), (cx - 1.5, Y - 0.02, cz + 1.0)), 5)
    cam = CameraDesc((-18.5, 2.6, 1.2), (0.0, 3.4, -0.4), math.radians(60.0), 16.0 / 9.0)
    return SceneDesc(mats, meshes, inst, cam, "atrium")


def two_triangles_and_sphere() -> SceneDesc:
    """Raster-compat fixture scene (SURVEY §8c fixture 3): two triangles + a sphere in front of the
    reference's default camera (position 0, direction −z, fovY π/2 — CameraController.hpp:25-40)."""
    mats = [Material((0.8, 0.3, 0.2, 1.0), 0.0, 1.0), Material((0.2, 0.5, 0.9, 0.5), 0.0, 1.0)]
    qv, qi = _quad((-2.0, -1.5, -4.0), (2.0, -1.5, -4.0), (2.0, 1.5, -4.0), (-2.0, 1.5, -4.0))
    sv, si = uv_sphere(32, 16, 0.8)
    meshes = [MeshDesc(qv, qi, 0), MeshDesc(sv, si, 1)]
    inst = [
        InstanceDesc(0, (0.0, 0.0, 0.0), _ID_Q, (1.0, 1.0, 1.0)),
        InstanceDesc(1, (0.3, -0.2, -2.5), (math.cos(0.3), math.sin(0.3), 0.0, 0.0), (1.0, 1.2, 0.9)),
    ]
    cam = CameraDesc((0.0, 0.0, 0.0), (0.0, 0.0, -1.0), math.pi / 2, 1.0)
    return SceneDesc(mats, meshes, inst, cam, "two_tris_sphere")




# ----------------------------------------------------------------------------------------------
# Config 5: textures + image-based environment light (all procedural and seeded: the reference's
# assets/textures/rusty_metal_grid_{diff,nor_gl}_1k.png are stripped from the checkout).
def _value_noise(size: int, seed: int, octaves: int = 4) -> np.ndarray:
    """Tileable value noise in [0,1], size × size (REPEAT-safe: lattice wraps)."""
    rng = np.random.default_rng(seed)
    out = np.zeros((size, size), np.float64)
    amp, total = 1.0, 0.0
    for o in range(octaves):
        cells = 4 << o
        lat = rng.random((cells, cells))
        t = np.arange(size) * cells / size
        i0 = np.floor(t).astype(int) % cells
        i1 = (i0 + 1) % cells
        f = t - np.floor(t)
        f = f * f * (3 - 2 * f)
        a = lat[i0][:, i0] * (1 - f)[None, :] + lat[i0][:, i1] * f[None, :]
        b = lat[i1][:, i0] * (1 - f)[None, :] + lat[i1][:, i1] * f[None, :]
        out += amp * (a * (1 - f)[:, None] + b * f[:, None])
        total += amp
        amp *= 0.5
    return out / total


def texture_albedo(size: int = 256, seed: int = 5) -> np.ndarray:
    n1, n2 = _value_noise(size, seed), _value_noise(size, seed + 1)
    rgb = np.stack([0.55 + 0.4 * n1, 0.35 + 0.4 * n2, 0.25 + 0.3 * n1 * n2], -1)
    grid = ((np.arange(size) // max(1, size // 8)) % 2)[:, None] ^ ((np.arange(size) // max(1, size // 8)) % 2)[None, :]
    rgb *= (0.75 + 0.25 * grid)[..., None]
    return np.concatenate([np.clip(rgb * 255 + 0.5, 0, 255).astype(np.uint8), np.full((size, size, 1), 255, np.uint8)], -1)


def texture_normal(size: int = 256, seed: int = 6, strength: float = 2.0) -> np.ndarray:
    h = _value_noise(size, seed)
    dx = (np.roll(h, -1, 1) - np.roll(h, 1, 1)) * 0.5 * size / 64.0 * strength
    dy = (np.roll(h, -1, 0) - np.roll(h, 1, 0)) * 0.5 * size / 64.0 * strength
    n = np.stack([-dx, -dy, np.ones_like(h)], -1)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    return np.concatenate([np.clip((n * 0.5 + 0.5) * 255 + 0.5, 0, 255).astype(np.uint8), np.full((size, size, 1), 255, np.uint8)], -1)


def texture_metal_rough(size: int = 256, seed: int = 7) -> np.ndarray:
    r = 0.15 + 0.75 * _value_noise(size, seed)
    m = (_value_noise(size, seed + 1, 2) > 0.55).astype(np.float64)
    z = np.zeros_like(r)
    return np.clip(np.stack([z, r, m, z + 1.0], -1) * 255 + 0.5, 0, 255).astype(np.uint8)   # glTF: G = roughness, B = metallic


def analytic_sky(w: int = 64, h: int = 32, sun_dir=(0.4, 0.7, 0.3), sun_radiance: float = 60.0) -> np.ndarray:
    """Lat-long RGB32F sky (row 0 = +y): horizon-to-zenith gradient, dark ground, and a small bright sun (which is what makes
    importance sampling necessary)."""
    v = (np.arange(h) + 0.5) / h
    u = (np.arange(w) + 0.5) / w
    th = np.pi * v[:, None]
    ph = 2 * np.pi * u[None, :] - np.pi
    d = np.stack([np.sin(th) * np.cos(ph), np.cos(th) * np.ones_like(ph), np.sin(th) * np.sin(ph)], -1)
    up = np.clip(d[..., 1], 0, 1)
    sky = np.stack([0.35 + 0.25 * (1 - up), 0.5 + 0.25 * (1 - up), 0.9 - 0.2 * (1 - up)], -1) * (0.6 + 0.8 * up[..., None])
    ground = np.array([0.12, 0.10, 0.08])
    img = np.where((d[..., 1] > 0)[..., None], sky, ground[None, None, :])
    s = np.asarray(sun_dir, np.float64)
    s /= np.linalg.norm(s)
    img = img + (np.einsum("ijk,k->ij", d, s) > np.cos(np.radians(6.0)))[..., None] * sun_radiance * np.array([1.0, 0.9, 0.7])
    return img.astype(np.float32)


def textured_atrium(scale: float = 1.0, tex_size: int = 1024, env_size=(2048, 1024), keep_panels: bool = True) -> SceneDesc:
    """Config 5: the atrium with albedo / normal / metallic-roughness textures on floor, columns and arches, its ceiling opened to an
    analytic-sky environment map (the emissive panels stay unless keep_panels=False: both light kinds are then sampled)."""
    d = atrium(scale)
    d.name = "textured_atrium"
    d.textures = [texture_albedo(tex_size, 5), texture_normal(tex_size, 6), texture_metal_rough(tex_size, 7)]
    for k in (1, 2, 3):                               # floor, columns, arches
        d.materials[k].tex_color, d.materials[k].tex_normal = 0, 1
    d.materials[3].tex_mr = 2
    d.materials[3].metallic, d.materials[3].roughness = 1.0, 1.0
    d.env = analytic_sky(env_size[0], env_size[1])
    # open the roof: drop the ceiling patch (mesh 1) and, optionally, the emissive panels
    drop = {1} | (set() if keep_panels else {i for i, m in enumerate(d.meshes) if m.material == 5})
    d.instances = [it for it in d.instances if it.mesh not in drop]
    return d


def textured_objects() -> SceneDesc:
    """Small textured + environment-lit test scene: a textured ground, a normal-mapped sphere with a metallic-roughness map and a
    plain diffuse sphere under the analytic sky; no emissive triangles (environment NEE only)."""
    d = sphere_scene(48, 25)
    d.name = "textured_objects"
    d.textures = [texture_albedo(64, 5), texture_normal(64, 6), texture_metal_rough(64, 7)]
    d.materials[0] = Material((1.0, 1.0, 1.0, 1.0), 1.0, 1.0, (0, 0, 0), 0, 1, 2)       # big sphere: all three maps
    d.materials[1] = Material((0.9, 0.9, 0.9, 1.0), 0.0, 1.0, (0, 0, 0), 0, -1, -1)     # ground: albedo map only (stays Lambert class)
    d.materials[3] = Material((0.2, 0.3, 0.8, 0.5), 0.0, 0.5, (0, 0, 0), -1, 1, -1)     # small spheres: normal map only
    d.instances = [it for it in d.instances if it.mesh != 2]                            # no emissive quad
    d.env = analytic_sky(64, 32)
    return d

def by_name(name: str, **kw) -> SceneDesc:
    return {"cornell": cornell_box, "sphere10k": sphere_scene, "atrium": atrium, "two_tris_sphere": two_triangles_and_sphere,
            "textured_objects": textured_objects, "textured_atrium": textured_atrium}[name](**kw)
