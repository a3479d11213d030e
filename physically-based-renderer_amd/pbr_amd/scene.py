"""Host-side scene model — the Python mirror of the reference's scene/mesh interface.

Names and argument meaning follow the reference so that code written against it reads the same:

  MESH_VERTEX (dtype)   pbr::MeshVertex        src/pbr_engine/engine/pbr/MeshVertex.hpp:14-19
  MeshBuilder           pbr::MeshBuilder       src/pbr_engine/engine/pbr/MeshBuilder.hpp:12-37, .cpp:16-55
  PrimitiveSpan         pbr::PrimitiveSpan     src/pbr_engine/engine/pbr/Mesh.hpp:15-21
  Transform             pbr::Transform         src/pbr_engine/engine/pbr/Scene.hpp:19-23
  Node / Scene          pbr::Node / pbr::Scene src/pbr_engine/engine/pbr/Scene.hpp:24-115
  Material              pbr::MaterialData      src/pbr_engine/engine/pbr/Material.hpp:14-16 (+ metal-rough, emissive)
  make_camera_data      pbr::makeCameraData    src/pbr_engine/engine/pbr/CameraData.hpp:22-32

Deliberate deviations: indices are u32 (reference: u16, silently truncated);
parent transforms ARE composed when a Scene is flattened (reference draws each node with its local
TRS only); objects are keyed by index, not glTF name.

This module is pure numpy: it describes scenes, it does not render them.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Iterator, List, Optional, Sequence

import numpy as np

# R1: {vec3 position; vec3 normal; vec4 tangent; vec2 texCoords}, 48 bytes, tightly packed.
MESH_VERTEX = np.dtype(
    [("position", "<f4", (3,)), ("normal", "<f4", (3,)), ("tangent", "<f4", (4,)), ("texCoords", "<f4", (2,))]
)
assert MESH_VERTEX.itemsize == 48


@dataclasses.dataclass
class Material:
    base_color: Sequence[float] = (1.0, 1.0, 1.0, 1.0)  # MaterialData::color
    metallic: float = 0.0
    roughness: float = 1.0
    emissive: Sequence[float] = (0.0, 0.0, 0.0)
    tex_color: int = -1
    tex_normal: int = -1
    tex_mr: int = -1


@dataclasses.dataclass
class Transform:
    position: Sequence[float] = (0.0, 0.0, 0.0)
    rotation: Sequence[float] = (1.0, 0.0, 0.0, 0.0)  # quaternion (w, x, y, z) — gltf/Asset.cpp:242
    scale: Sequence[float] = (1.0, 1.0, 1.0)


@dataclasses.dataclass
class PrimitiveSpan:
    material: int
    firstVertex: int
    vertexCount: int
    firstIndex: int
    indexCount: int


@dataclasses.dataclass
class BuiltMesh:
    vertices: np.ndarray  # MESH_VERTEX[n]
    indices: np.ndarray  # uint32[m], primitive-local (firstVertex is added at draw time)
    primitives: List[PrimitiveSpan]


class MeshBuilder:
    """Concatenates primitives into one vertex array + one index array + spans (MeshBuilder.cpp:16-55)."""

    def __init__(self) -> None:
        self._primitives: list = []

    def addPrimitive(self, vertices: np.ndarray, indices: np.ndarray, material: int = 0) -> "MeshBuilder":
        v = np.ascontiguousarray(vertices, dtype=MESH_VERTEX)
        i = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
        if i.size % 3:
            raise ValueError("triangle list: index count must be a multiple of 3")
        if i.size and int(i.max()) >= v.size:
            raise ValueError("index out of range")
        self._primitives.append((v, i, int(material)))
        return self

    def build(self) -> BuiltMesh:
        verts = [p[0] for p in self._primitives]
        idx = [p[1] for p in self._primitives]
        spans = []
        cv = ci = 0
        for v, i, m in self._primitives:
            spans.append(PrimitiveSpan(m, cv, v.size, ci, i.size))
            cv += v.size
            ci += i.size
        return BuiltMesh(
            np.concatenate(verts) if verts else np.zeros(0, MESH_VERTEX),
            np.concatenate(idx) if idx else np.zeros(0, np.uint32),
            spans,
        )


@dataclasses.dataclass
class Node:
    name: str = ""
    transform: Transform = dataclasses.field(default_factory=Transform)
    mesh: Optional[BuiltMesh] = None
    children: List["Node"] = dataclasses.field(default_factory=list)

    def addChild(self, node: "Node") -> "Node":
        self.children.append(node)
        return node


class Scene:
    def __init__(self) -> None:
        self._nodes: List[Node] = []

    def addNode(self, node: Node) -> Node:
        self._nodes.append(node)
        return node

    def getTopLevelNodes(self) -> List[Node]:
        return self._nodes

    def iterateAllNodes(self) -> Iterator[Node]:
        """Post-order, like Scene.cpp:77-82."""

        def walk(n: Node):
            for c in n.children:
                yield from walk(c)
            yield n

        for n in self._nodes:
            yield from walk(n)


@dataclasses.dataclass
class CameraData:
    view: np.ndarray  # 4x4, column-major like glm (view[col][row])
    proj: np.ndarray
    position: np.ndarray


ZNEAR = 0.01
ZFAR = 1024.0


def make_camera_data(position, target, fov: float, aspect: float) -> CameraData:
    """glm::lookAtRH(position, target, (0,-1,0)) and glm::perspectiveRH_NO(fov, aspect, 0.01, 1024)."""
    p = np.asarray(position, np.float64)
    f = np.asarray(target, np.float64) - p
    f /= np.linalg.norm(f)
    s = np.cross(f, (0.0, -1.0, 0.0))
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    view = np.zeros((4, 4), np.float64)  # view[col][row]
    view[0][0], view[1][0], view[2][0], view[3][0] = s[0], s[1], s[2], -np.dot(s, p)
    view[0][1], view[1][1], view[2][1], view[3][1] = u[0], u[1], u[2], -np.dot(u, p)
    view[0][2], view[1][2], view[2][2], view[3][2] = -f[0], -f[1], -f[2], np.dot(f, p)
    view[3][3] = 1.0
    th = math.tan(fov * 0.5)
    proj = np.zeros((4, 4), np.float64)
    proj[0][0] = 1.0 / (aspect * th)
    proj[1][1] = 1.0 / th
    proj[2][2] = -(ZFAR + ZNEAR) / (ZFAR - ZNEAR)
    proj[2][3] = -1.0
    proj[3][2] = -(2.0 * ZFAR * ZNEAR) / (ZFAR - ZNEAR)
    return CameraData(view.astype(np.float32), proj.astype(np.float32), p.astype(np.float32))


# ----------------------------------------------------------------------------------------------
# Flat description handed across the C-ABI: exactly the argument lists of ptc_add_* (include/ptc.h).


@dataclasses.dataclass
class MeshDesc:
    vertices: np.ndarray  # MESH_VERTEX[n]
    indices: np.ndarray  # uint32[3k], mesh-local
    material: int


@dataclasses.dataclass
class InstanceDesc:
    mesh: int
    t: Sequence[float] = (0.0, 0.0, 0.0)
    q_wxyz: Sequence[float] = (1.0, 0.0, 0.0, 0.0)
    s: Sequence[float] = (1.0, 1.0, 1.0)
    matrix: Optional[Sequence[float]] = None  # 16 floats, column-major (glm layout); overrides t/q/s when given


@dataclasses.dataclass
class CameraDesc:
    position: Sequence[float]
    target: Sequence[float]
    fov_y: float
    aspect: float


@dataclasses.dataclass
class SceneDesc:
    materials: List[Material]
    meshes: List[MeshDesc]
    instances: List[InstanceDesc]
    camera: CameraDesc
    name: str = ""
    textures: List[np.ndarray] = dataclasses.field(default_factory=list)  # RGBA8 images, (h, w, 4) uint8; Material.tex_* index this list
    env: Optional[np.ndarray] = None  # lat-long RGB32F environment map, (h, w, 3) float32, row 0 = +y
    texture_filter: str = "nearest"  # "nearest" (the reference's default-constructed sampler) or "linear" (bilinear, REPEAT)
    bvh_builder: "str | None" = None  # "sah" (binned surface-area splits), "lbvh" (Morton-code radix tree) or None = the context's default

    @property
    def n_triangles(self) -> int:
        return sum(self.meshes[i.mesh].indices.size // 3 for i in self.instances)


def _quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return (
        aw * bw - ax * bx - ay * by - az * bz,
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
    )


def flatten_scene(scene: Scene, materials: List[Material], camera: CameraDesc, compose_parents: bool = True) -> SceneDesc:
    """pbr::Scene → SceneDesc.  One MeshDesc per PrimitiveSpan (firstVertex applied, like
    drawIndexed(..., vertexOffset=firstVertex), PbrRenderSystem.cpp:454-460), one instance per node.

    compose_parents=True composes ancestor transforms, which is only representable as a single TRS when
    every ancestor scale is uniform (otherwise ValueError); False reproduces the reference quirk of
    drawing each node with its local TRS."""
    meshes: List[MeshDesc] = []
    instances: List[InstanceDesc] = []
    cache = {}

    def emit(node: Node, t, q, s):
        if node.mesh is None:
            return
        key = id(node.mesh)
        if key not in cache:
            ids = []
            for sp in node.mesh.primitives:
                v = node.mesh.vertices[sp.firstVertex : sp.firstVertex + sp.vertexCount]
                i = node.mesh.indices[sp.firstIndex : sp.firstIndex + sp.indexCount]
                meshes.append(MeshDesc(v, i, sp.material))
                ids.append(len(meshes) - 1)
            cache[key] = ids
        for mid in cache[key]:
            instances.append(InstanceDesc(mid, tuple(t), tuple(q), tuple(s)))

    def rot(q, v):
        w, x, y, z = q
        qv = np.array([x, y, z], np.float64)
        v = np.asarray(v, np.float64)
        return v + 2.0 * np.cross(qv, np.cross(qv, v) + w * v)

    def walk(node: Node, pt, pq, ps):
        tr = node.transform
        if compose_parents:
            if not (abs(ps[0] - ps[1]) < 1e-12 and abs(ps[1] - ps[2]) < 1e-12):
                raise ValueError("non-uniform parent scale cannot be composed into one TRS")
            t = np.asarray(pt, np.float64) + rot(pq, np.asarray(tr.position, np.float64) * ps[0])
            q = _quat_mul(pq, tr.rotation)
            s = tuple(ps[0] * c for c in tr.scale)
        else:
            t, q, s = tr.position, tr.rotation, tr.scale
        for c in node.children:
            walk(c, t, q, s)
        emit(node, [float(x) for x in t], [float(x) for x in q], [float(x) for x in s])

    for n in scene.getTopLevelNodes():
        walk(n, (0.0, 0.0, 0.0), (1.0, 0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
    return SceneDesc(list(materials), meshes, instances, camera)
