"""glTF 2.0 / GLB: ctypes binding of the C++ loader (include/ptc_gltf.h, libptc_gltf.so) and a small writer.

`load_into(pt, path)` is the path-tracer counterpart of the reference's `gltf::Loader::loadAsset` +
`Asset::loadScene` (src/pbr_engine/gltf/pbr/gltf/Loader.cpp:10-32, Asset.cpp:259-273): it fills a PathTracer's scene
from a file; the camera stays the caller's (the reference ignores glTF cameras too, Asset.cpp:262-265).

`write_glb(desc, path)` serialises a SceneDesc — used by the tests to produce assets, since the reference's own
(assets/models/test_scene.glb) is stripped from the checkout (.MISSING_LARGE_BLOBS).
"""
from __future__ import annotations

import ctypes as C
import json
import os
import struct

import numpy as np

from . import ptc as _ptc

_LIB = os.path.join(os.path.dirname(_ptc.LIB_PATH), "libptc_gltf.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        _ptc.load_library()
        if not os.path.exists(_LIB):
            raise _ptc.PtcError(f"{_LIB} is missing: run __graft_entry__.build()")
        L = C.CDLL(_LIB)
        L.ptc_gltf_load.restype = C.c_longlong
        L.ptc_gltf_load.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_char_p, C.c_int]
        _lib = L
    return _lib


def load_into(pt, path: str, camera=None, scene_index: int = -1, compose_parents: bool = True):
    """scene_begin → materials/meshes/instances from the file → camera → scene_commit.
    Returns (n_triangles, bbox_lo, bbox_hi).  camera: CameraDesc, or None to frame the bounding box from +z."""
    L = _load()
    h = pt._h
    pt._ck(pt._L.ptc_scene_begin(h))
    bbox = (C.c_float * 6)()
    err = C.create_string_buffer(512)
    n = L.ptc_gltf_load(h, os.fsencode(path), scene_index, 1 if compose_parents else 0, bbox, err, 512)
    if n < 0:
        raise _ptc.PtcError(err.value.decode() or f"ptc_gltf_load failed ({n})")
    lo, hi = np.array(bbox[0:3], np.float32), np.array(bbox[3:6], np.float32)
    if camera is None:
        c = 0.5 * (lo + hi)
        r = 0.5 * float(np.linalg.norm(hi - lo))
        fov = np.radians(60.0)
        pt.set_camera((c[0], c[1], c[2] + r / np.tan(0.5 * fov) + r), c, float(fov), 1.0)
    else:
        pt.set_camera(camera.position, camera.target, camera.fov_y, camera.aspect)
    pt._ck(pt._L.ptc_scene_commit(h))
    return int(n), lo, hi


# ----------------------------------------------------------------------------------------------------
def _pad4(b: bytes, fill: bytes = b"\x00") -> bytes:
    return b + fill * ((4 - len(b) % 4) % 4)


def write_glb(desc, path: str, index_type: str = "auto", interleaved: bool = False, nodes=None) -> None:
    """SceneDesc → GLB.  One glTF mesh per MeshDesc (one primitive each), one node per instance (TRS, or `matrix`
    when the instance carries one).  `nodes`: optional explicit node list (dicts with mesh/translation/rotation/
    scale/matrix/children) + root list, to exercise hierarchies: nodes=(node_dicts, root_indices).
    index_type: "u16" | "u32" | "auto" (u16 when the mesh has < 65536 vertices, like the reference's indices)."""
    blob = bytearray()
    views, accessors, meshes = [], [], []

    def add_view(data: bytes, stride: int = 0, target: int = 0) -> int:
        while len(blob) % 4:
            blob.append(0)
        v = {"buffer": 0, "byteOffset": len(blob), "byteLength": len(data)}
        if stride:
            v["byteStride"] = stride
        if target:
            v["target"] = target
        blob.extend(data)
        views.append(v)
        return len(views) - 1

    def add_acc(view: int, ctype: int, count: int, typ: str, offset: int = 0, minmax=None) -> int:
        a = {"bufferView": view, "componentType": ctype, "count": count, "type": typ}
        if offset:
            a["byteOffset"] = offset
        if minmax is not None:
            a["min"], a["max"] = [float(x) for x in minmax[0]], [float(x) for x in minmax[1]]
        accessors.append(a)
        return len(accessors) - 1

    for m in desc.meshes:
        v = np.ascontiguousarray(m.vertices)
        n = v.size
        pos = np.ascontiguousarray(v["position"], "<f4")
        mm = (pos.min(0), pos.max(0))
        if interleaved:  # one 48-byte-stride view holding the R1 record as is
            bv = add_view(v.tobytes(), stride=48, target=34962)
            attrs = {"POSITION": add_acc(bv, 5126, n, "VEC3", 0, mm), "NORMAL": add_acc(bv, 5126, n, "VEC3", 12),
                     "TANGENT": add_acc(bv, 5126, n, "VEC4", 24), "TEXCOORD_0": add_acc(bv, 5126, n, "VEC2", 40)}
        else:
            attrs = {"POSITION": add_acc(add_view(pos.tobytes(), target=34962), 5126, n, "VEC3", 0, mm),
                     "NORMAL": add_acc(add_view(np.ascontiguousarray(v["normal"], "<f4").tobytes(), target=34962), 5126, n, "VEC3"),
                     "TANGENT": add_acc(add_view(np.ascontiguousarray(v["tangent"], "<f4").tobytes(), target=34962), 5126, n, "VEC4"),
                     "TEXCOORD_0": add_acc(add_view(np.ascontiguousarray(v["texCoords"], "<f4").tobytes(), target=34962), 5126, n, "VEC2")}
        idx = np.ascontiguousarray(m.indices, np.uint32)
        use16 = index_type == "u16" or (index_type == "auto" and n < 65536)
        ib = idx.astype("<u2").tobytes() if use16 else idx.astype("<u4").tobytes()
        ia = add_acc(add_view(ib, target=34963), 5123 if use16 else 5125, idx.size, "SCALAR")
        meshes.append({"primitives": [{"attributes": attrs, "indices": ia, "material": int(m.material), "mode": 4}]})

    mats = []
    for m in desc.materials:
        e = [float(x) for x in m.emissive]
        strength = max(1.0, max(e))
        g = {"pbrMetallicRoughness": {"baseColorFactor": [float(x) for x in m.base_color], "metallicFactor": float(m.metallic), "roughnessFactor": float(m.roughness)},
             "emissiveFactor": [x / strength for x in e]}
        if strength > 1.0:
            g["extensions"] = {"KHR_materials_emissive_strength": {"emissiveStrength": strength}}
        mats.append(g)

    if nodes is None:
        node_list, roots = [], []
        for k, it in enumerate(desc.instances):
            nd = {"mesh": int(it.mesh), "name": f"inst{k}"}
            if getattr(it, "matrix", None) is not None:
                nd["matrix"] = [float(x) for x in np.asarray(it.matrix, np.float32).reshape(16)]
            else:
                q = it.q_wxyz
                nd["translation"] = [float(x) for x in it.t]
                nd["rotation"] = [float(q[1]), float(q[2]), float(q[3]), float(q[0])]  # glTF stores (x,y,z,w)
                nd["scale"] = [float(x) for x in it.s]
            node_list.append(nd)
            roots.append(k)
    else:
        node_list, roots = nodes
    doc = {"asset": {"version": "2.0", "generator": "pbr_amd.gltf.write_glb"}, "scene": 0, "scenes": [{"nodes": list(roots)}], "nodes": node_list,
           "meshes": meshes, "materials": mats, "accessors": accessors, "bufferViews": views, "buffers": [{"byteLength": len(blob)}]}
    if any("extensions" in g for g in mats):
        doc["extensionsUsed"] = ["KHR_materials_emissive_strength"]
    js = _pad4(json.dumps(doc, separators=(",", ":")).encode(), b" ")
    bn = _pad4(bytes(blob))
    with open(path, "wb") as f:
        f.write(struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(js) + 8 + len(bn)))
        f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
        f.write(struct.pack("<II", len(bn), 0x004E4942) + bn)
