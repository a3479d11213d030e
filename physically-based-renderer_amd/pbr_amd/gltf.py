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
import zlib

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


def load_into(pt, path: str, camera=None, scene_index: int = -1, compose_parents: bool = True, env=None):
    """scene_begin → materials/meshes/instances from the file → camera → scene_commit.
    Returns (n_triangles, bbox_lo, bbox_hi).  camera: CameraDesc, or None to frame the bounding box from +z.
    env: optional (h, w, 3) float32 lat-long environment map (glTF has no such thing; ptc_set_env_latlong_rgb32f)."""
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
    if env is not None:
        e = np.ascontiguousarray(env, np.float32)
        assert e.ndim == 3 and e.shape[2] == 3, "env is (h, w, 3) float32"
        pt._ck(pt._L.ptc_set_env_latlong_rgb32f(h, e.ctypes.data_as(C.POINTER(C.c_float)), e.shape[1], e.shape[0]))
    pt._ck(pt._L.ptc_scene_commit(h))
    return int(n), lo, hi


def _decode(fn_name: str, data: bytes) -> np.ndarray:
    L = _load()
    fn = getattr(L, fn_name)
    fn.restype = C.c_int
    fn.argtypes = [C.c_char_p, C.c_ulonglong, C.c_void_p, C.c_ulonglong, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]
    w, h = C.c_int(0), C.c_int(0)
    err = C.create_string_buffer(256)
    if fn(data, len(data), None, 0, C.byref(w), C.byref(h), err, 256):
        raise _ptc.PtcError(err.value.decode())
    out = np.empty((h.value, w.value, 4), np.uint8)
    if fn(data, len(data), out.ctypes.data, out.nbytes, C.byref(w), C.byref(h), err, 256):
        raise _ptc.PtcError(err.value.decode())
    return out


def png_decode(data: bytes) -> np.ndarray:
    """PNG file image → (h, w, 4) uint8 through the loader's own decoder (ptc_png_decode_rgba8)."""
    return _decode("ptc_png_decode_rgba8", data)


def jpeg_decode(data: bytes) -> np.ndarray:
    """JPEG file image → (h, w, 4) uint8 through the loader's own decoder (ptc_jpeg_decode_rgba8)."""
    return _decode("ptc_jpeg_decode_rgba8", data)


def image_decode(data: bytes, kind: int = 0) -> np.ndarray:
    """BMP / TGA / binary PGM-PPM file image → (h, w, 4) uint8 through the loader's decoders (ptc_image_decode_rgba8; kind 0 = by content, 1 BMP, 2 TGA, 3 PNM)."""
    L = _load()
    fn = L.ptc_image_decode_rgba8
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_char_p, C.c_ulonglong, C.c_void_p, C.c_ulonglong, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]
    w, h = C.c_int(0), C.c_int(0)
    err = C.create_string_buffer(256)
    if fn(kind, data, len(data), None, 0, C.byref(w), C.byref(h), err, 256):
        raise ValueError(err.value.decode())
    out = np.empty((h.value, w.value, 4), np.uint8)
    if fn(kind, data, len(data), out.ctypes.data, out.nbytes, C.byref(w), C.byref(h), err, 256):
        raise ValueError(err.value.decode())
    return out


def hdr_decode(data: bytes) -> np.ndarray:
    """Radiance RGBE file image → (h, w, 3) float32 through the library's decoder (ptc_hdr_decode_rgb32f)."""
    L = _load()
    L.ptc_hdr_decode_rgb32f.argtypes = [C.c_char_p, C.c_ulonglong, C.POINTER(C.c_float), C.c_ulonglong, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]
    w, h = C.c_int(0), C.c_int(0)
    err = C.create_string_buffer(256)
    if L.ptc_hdr_decode_rgb32f(data, len(data), None, 0, C.byref(w), C.byref(h), err, 256) < 0:
        raise ValueError(err.value.decode())
    out = np.zeros((h.value, w.value, 3), np.float32)
    if L.ptc_hdr_decode_rgb32f(data, len(data), out.ctypes.data_as(C.POINTER(C.c_float)), out.size, C.byref(w), C.byref(h), err, 256) < 0:
        raise ValueError(err.value.decode())
    return out


def hdr_encode(rgb: np.ndarray, rle: bool = True, magic: str = "#?RADIANCE") -> bytes:
    """Small Radiance RGBE writer for tests and assets: (h, w, 3) float → shared-exponent pixels (mantissas by truncation, as Ward's float2rgbe), "-Y h +X w",
    new-style run-length scanlines when `rle` and 8 <= w < 32768 (runs of >= 3 equal bytes, literal blocks otherwise), flat pixels else."""
    a = np.asarray(rgb, np.float64)
    h, w = a.shape[:2]
    v = a.max(2)
    m, e = np.frexp(np.where(v > 1e-32, v, 1.0))
    scale = np.where(v > 1e-32, m * 256.0 / np.where(v > 1e-32, v, 1.0), 0.0)
    px = np.zeros((h, w, 4), np.uint8)
    px[..., :3] = np.clip((a * scale[..., None]).astype(np.int64), 0, 255)
    px[..., 3] = np.where(v > 1e-32, e + 128, 0)
    out = bytearray(f"{magic}\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n-Y {h} +X {w}\n".encode())
    if not rle or w < 8 or w >= 32768:
        return bytes(out) + px.tobytes()
    for y in range(h):
        out += bytes((2, 2, w >> 8, w & 255))
        for k in range(4):
            row = px[y, :, k]
            i = 0
            while i < w:
                run = 1
                while i + run < w and run < 127 and row[i + run] == row[i]:
                    run += 1
                if run >= 3:
                    out += bytes((128 + run, int(row[i])))
                    i += run
                    continue
                j = i                                                  # a literal block: up to the next run of >= 3, at most 128 bytes
                while j < w and j - i < 128:
                    if j + 2 < w and row[j] == row[j + 1] == row[j + 2]:
                        break
                    j += 1
                if j == i:
                    j = i + 1
                out += bytes((j - i,)) + row[i:j].tobytes()
                i = j
    return bytes(out)


# ----------------------------------------------------------------------------------------------------
_ADAM7 = ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2))


def png_encode(samples: np.ndarray, color_type: int = 6, depth: int = 8, interlace: bool = False, filters="cycle",
               level: int = 9, strategy: int = 0, palette=None, trns: bytes = None, idat_split: int = 0) -> bytes:
    """Small PNG writer for assets and decoder tests.  samples: (h, w, channels) unsigned integers already in the
    file's sample range (channels = 1, 3, 1, 2, 4 for colour types 0, 2, 3, 4, 6).  filters: "cycle" (row y uses
    filter y % 5), or a fixed filter type 0..4.  strategy: zlib strategy (zlib.Z_FIXED forces fixed Huffman blocks)."""
    s = np.asarray(samples)
    h, w, ch = s.shape
    assert ch == {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color_type]
    bpp = max(1, ch * depth // 8)

    def pack_rows(img):
        hh, ww, _ = img.shape
        flat = img.reshape(hh, ww * ch)
        if depth == 16:
            return flat.astype(">u2").view(np.uint8).reshape(hh, -1)
        if depth == 8:
            return flat.astype(np.uint8)
        bits = np.zeros((hh, (ww * ch * depth + 7) // 8 * 8), np.uint8)
        for b in range(depth):
            bits[:, b : ww * ch * depth : depth] = (flat >> (depth - 1 - b)) & 1
        return np.packbits(bits, axis=1)

    def paeth(a, b, c):
        p = a + b - c
        pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
        return a if pa <= pb and pa <= pc else (b if pb <= pc else c)

    def filtered(rows):
        out = bytearray()
        prev = np.zeros(rows.shape[1], np.int32)
        for y in range(rows.shape[0]):
            cur = rows[y].astype(np.int32)
            ft = y % 5 if filters == "cycle" else int(filters)
            left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]]) if cur.size > bpp else np.zeros(cur.size, np.int32)
            ul = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]]) if cur.size > bpp else np.zeros(cur.size, np.int32)
            if ft == 0:
                f = cur
            elif ft == 1:
                f = cur - left
            elif ft == 2:
                f = cur - prev
            elif ft == 3:
                f = cur - ((left + prev) >> 1)
            else:
                f = cur - np.array([paeth(int(a), int(b), int(c)) for a, b, c in zip(left, prev, ul)], np.int32)
            out.append(ft)
            out.extend((f & 255).astype(np.uint8).tobytes())
            prev = cur
        return bytes(out)

    if interlace:
        raw = b"".join(filtered(pack_rows(s[y0::dy, x0::dx])) for x0, y0, dx, dy in _ADAM7 if s[y0::dy, x0::dx].size)
    else:
        raw = filtered(pack_rows(s))
    co = zlib.compressobj(level, zlib.DEFLATED, 15, 8, strategy)
    z = co.compress(raw) + co.flush()

    def chunk(t: bytes, body: bytes) -> bytes:
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xFFFFFFFF)

    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, 1 if interlace else 0))
    if palette is not None:
        out += chunk(b"PLTE", bytes(np.asarray(palette, np.uint8).reshape(-1)))
    if trns is not None:
        out += chunk(b"tRNS", bytes(trns))
    if idat_split:
        for k in range(0, len(z), idat_split):
            out += chunk(b"IDAT", z[k : k + idat_split])
    else:
        out += chunk(b"IDAT", z)
    return out + chunk(b"IEND", b"")


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

    images = []
    for t in getattr(desc, "textures", []) or []:   # RGBA8 images embedded as PNG bufferViews, one glTF texture per image
        images.append({"bufferView": add_view(png_encode(np.ascontiguousarray(t, np.uint8), 6, 8)), "mimeType": "image/png"})
    mats = []
    for m in desc.materials:
        e = [float(x) for x in m.emissive]
        strength = max(1.0, max(e))
        g = {"pbrMetallicRoughness": {"baseColorFactor": [float(x) for x in m.base_color], "metallicFactor": float(m.metallic), "roughnessFactor": float(m.roughness)},
             "emissiveFactor": [x / strength for x in e]}
        if getattr(m, "tex_color", -1) >= 0:
            g["pbrMetallicRoughness"]["baseColorTexture"] = {"index": int(m.tex_color)}
        if getattr(m, "tex_mr", -1) >= 0:
            g["pbrMetallicRoughness"]["metallicRoughnessTexture"] = {"index": int(m.tex_mr)}
        if getattr(m, "tex_normal", -1) >= 0:
            g["normalTexture"] = {"index": int(m.tex_normal)}
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
    if images:
        doc["images"] = images
        doc["samplers"] = [{"magFilter": 9728, "minFilter": 9728, "wrapS": 10497, "wrapT": 10497}]   # NEAREST, REPEAT (what the reference uses regardless)
        doc["textures"] = [{"source": k, "sampler": 0} for k in range(len(images))]
    if any("extensions" in g for g in mats):
        doc["extensionsUsed"] = ["KHR_materials_emissive_strength"]
    js = _pad4(json.dumps(doc, separators=(",", ":")).encode(), b" ")
    bn = _pad4(bytes(blob))
    with open(path, "wb") as f:
        f.write(struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(js) + 8 + len(bn)))
        f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
        f.write(struct.pack("<II", len(bn), 0x004E4942) + bn)
