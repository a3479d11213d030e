"""glTF 2.0 / GLB loader without a GPU: a PTC_DEVICE_NONE context is filled from files that the
test writes itself (the reference's asset assets/models/test_scene.glb is stripped: .MISSING_LARGE_BLOBS), and the
flattened scene must equal the one obtained by handing the same SceneDesc straight to the C-ABI."""
import base64
import json
import math
import os
import struct

import numpy as np
import pytest


def _flat(pbr, desc):
    return pbr.PathTracer(pbr.DEVICE_NONE).load_scene(desc).flat_scene()


def _flat_glb(pbr, path, **kw):
    pt = pbr.PathTracer(pbr.DEVICE_NONE)
    info = pbr.gltf.load_into(pt, path, **kw)
    return pt.flat_scene(), info, pt


@pytest.mark.parametrize("name,kw", [("cornell", {}), ("sphere10k", {}), ("two_tris_sphere", {}), ("atrium", {"scale": 0.05})])
@pytest.mark.parametrize("index_type,interleaved", [("auto", False), ("u32", True)])
def test_glb_round_trip_is_bit_exact(pbr, tmp_path, name, kw, index_type, interleaved):
    d = pbr.scenes.by_name(name, **kw)
    p = str(tmp_path / "scene.glb")
    pbr.gltf.write_glb(d, p, index_type=index_type, interleaved=interleaved)
    (v1, i1, m1), (n, lo, hi), pt = _flat_glb(pbr, p, camera=d.camera)
    v2, i2, m2 = _flat(pbr, d)
    assert n == d.n_triangles
    assert np.array_equal(v1.view(np.uint32), v2.view(np.uint32)) and np.array_equal(i1, i2) and np.array_equal(m1, m2)
    assert np.allclose(lo, v2[:, :3].min(0), atol=1e-5) and np.allclose(hi, v2[:, :3].max(0), atol=1e-5)
    assert pt.stats()["n_emitters"] == pbr.PathTracer(pbr.DEVICE_NONE).load_scene(d).stats()["n_emitters"]


def _tri_mesh(pbr):
    v = np.zeros(3, pbr.scene.MESH_VERTEX)
    v["position"] = [(0, 0, 0), (1, 0, 0), (0, 1, 0)]
    v["normal"] = (0, 0, 1)
    v["tangent"] = (1, 0, 0, 1)
    v["texCoords"] = [(0, 0), (1, 0), (0, 1)]
    return pbr.scene.MeshDesc(v, np.array([0, 1, 2], np.uint32), 0)


def test_node_hierarchy_composition(pbr, tmp_path):
    sc = pbr.scene
    d = sc.SceneDesc([sc.Material()], [_tri_mesh(pbr)], [sc.InstanceDesc(0)], sc.CameraDesc((0, 0, 5), (0, 0, 0), 1.0, 1.0))
    c, s_ = math.cos(math.pi / 4), math.sin(math.pi / 4)   # 90° about +y, glTF rotation order (x,y,z,w)
    M = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, -2, 1]  # column-major translate(0,0,-2)
    nodes = [
        {"name": "root", "translation": [1, 2, 3], "scale": [2, 2, 2], "children": [1, 2]},
        {"name": "rotated", "rotation": [0, s_, 0, c], "mesh": 0, "children": [3]},
        {"name": "by_matrix", "matrix": M, "mesh": 0},
        {"name": "leaf", "translation": [0.5, 0, 0], "mesh": 0},
    ]
    p = str(tmp_path / "h.glb")
    pbr.gltf.write_glb(d, p, nodes=(nodes, [0]))
    (v, idx, _), (n, _, _), _ = _flat_glb(pbr, p, camera=d.camera)
    assert n == 3 and v.shape[0] == 9
    P = v[:, :3].reshape(3, 3, 3)                                     # emission order: children before the node (post-order): leaf, rotated, by_matrix
    local = np.array([(0, 0, 0), (1, 0, 0), (0, 1, 0)], np.float64)
    Ry = np.array([[0, 0, 1], [0, 1, 0], [-1, 0, 0]], np.float64)    # +x → -z
    root = lambda q: 2.0 * q + np.array([1, 2, 3])
    assert np.allclose(P[0], root((local + [0.5, 0, 0]) @ Ry.T), atol=1e-5)   # leaf: root · rotated · translate(0.5,0,0)
    assert np.allclose(P[1], root(local @ Ry.T), atol=1e-5)
    assert np.allclose(P[2], root(local + [0, 0, -2]), atol=1e-5)
    N = v[:, 3:6].reshape(3, 3, 3)
    assert np.allclose(N[1], [[1, 0, 0]] * 3, atol=1e-6)             # +z normal rotated to +x
    # the reference quirk: each node with its local transform only (PbrRenderSystem.cpp:444-446)
    (v2, _, _), _, _ = _flat_glb(pbr, p, camera=d.camera, compose_parents=False)
    P2 = v2[:, :3].reshape(3, 3, 3)
    assert np.allclose(P2[0], local + [0.5, 0, 0], atol=1e-6) and np.allclose(P2[1], local @ Ry.T, atol=1e-6)


def _write_gltf(tmp_path, doc, bins=()):
    for name, data in bins:
        (tmp_path / name).write_bytes(data)
    p = tmp_path / "a.gltf"
    p.write_text(json.dumps(doc))
    return str(p)


def test_defaults_external_buffers_and_normalized_accessors(pbr, tmp_path):
    pos = np.array([(0, 0, 0), (2, 0, 0), (0, 2, 0), (2, 0, 0), (2, 2, 0), (0, 2, 0)], "<f4")   # unindexed: 2 triangles
    uv8 = np.array([(0, 0), (255, 0), (0, 255), (255, 0), (255, 255), (0, 255)], np.uint8)         # normalized UNSIGNED_BYTE
    uv16 = (uv8.astype(np.uint16) * 257).astype("<u2")
    pad = b"\x00" * ((4 - uv8.nbytes % 4) % 4)
    bin0 = pos.tobytes()
    bin1 = uv8.tobytes() + pad + uv16.tobytes()
    doc = {
        "asset": {"version": "2.0"},
        "buffers": [{"uri": "geo.bin", "byteLength": len(bin0)}, {"uri": "data:application/octet-stream;base64," + base64.b64encode(bin1).decode(), "byteLength": len(bin1)}],
        "bufferViews": [{"buffer": 0, "byteLength": len(bin0)}, {"buffer": 1, "byteLength": uv8.nbytes, "byteStride": 2},
                        {"buffer": 1, "byteOffset": uv8.nbytes + len(pad), "byteLength": uv16.nbytes}],
        "accessors": [{"bufferView": 0, "componentType": 5126, "count": 6, "type": "VEC3"},
                      {"bufferView": 1, "componentType": 5121, "normalized": True, "count": 6, "type": "VEC2"},
                      {"bufferView": 2, "componentType": 5123, "normalized": True, "count": 6, "type": "VEC2"}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "TEXCOORD_0": 1}}]},          # no normals/tangents/indices/material/mode
                   {"primitives": [{"attributes": {"POSITION": 0, "TEXCOORD_0": 2}, "material": 0}]}],
        "materials": [{"pbrMetallicRoughness": {"baseColorFactor": [0.5, 0.25, 1, 1], "roughnessFactor": 0.4}, "emissiveFactor": [1, 0.5, 0.25],
                       "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 8}}}],
        "nodes": [{"mesh": 0}, {"mesh": 1, "translation": [0, 0, -1]}],
        "scenes": [{"nodes": [0, 1]}],
    }
    p = _write_gltf(tmp_path, doc, [("geo.bin", bin0)])
    (v, idx, mats), (n, lo, hi), pt = _flat_glb(pbr, p)
    assert n == 4 and np.array_equal(idx.reshape(-1), np.arange(12))              # GenerateMeshIndices
    assert np.allclose(v[:6, 3:6], [[0, 0, 1]] * 6)                               # generated normals: +z for CCW in the xy-plane
    assert np.allclose(np.einsum("ij,ij->i", v[:6, 3:6], v[:6, 6:9]), 0, atol=1e-6) and np.allclose(v[:6, 9], 1)   # generated tangents ⟂ n, w = +1
    exp_uv = uv8.astype(np.float32) / np.float32(255)
    assert np.array_equal(v[:6, 10:12], exp_uv) and np.array_equal(v[6:, 10:12], exp_uv)   # 255-normalised == 65535-normalised here
    assert list(mats) == [1, 1, 0, 0]                                              # default material appended after the file's materials
    assert pt.stats()["n_emitters"] == 2                                           # emissive strength 8 × factor
    assert np.allclose(lo, (0, 0, -1)) and np.allclose(hi, (2, 2, 0))


@pytest.mark.parametrize("mutate,msg", [
    (lambda d: d["meshes"][0]["primitives"][0]["attributes"].pop("POSITION"), "POSITION"),
    (lambda d: d["meshes"][0]["primitives"][0].update(mode=1), "mode 1"),
    (lambda d: d["accessors"][0].update(count=7), "exceeds"),
    (lambda d: d["asset"].update(version="1.0"), "2.x"),
    (lambda d: d["scenes"][0].update(nodes=[5]), "node index"),
    (lambda d: d["nodes"][0].update(children=[0]), "too deep"),
    (lambda d: d["accessors"][0].update(sparse={"count": 1}), "indices / values missing"),
])
def test_loader_errors_are_reported(pbr, tmp_path, mutate, msg):
    pos = np.array([(0, 0, 0), (1, 0, 0), (0, 1, 0)], "<f4")
    doc = {"asset": {"version": "2.0"}, "buffers": [{"uri": "data:application/octet-stream;base64," + base64.b64encode(pos.tobytes()).decode(), "byteLength": 36}],
           "bufferViews": [{"buffer": 0, "byteLength": 36}], "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"}],
           "meshes": [{"primitives": [{"attributes": {"POSITION": 0}}]}], "nodes": [{"mesh": 0}], "scenes": [{"nodes": [0]}]}
    mutate(doc)
    p = _write_gltf(tmp_path, doc)
    with pytest.raises(pbr.PtcError, match=msg):
        _flat_glb(pbr, p)


def test_broken_containers(pbr, tmp_path):
    for name, data, msg in [("x.glb", b"glTF" + struct.pack("<II", 1, 12), "version 1"), ("y.glb", b"glTF" + struct.pack("<II", 2, 9999), "length"),
                            ("z.gltf", b"{ not json", "JSON"), ("w.gltf", b'{"asset": {"version": "2.0"}}', "no scenes")]:
        (tmp_path / name).write_bytes(data)
        with pytest.raises(pbr.PtcError, match=msg):
            _flat_glb(pbr, str(tmp_path / name))
    with pytest.raises(pbr.PtcError, match="cannot open"):
        _flat_glb(pbr, str(tmp_path / "missing.glb"))


def test_textured_glb_carries_images_and_texture_indices(pbr, tmp_path):
    """Config-5 material set through a file: RGBA8 textures embedded as PNG bufferViews, baseColorTexture /
    normalTexture (what Asset::loadMaterial reads, Asset.cpp:147-150) + metallicRoughnessTexture.  The context must
    receive the same texels and the same material → texture links as when the SceneDesc is handed over directly."""
    d = pbr.scenes.by_name("textured_objects")
    assert d.textures and any(m.tex_color >= 0 for m in d.materials) and any(m.tex_normal >= 0 for m in d.materials)
    p = str(tmp_path / "textured.glb")
    pbr.gltf.write_glb(d, p)
    (v1, i1, m1), (n, _, _), pt = _flat_glb(pbr, p, camera=d.camera)
    ref = pbr.PathTracer(pbr.DEVICE_NONE).load_scene(d)
    v2, i2, m2 = ref.flat_scene()
    assert np.array_equal(v1.view(np.uint32), v2.view(np.uint32)) and np.array_equal(i1, i2) and np.array_equal(m1, m2)
    (mats_a, tex_a), (mats_b, tex_b) = pt.description(), ref.description()
    assert len(mats_a) == len(mats_b)
    for (fa, ta), (fb, tb) in zip(mats_a, mats_b):
        assert np.array_equal(fa, fb)
        for ka, kb in zip(ta, tb):                       # the loader numbers textures in first-use order: compare the texels
            assert (ka < 0) == (kb < 0)
            if ka >= 0:
                assert np.array_equal(tex_a[ka], tex_b[kb])


def test_image_sources_and_errors(pbr, tmp_path):
    """Images from a bufferView, an external file and a base64 data URI decode alike; JPEG, a second uv set and bad
    indices are reported."""
    sc = pbr.scene
    rng = np.random.default_rng(3)
    tex = rng.integers(0, 256, (8, 8, 4), dtype=np.uint8)
    png = pbr.gltf.png_encode(tex, 6, 8)
    d = sc.SceneDesc([sc.Material(tex_color=0, tex_normal=0)], [_tri_mesh(pbr)], [sc.InstanceDesc(0)], sc.CameraDesc((0, 0, 5), (0, 0, 0), 1.0, 1.0), textures=[tex])
    glb = str(tmp_path / "a.glb")
    pbr.gltf.write_glb(d, glb)
    # re-express the GLB as .gltf + external .bin, then vary the image source
    raw = open(glb, "rb").read()
    jlen = struct.unpack("<I", raw[12:16])[0]
    doc = json.loads(raw[20 : 20 + jlen])
    blob = raw[20 + jlen + 8 :]
    (tmp_path / "a.bin").write_bytes(blob)
    (tmp_path / "tex.png").write_bytes(png)
    doc["buffers"][0]["uri"] = "a.bin"

    def load(mut):
        j = json.loads(json.dumps(doc))
        mut(j)
        q = str(tmp_path / "v.gltf")
        open(q, "w").write(json.dumps(j))
        pt = pbr.PathTracer(pbr.DEVICE_NONE)
        pbr.gltf.load_into(pt, q, camera=d.camera)
        return pt.description()

    for mut in (lambda j: None,
                lambda j: j["images"].__setitem__(0, {"uri": "tex.png"}),
                lambda j: j["images"].__setitem__(0, {"uri": "data:image/png;base64," + base64.b64encode(png).decode()})):
        mats, texs = load(mut)
        assert len(texs) == 1 and np.array_equal(texs[0], tex) and mats[0][1] == (0, 0, -1)   # one image, shared by both slots
    # a JPEG image goes through the loader's own JPEG decoder; the fixture's texels are the reference decoder's (tests/golden/make_jpeg_golden.py)
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_progressive_422")
    (tmp_path / "tex.jpg").write_bytes(open(gold + ".jpg", "rb").read())
    mats, texs = load(lambda j: j["images"].__setitem__(0, {"uri": "tex.jpg"}))
    assert len(texs) == 1 and np.array_equal(texs[0], np.load(gold + ".npy"))
    (tmp_path / "bad.jpg").write_bytes(b"\xff\xd8\xff\xe0" + b"\0" * 32)
    for mut, msg in ((lambda j: j["images"].__setitem__(0, {"uri": "bad.jpg"}), "JPEG:"),
                     (lambda j: j["materials"][0]["normalTexture"].__setitem__("texCoord", 1), "TEXCOORD_0"),
                     (lambda j: j["materials"][0]["normalTexture"].__setitem__("index", 5), "texture index out of range"),
                     (lambda j: j["textures"][0].__setitem__("source", 3), "image source"),
                     (lambda j: j["images"].__setitem__(0, {"uri": "a.bin"}), "none of PNG, BMP, GIF, PSD, PIC")):
        with pytest.raises(pbr.PtcError, match=msg):
            load(mut)


def test_hostile_accessors_and_json_are_rejected(pbr, tmp_path):
    """Found by fuzzing the loader under ASan/UBSan: negative offsets/counts/strides wrapped the bounds check."""
    d = pbr.scenes.by_name("cornell")
    glb = str(tmp_path / "c.glb")
    pbr.gltf.write_glb(d, glb)
    raw = open(glb, "rb").read()
    jlen = struct.unpack("<I", raw[12:16])[0]
    doc = json.loads(raw[20 : 20 + jlen])
    body = raw[20 + jlen :]

    def load(j):
        js = json.dumps(j).encode() if not isinstance(j, bytes) else j
        js += b" " * ((4 - len(js) % 4) % 4)
        out = raw[:8] + struct.pack("<I", 12 + 8 + len(js) + len(body)) + struct.pack("<II", len(js), 0x4E4F534A) + js + body
        q = str(tmp_path / "h.glb")
        open(q, "wb").write(out)
        pbr.gltf.load_into(pbr.PathTracer(pbr.DEVICE_NONE), q, camera=d.camera)

    load(doc)
    for key, where, val in (("byteOffset", "accessors", -1), ("byteOffset", "bufferViews", -8), ("count", "accessors", -3), ("byteStride", "bufferViews", -4),
                            ("byteOffset", "accessors", 2 ** 62), ("count", "accessors", 2 ** 40), ("byteStride", "bufferViews", 2 ** 61)):
        j = json.loads(json.dumps(doc))
        j[where][0][key] = val
        with pytest.raises(pbr.PtcError, match="negative|exceeds"):
            load(j)
    with pytest.raises(pbr.PtcError, match="nesting"):
        load(b'{"asset":{"version":"2.0"},"x":' + b"[" * 5000 + b"]" * 5000 + b"}")


def test_uris_are_percent_decoded_and_kept_inside_the_asset_directory(pbr, tmp_path):
    """glTF URIs are RFC 3986 relative references: "my%20mesh.bin" names the file "my mesh.bin", sub-directories are fine — and an asset is
    untrusted input: absolute paths, schemes, and ".." segments that climb out of the asset's directory (percent-encoded or not) are refused,
    with an error that says so, before any file is opened."""
    pos = np.array([(0, 0, 0), (1, 0, 0), (0, 1, 0)], "<f4")

    def doc(uri):
        return {"asset": {"version": "2.0"}, "buffers": [{"uri": uri, "byteLength": 36}], "bufferViews": [{"buffer": 0, "byteLength": 36}],
                "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"}],
                "meshes": [{"primitives": [{"attributes": {"POSITION": 0}}]}], "nodes": [{"mesh": 0}], "scenes": [{"nodes": [0]}]}

    (tmp_path / "sub dir").mkdir()
    (tmp_path / "sub dir" / "my mesh.bin").write_bytes(pos.tobytes())
    p = _write_gltf(tmp_path, doc("sub%20dir/my%20mesh.bin"), [])
    (v, idx, mats), (n, lo, hi), _ = _flat_glb(pbr, p)
    assert n == 1 and np.allclose(hi, (1, 1, 0))
    p = _write_gltf(tmp_path, doc("sub%20dir/../sub%20dir/./my%20mesh.bin"), [])      # ".." that stays inside
    assert _flat_glb(pbr, p)[1][0] == 1
    outside = tmp_path.parent / "outside.bin"
    outside.write_bytes(pos.tobytes())
    try:
        for bad, msg in (("../outside.bin", "leaves the asset"), ("%2e%2e/outside.bin", "leaves the asset"), ("sub%20dir/../../outside.bin", "leaves the asset"),
                         (str(outside), "only relative"), ("file:///etc/passwd", "only relative"), ("a%zz.bin", "percent escape"), ("a%2", "percent escape")):
            p = _write_gltf(tmp_path, doc(bad), [])
            with pytest.raises(pbr.PtcError, match=msg):
                _flat_glb(pbr, p)
    finally:
        outside.unlink()


def test_sparse_accessors(pbr, tmp_path):
    """glTF 2.0 §3.6.2.3 (fastgltf, the reference's loader, reads them): a sparse accessor is its bufferView's elements — zeros when it has none — with
    `sparse.count` of them replaced.  Positions with two displaced vertices, and a position accessor WITHOUT a bufferView built from zeros; hostile
    index lists are refused."""
    pos = np.array([(0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0)], "<f4")
    idx = np.array([0, 1, 2, 1, 3, 2], "<u2")
    sp_i = np.array([1, 3], "<u2")                                        # vertices 1 and 3 ...
    sp_v = np.array([(2, 0, 0), (2, 2, 0)], "<f4")                        # ... move out to x = 2
    blob = pos.tobytes() + idx.tobytes() + sp_i.tobytes() + sp_v.tobytes()
    o_idx, o_si, o_sv = pos.nbytes, pos.nbytes + idx.nbytes, pos.nbytes + idx.nbytes + sp_i.nbytes

    def doc(sparse, with_view=True):
        acc0 = {"componentType": 5126, "count": 4, "type": "VEC3", "sparse": sparse}
        if with_view:
            acc0["bufferView"] = 0
        return {"asset": {"version": "2.0"}, "buffers": [{"uri": "s.bin", "byteLength": len(blob)}],
                "bufferViews": [{"buffer": 0, "byteLength": pos.nbytes}, {"buffer": 0, "byteOffset": o_idx, "byteLength": idx.nbytes},
                                {"buffer": 0, "byteOffset": o_si, "byteLength": sp_i.nbytes}, {"buffer": 0, "byteOffset": o_sv, "byteLength": sp_v.nbytes}],
                "accessors": [acc0, {"bufferView": 1, "componentType": 5123, "count": 6, "type": "SCALAR"}],
                "meshes": [{"primitives": [{"attributes": {"POSITION": 0}, "indices": 1}]}], "nodes": [{"mesh": 0}], "scenes": [{"nodes": [0]}]}

    good = {"count": 2, "indices": {"bufferView": 2, "componentType": 5123}, "values": {"bufferView": 3}}
    (v, _, _), (n, lo, hi), _ = _flat_glb(pbr, _write_gltf(tmp_path, doc(good), [("s.bin", blob)]))
    assert n == 2 and np.array_equal(v[:4, :3], [[0, 0, 0], [2, 0, 0], [0, 1, 0], [2, 2, 0]]) and np.allclose(hi, (2, 2, 0))
    (v, _, _), _, _ = _flat_glb(pbr, _write_gltf(tmp_path, doc(good, with_view=False), [("s.bin", blob)]))
    assert np.array_equal(v[:4, :3], [[0, 0, 0], [2, 0, 0], [0, 0, 0], [2, 2, 0]])             # zeros + the two replacements
    for bad, msg in (({"count": 2, "indices": {"bufferView": 2, "componentType": 5126}, "values": {"bufferView": 3}}, "UNSIGNED"),
                     ({"count": 5, "indices": {"bufferView": 2, "componentType": 5123}, "values": {"bufferView": 3}}, "more replacements"),
                     ({"count": 2, "indices": {"bufferView": 2, "componentType": 5123, "byteOffset": 2}, "values": {"bufferView": 3}}, "exceeds"),
                     ({"count": 2, "indices": {"bufferView": 1, "componentType": 5123}, "values": {"bufferView": 3}}, "strictly increasing"),   # 0, 1 is fine; use idx view: 0,1 ok -> make it fail below
                     ({"count": 2, "indices": {"bufferView": 2, "componentType": 5123}, "values": {"bufferView": 9}}, "out of range")):
        if msg == "strictly increasing":
            bad = {"count": 3, "indices": {"bufferView": 1, "componentType": 5123, "byteOffset": 2}, "values": {"bufferView": 0}}      # indices 1, 2, 1
        with pytest.raises(pbr.PtcError, match=msg):
            _flat_glb(pbr, _write_gltf(tmp_path, doc(bad), [("s.bin", blob)]))
