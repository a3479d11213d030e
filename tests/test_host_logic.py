"""Host logic of the product without a GPU: a PTC_DEVICE_NONE context flattens the scene and builds
the BVH on the host (no compute call is made), and both must equal the oracle's bit for bit."""
import math

import numpy as np
import pytest


EMPTY = -(2 ** 31)


def _decode_product(nodes, tris):
    """Product node = 20 words (80 B, layout in ptc_scene.cpp) → per node: org bits, exponents, and per slot
    (type, qlo[3], qhi[3], interior index or leaf triangle-record range)."""
    w = nodes.view(np.uint32)
    out = []
    for i in range(w.shape[0]):
        org = tuple(int(x) for x in w[i, 0:3])
        ew = int(w[i, 3])
        e = (ew & 255, (ew >> 8) & 255, (ew >> 16) & 255)
        imask = ew >> 24
        next_child, next_tri = int(w[i, 16]), int(w[i, 17])
        lmask, two = int(w[i, 18]) & 255, (int(w[i, 18]) >> 8) & 255
        assert imask & lmask == 0 and two & ~lmask == 0 and int(w[i, 18]) >> 16 == 0 and int(w[i, 19]) == 0
        kids = []
        for c in range(8):
            q = [(int(w[i, 4 + 2 * k + c // 4]) >> (8 * (c % 4))) & 255 for k in range(6)]          # qlo.xyz, qhi.xyz
            if (imask >> c) & 1:
                kids.append((2, q, next_child, ()))
                next_child += 1
            elif (lmask >> c) & 1:
                cnt = 1 + ((two >> c) & 1)
                prims = tuple(int(x) for x in tris[next_tri:next_tri + cnt, 3].view(np.uint32))
                kids.append((1, q, next_tri, prims))
                next_tri += cnt
            else:
                kids.append((0, q, -1, ()))
        out.append((org, e, kids))
    return out


def _decode_oracle(nodes, tris):
    w = nodes.view(np.uint32)
    out = []
    for i in range(w.shape[0]):
        org = tuple(int(x) for x in w[i, 0:3])
        e = tuple(int(x) for x in w[i, 3:6])
        qlo, qhi, code = w[i, 6:30].reshape(3, 8), w[i, 30:54].reshape(3, 8), w[i, 54:62].view(np.int32)
        kids = []
        for c in range(8):
            q = [int(qlo[k, c]) for k in range(3)] + [int(qhi[k, c]) for k in range(3)]
            k = int(code[c])
            if k == EMPTY:
                kids.append((0, q, -1, ()))
            elif k < 0:
                v = ~k & 0xFFFFFFFF
                first, cnt = v & 0x0FFFFFFF, (v >> 28) + 1
                kids.append((1, q, first, tuple(int(x) for x in tris[first:first + cnt, 3].view(np.uint32))))
            else:
                kids.append((2, q, k, ()))
        out.append((org, e, kids))
    return out


def _canon(decoded):
    """Layout-independent rows: boxes, child types and the primitive ids of leaf children (node indices and
    triangle-record positions differ between the two builders by design)."""
    rows = []
    for org, e, kids in decoded:
        r = list(org) + list(e)
        for typ, q, _, prims in kids:
            r += [typ] + q + list(prims) + [-1] * (2 - len(prims))
        rows.append(r)
    a = np.asarray(rows, np.int64)
    return a[np.lexsort(a.T[::-1])]


@pytest.mark.parametrize("name,kw", [("cornell", {}), ("sphere10k", {}), ("two_tris_sphere", {}), ("atrium", {"scale": 0.05}), ("atrium", {})])
def test_flatten_and_lbvh_equal_oracle(ora, pbr, name, kw):
    d = pbr.scenes.by_name(name, **kw)
    pt = pbr.PathTracer(pbr.DEVICE_NONE).load_scene(d)
    o = ora.Oracle().load_scene(d)
    v1, i1, m1 = pt.flat_scene()
    v2, i2, m2 = o.flat_scene()
    assert np.array_equal(v1.view(np.uint32), v2.view(np.uint32)) and np.array_equal(i1, i2) and np.array_equal(m1, m2)
    n1, t1 = pt.bvh()
    n2, t2 = o.bvh()
    assert sorted(map(bytes, t1.view(np.uint8))) == sorted(map(bytes, t2.view(np.uint8))) or t1.shape[0] != t2.shape[0]   # same records, different order
    assert n1.shape[0] == n2.shape[0] and np.array_equal(_canon(_decode_product(n1, t1)), _canon(_decode_oracle(n2, t2)))
    s1, s2 = pt.stats(), o.stats()
    for k in ("n_triangles", "n_bvh_nodes", "n_emitters", "bvh_max_depth"):
        assert s1[k] == s2[k]
    assert s1["n_triangles"] == d.n_triangles


def test_bvh_is_a_valid_partition(pbr):
    d = pbr.scenes.sphere_scene()
    pt = pbr.PathTracer(pbr.DEVICE_NONE).load_scene(d)
    nodes, tris = pt.bvh()
    dec = _decode_product(nodes, tris)
    n_rec = tris.shape[0]
    seen = np.zeros(n_rec, int)
    interior_refs = np.zeros(len(dec), int)
    P0, E1, E2 = tris[:, 0:3].astype(np.float64), tris[:, 4:7].astype(np.float64), tris[:, 8:11].astype(np.float64)
    lo_t = np.minimum(np.minimum(P0, P0 + E1), P0 + E2)
    hi_t = np.maximum(np.maximum(P0, P0 + E1), P0 + E2)
    n_children = 0

    def box(org, e, q):
        o = np.asarray(org, np.uint32).view(np.float32).astype(np.float64)
        s = np.array([2.0 ** (x - 127) for x in e])
        return o + np.asarray(q[:3]) * s, o + np.asarray(q[3:]) * s

    exact = {}

    def subtree_box(i):                                         # exact bounds of everything below node i
        if i not in exact:
            lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
            for typ, q, ref, prims in dec[i][2]:
                if typ == 1:
                    lo, hi = np.minimum(lo, lo_t[ref:ref + len(prims)].min(0)), np.maximum(hi, hi_t[ref:ref + len(prims)].max(0))
                elif typ == 2:
                    clo, chi = subtree_box(ref)
                    lo, hi = np.minimum(lo, clo), np.maximum(hi, chi)
            exact[i] = (lo, hi)
        return exact[i]

    for i, (org, e, kids) in enumerate(dec):
        assert sum(1 for k in kids if k[0] != 0) >= 2           # at least two children
        nxt = None
        for typ, q, ref, prims in kids:
            if typ == 0:
                continue
            n_children += 1
            lo, hi = box(org, e, q)                             # quantised box: must contain the exact one
            if typ == 1:
                cnt = len(prims)
                assert 1 <= cnt <= 2
                seen[ref:ref + cnt] += 1
                assert (lo_t[ref:ref + cnt] >= lo - 1e-4).all() and (hi_t[ref:ref + cnt] <= hi + 1e-4).all()   # e1/e2 are rounded differences
            else:
                interior_refs[ref] += 1
                assert nxt is None or ref == nxt                # interior children are consecutive nodes
                nxt = ref + 1
                clo, chi = subtree_box(ref)
                assert (clo >= lo - 1e-4).all() and (chi <= hi + 1e-4).all()
                ext = np.maximum(chi - clo, 1e-6)
    assert (seen == 1).all()                                    # every triangle record in exactly one leaf
    assert interior_refs[0] == 0 and (interior_refs[1:] == 1).all()   # a tree rooted at node 0
    assert n_children / len(dec) > 4.0                          # the greedy collapse fills the nodes (only all-leaf nodes stay short)
    prim = tris[:, 3].view(np.uint32)
    assert np.array_equal(np.sort(prim), np.arange(n_rec))       # a permutation of the primitives


def test_single_triangle_and_tiny_scenes(ora, pbr):
    sc = pbr.scene
    v, i = pbr.scenes._quad((-1, -1, -3), (1, -1, -3), (1, 1, -3), (-1, 1, -3))
    for ntri in (1, 2):
        d = sc.SceneDesc([sc.Material()], [sc.MeshDesc(v, i[: 3 * ntri], 0)], [sc.InstanceDesc(0, (0, 0, 0), (1, 0, 0, 0), (1, 1, 1))],
                         sc.CameraDesc((0, 0, 0), (0, 0, -1), math.pi / 2, 1.0))
        pt = pbr.PathTracer(pbr.DEVICE_NONE).load_scene(d)
        o = ora.Oracle().load_scene(d)
        assert np.array_equal(_canon(_decode_product(*pt.bvh())), _canon(_decode_oracle(*o.bvh())))
        assert pt.stats()["n_bvh_nodes"] == 1 and pt.stats()["n_emitters"] == 0


def test_mesh_builder_and_scene_flatten(pbr):
    sc = pbr.scene
    qv, qi = pbr.scenes._quad((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0))
    sv, si = pbr.scenes.uv_sphere(8, 4, 1.0)
    built = sc.MeshBuilder().addPrimitive(qv, qi, 0).addPrimitive(sv, si, 1).build()
    # MeshBuilder.cpp:35-49: spans carry firstVertex/firstIndex, indices stay primitive-local
    assert [(p.firstVertex, p.vertexCount, p.firstIndex, p.indexCount) for p in built.primitives] == [(0, 4, 0, 6), (4, sv.size, 6, si.size)]
    assert built.vertices.dtype.itemsize == 48 and built.indices.max() < max(4, sv.size)
    scene = sc.Scene()
    parent = scene.addNode(sc.Node("parent", sc.Transform((1, 2, 3), (math.cos(0.25), 0, math.sin(0.25), 0), (2, 2, 2))))
    parent.addChild(sc.Node("child", sc.Transform((0.5, 0, 0)), built))
    assert [n.name for n in scene.iterateAllNodes()] == ["child", "parent"]          # post-order, Scene.cpp:77-82
    cam = sc.CameraDesc((0, 0, 5), (0, 0, 0), 1.0, 1.0)
    d = sc.flatten_scene(scene, [sc.Material(), sc.Material()], cam)
    assert len(d.meshes) == 2 and len(d.instances) == 2
    t = np.asarray(d.instances[0].t)
    # parent composed: t = p_t + R_p·(s_p·c_t); rotation 0.5 rad about +y maps +x to (cos, 0, -sin)
    assert np.allclose(t, np.array([1, 2, 3]) + 2 * 0.5 * np.array([math.cos(0.5), 0, -math.sin(0.5)]), atol=1e-12)
    assert np.allclose(d.instances[0].s, (2, 2, 2))
    d2 = sc.flatten_scene(scene, [sc.Material(), sc.Material()], cam, compose_parents=False)   # the reference quirk
    assert np.allclose(d2.instances[0].t, (0.5, 0, 0))
    with pytest.raises(ValueError):
        sc.MeshBuilder().addPrimitive(qv, np.array([0, 1, 9], np.uint32))


def test_atrium_matches_baseline_config(pbr):
    d = pbr.scenes.atrium()
    assert abs(d.n_triangles - 250_000) <= 2_500                # 250 000 ± 1 %
    assert max(m.vertices.size for m in d.meshes) < 65_536      # every primitive u16-expressible (reference indices)
    assert len(d.materials) == 6
