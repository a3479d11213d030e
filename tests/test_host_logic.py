"""Host logic of the product without a GPU: a PTC_DEVICE_NONE context flattens the scene and builds
the LBVH on the host (no compute call is made), and both must equal the oracle's bit for bit."""
import math

import numpy as np
import pytest


EMPTY = -(2 ** 31)


def _canon(nodes):
    """Layout-independent form of the 4-wide node array (32 floats per node: lo.x[4] lo.y[4] lo.z[4] hi.x[4]
    hi.y[4] hi.z[4] code[4] pad[4]): child boxes + leaf codes (interior links → 0, empty slots → EMPTY), rows sorted."""
    b = nodes[:, :24].view(np.uint32)
    c = nodes[:, 24:28].view(np.int32)
    rows = np.concatenate([b, np.where(c < 0, c, 0).view(np.uint32)], axis=1)
    return rows[np.lexsort(rows.T[::-1])]


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
    assert np.array_equal(t1.view(np.uint32), t2.view(np.uint32))
    assert n1.shape == n2.shape and np.array_equal(_canon(n1), _canon(n2))
    s1, s2 = pt.stats(), o.stats()
    for k in ("n_triangles", "n_bvh_nodes", "n_emitters", "bvh_max_depth"):
        assert s1[k] == s2[k]
    assert s1["n_triangles"] == d.n_triangles


def test_bvh_is_a_valid_partition(pbr):
    d = pbr.scenes.sphere_scene()
    pt = pbr.PathTracer(pbr.DEVICE_NONE).load_scene(d)
    nodes, tris = pt.bvh()
    n_tris = tris.shape[0]
    seen = np.zeros(n_tris, int)
    code = nodes[:, 24:28].view(np.int32)
    interior_refs = np.zeros(nodes.shape[0], int)
    P0, E1, E2 = tris[:, 0:3], tris[:, 4:7], tris[:, 8:11]
    lo_t = np.minimum(np.minimum(P0, P0 + E1), P0 + E2)
    hi_t = np.maximum(np.maximum(P0, P0 + E1), P0 + E2)
    n_children = 0
    for i in range(nodes.shape[0]):
        for c in range(4):
            k = int(code[i, c])
            if k == EMPTY:
                assert c >= 2                                   # slots fill from the front, at least two children
                continue
            n_children += 1
            lo, hi = nodes[i, [c, 4 + c, 8 + c]], nodes[i, [12 + c, 16 + c, 20 + c]]
            if k < 0:
                v = ~k & 0xFFFFFFFF
                first, cnt = v & 0x0FFFFFFF, (v >> 28) + 1
                assert 1 <= cnt <= 4
                seen[first:first + cnt] += 1
                assert (lo_t[first:first + cnt] >= lo - 1e-5).all() and (hi_t[first:first + cnt] <= hi + 1e-5).all()
            else:
                interior_refs[k] += 1
                ch = nodes[k]
                used = ch[24:28].view(np.int32) != EMPTY
                clo = np.array([ch[0:4][used].min(), ch[4:8][used].min(), ch[8:12][used].min()])
                chi = np.array([ch[12:16][used].max(), ch[16:20][used].max(), ch[20:24][used].max()])
                assert (clo >= lo - 1e-5).all() and (chi <= hi + 1e-5).all()
    assert (seen == 1).all()                                    # every triangle in exactly one leaf
    assert interior_refs[0] == 0 and (interior_refs[1:] == 1).all()   # a tree rooted at node 0
    assert n_children / nodes.shape[0] > 3.0                    # the greedy collapse fills the nodes
    prim = tris[:, 3].view(np.uint32)
    assert np.array_equal(np.sort(prim), np.arange(n_tris))      # Morton order is a permutation


def test_single_triangle_and_tiny_scenes(ora, pbr):
    sc = pbr.scene
    v, i = pbr.scenes._quad((-1, -1, -3), (1, -1, -3), (1, 1, -3), (-1, 1, -3))
    for ntri in (1, 2):
        d = sc.SceneDesc([sc.Material()], [sc.MeshDesc(v, i[: 3 * ntri], 0)], [sc.InstanceDesc(0, (0, 0, 0), (1, 0, 0, 0), (1, 1, 1))],
                         sc.CameraDesc((0, 0, 0), (0, 0, -1), math.pi / 2, 1.0))
        pt = pbr.PathTracer(pbr.DEVICE_NONE).load_scene(d)
        o = ora.Oracle().load_scene(d)
        assert np.array_equal(_canon(pt.bvh()[0]), _canon(o.bvh()[0]))
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
