"""Host logic of the product without a GPU: a PTC_DEVICE_NONE context flattens the scene and builds
the BVH on the host (no compute call is made), and both must equal the oracle's bit for bit."""
import math

import numpy as np
import pytest


EMPTY = -(2 ** 31)


def _decode_product(units, n_nodes, n_tris, grid):
    """Product BVH = one array of 16-byte units (layout in ptc_scene.cpp): walk it from the root at unit 0 → per node: the origin's
    16-bit grid coordinates, exponents, and per slot (type, qlo[3] + qhi[3], unit address of the interior child or of the leaf's first
    triangle record, primitive ids of the leaf).  Also checks the structural rules of the layout."""
    w = units.view(np.uint32)
    out, todo, seen_units = [], [0], np.zeros(len(units), np.int32)
    while todo:
        u = todo.pop()
        assert u % 4 == 0                                            # nodes sit on 64-byte boundaries
        seen_units[u:u + 4] += 1
        n = w[u:u + 4].reshape(16)
        oq = (int(n[0]) & 0xFFFF, int(n[0]) >> 16, int(n[1]) & 0xFFFF)
        e = ((int(n[1]) >> 16) & 255, int(n[1]) >> 24, int(n[2]) & 255)
        imask, lmask, two = (int(n[2]) >> 8) & 255, (int(n[2]) >> 16) & 255, int(n[2]) >> 24
        assert imask & lmask == 0 and two & ~lmask == 0
        block = int(n[3])
        assert block % 4 == 0 and block > u                         # children blocks are 64-byte aligned and come after their node
        next_child, next_tri = block, block + 4 * bin(imask).count("1")
        kids = []
        for c in range(8):
            q = [(int(n[4 + 2 * k + c // 4]) >> (8 * (c % 4))) & 255 for k in range(6)]          # qlo.xyz, qhi.xyz
            if (imask >> c) & 1:
                kids.append((2, q, next_child, ()))
                todo.append(next_child)
                next_child += 4
            elif (lmask >> c) & 1:
                cnt = 1 + ((two >> c) & 1)
                prims = tuple(int(w[next_tri + 3 * j, 3]) for j in range(cnt))
                seen_units[next_tri:next_tri + 3 * cnt] += 1
                kids.append((1, q, next_tri, prims))
                next_tri += 3 * cnt
            else:
                kids.append((0, q, -1, ()))
        out.append((oq, e, kids, u))
    assert len(out) == n_nodes and seen_units.max() == 1          # every unit belongs to at most one record
    assert int(seen_units.sum()) == 4 * n_nodes + 3 * n_tris       # the rest is alignment padding
    return out


def _decode_oracle(nodes, tris):
    w = nodes.view(np.uint32)
    out = []
    for i in range(w.shape[0]):
        oq = tuple(int(x) for x in w[i, 3:6])
        e = tuple(int(x) for x in w[i, 6:9])
        qlo, qhi, code = w[i, 9:33].reshape(3, 8), w[i, 33:57].reshape(3, 8), w[i, 57:65].view(np.int32)
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
        out.append((oq, e, kids, i))
    return out


def _canon(decoded):
    """Layout-independent rows: boxes, child types and the primitive ids of leaf children (node indices and
    triangle-record positions differ between the two builders by design)."""
    rows = []
    for org, e, kids, _ in decoded:
        r = list(org) + list(e)
        for typ, q, _, prims in kids:
            r += [typ] + q + list(prims) + [-1] * (2 - len(prims))
        rows.append(r)
    a = np.asarray(rows, np.int64)
    return a[np.lexsort(a.T[::-1])]


@pytest.mark.parametrize("builder", ["sah", "lbvh"])
@pytest.mark.parametrize("name,kw", [("cornell", {}), ("sphere10k", {}), ("two_tris_sphere", {}), ("atrium", {"scale": 0.05}), ("atrium", {}),
                                     ("textured_objects", {}), ("textured_atrium", {"scale": 0.05, "tex_size": 32, "env_size": (32, 16)})])
def test_flatten_and_bvh_equal_oracle(ora, pbr, name, kw, builder):
    d = pbr.scenes.by_name(name, **kw)
    d.bvh_builder = builder
    pt = pbr.PathTracer(pbr.DEVICE_NONE).load_scene(d)
    o = ora.Oracle().load_scene(d)
    v1, i1, m1 = pt.flat_scene()
    v2, i2, m2 = o.flat_scene()
    assert np.array_equal(v1.view(np.uint32), v2.view(np.uint32)) and np.array_equal(i1, i2) and np.array_equal(m1, m2)
    units, nn, nt, grid = pt.bvh()
    n2, t2 = o.bvh()
    dec = _decode_product(units, nn, nt, grid)
    assert nn == n2.shape[0] and np.array_equal(_canon(dec), _canon(_decode_oracle(n2, t2)))
    # the oracle's node origin is the grid point the product stores: org = fma(oq, step, scene_lo), here in exact arithmetic
    # (the float32 fma rounds the exact value once)
    from fractions import Fraction as F
    import struct
    for row in n2[:: max(1, len(n2) // 200)]:
        for k in range(3):
            exact = F(float(grid[k])) + F(int(row.view(np.uint32)[3 + k])) * F(float(grid[3 + k]))
            got = F(float(row[k]))
            ulp = F(float(np.spacing(np.float32(abs(float(row[k])) or 1e-30))))
            assert abs(got - exact) <= ulp / 2, (row[k], float(exact))
    # triangle records: the same (v0,prim | e1,class | e2) rows as the oracle's, in a different order
    recs = []
    for _, _, kids, _ in dec:
        for typ, _, ref, prims in kids:
            if typ == 1:
                recs += [units[ref + 3 * j: ref + 3 * j + 3].reshape(12) for j in range(len(prims))]
    assert sorted(map(bytes, np.asarray(recs, np.float32).view(np.uint8))) == sorted(map(bytes, t2.view(np.uint8)))
    s1, s2 = pt.stats(), o.stats()
    for k in ("n_triangles", "n_bvh_nodes", "n_emitters", "bvh_max_depth"):
        assert s1[k] == s2[k]
    assert s1["n_triangles"] == d.n_triangles


def test_bvh_is_a_valid_partition(pbr):
    d = pbr.scenes.sphere_scene()
    pt = pbr.PathTracer(pbr.DEVICE_NONE).load_scene(d)
    units, nn, nt, grid = pt.bvh()
    dec = _decode_product(units, nn, nt, grid)
    by_addr = {u: i for i, (_, _, _, u) in enumerate(dec)}
    tri = lambda ref: units[ref:ref + 3].reshape(12)
    n_children, prims_seen, interior_refs = 0, [], np.zeros(len(dec), int)

    def tri_box(ref):
        t = tri(ref).astype(np.float64)
        P0, E1, E2 = t[0:3], t[4:7], t[8:11]
        pts = np.stack([P0, P0 + E1, P0 + E2])
        return pts.min(0), pts.max(0)

    def box(oq, e, q):
        o = np.asarray(grid[:3], np.float64) + np.asarray(oq, np.float64) * np.asarray(grid[3:], np.float64)
        sc = np.array([2.0 ** (x - 127) for x in e])
        return o + np.asarray(q[:3]) * sc, o + np.asarray(q[3:]) * sc

    exact = {}

    def subtree_box(i):                                         # exact bounds of everything below node i
        if i not in exact:
            lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
            for typ, q, ref, prims in dec[i][2]:
                if typ == 1:
                    for j in range(len(prims)):
                        tl, th = tri_box(ref + 3 * j)
                        lo, hi = np.minimum(lo, tl), np.maximum(hi, th)
                elif typ == 2:
                    clo, chi = subtree_box(by_addr[ref])
                    lo, hi = np.minimum(lo, clo), np.maximum(hi, chi)
            exact[i] = (lo, hi)
        return exact[i]

    for i, (oq, e, kids, _) in enumerate(dec):
        assert sum(1 for k in kids if k[0] != 0) >= 2           # at least two children
        nxt = None
        for typ, q, ref, prims in kids:
            if typ == 0:
                continue
            n_children += 1
            lo, hi = box(oq, e, q)                              # quantised box: must contain the exact one
            if typ == 1:
                assert 1 <= len(prims) <= 2
                prims_seen += list(prims)
                for j in range(len(prims)):
                    tl, th = tri_box(ref + 3 * j)
                    assert (tl >= lo - 1e-4).all() and (th <= hi + 1e-4).all()   # e1/e2 are rounded differences
            else:
                interior_refs[by_addr[ref]] += 1
                assert nxt is None or ref == nxt                # interior children are consecutive 64-byte records
                nxt = ref + 4
                clo, chi = subtree_box(by_addr[ref])
                assert (clo >= lo - 1e-4).all() and (chi <= hi + 1e-4).all()
    assert interior_refs[by_addr[0]] == 0 and (np.delete(interior_refs, by_addr[0]) == 1).all()   # a tree rooted at unit 0
    assert n_children / len(dec) > 4.0                          # the cost-optimal collapse fills the nodes
    assert sorted(prims_seen) == list(range(d.n_triangles))     # every primitive in exactly one leaf


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


def test_lbvh_is_the_morton_radix_tree(ora, pbr):
    """PTC_BVH_LBVH (BASELINE north_star's "flattened LBVH"): same triangles, a different tree; the SAH tree needs fewer node visits
    for the same hits; the builder is reset by scene_begin and PTC_BVH=lbvh makes it the default of a new context."""
    import os, subprocess, sys
    d = pbr.scenes.by_name("atrium", scale=0.05)
    rng = np.random.default_rng(5)
    n = 4000
    org = np.tile(np.asarray(d.camera.position, np.float32), (n, 1))
    dirs = rng.normal(size=(n, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    res = {}
    for b in ("sah", "lbvh"):
        d.bvh_builder = b
        o = ora.Oracle().load_scene(d)
        t, prim, uv = o.trace_closest(org, dirs)
        res[b] = (t, prim, uv, o.stats())
    assert np.array_equal(res["sah"][0], res["lbvh"][0]) and np.array_equal(res["sah"][1], res["lbvh"][1])   # closest hit does not depend on the tree
    v_sah, v_lbvh = res["sah"][3]["node_visits_closest"], res["lbvh"][3]["node_visits_closest"]
    assert v_sah < v_lbvh, (v_sah, v_lbvh)
    # the product: explicit choice, reset by scene_begin, environment default
    pt = pbr.PathTracer(pbr.DEVICE_NONE)
    d.bvh_builder = "lbvh"
    u_l = pt.load_scene(d).bvh()[0].view(np.uint32).copy()
    d.bvh_builder = None
    u_d = pt.load_scene(d).bvh()[0].view(np.uint32).copy()          # scene_begin went back to the default
    d.bvh_builder = "sah"
    u_s = pt.load_scene(d).bvh()[0].view(np.uint32).copy()
    assert np.array_equal(u_d, u_s) and (u_l.shape != u_s.shape or not np.array_equal(u_l, u_s))
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import pbr_amd; d = pbr_amd.scenes.by_name('atrium', scale=0.05); "
            "pt = pbr_amd.PathTracer(pbr_amd.DEVICE_NONE).load_scene(d); print(pt.bvh()[0].tobytes().hex()[:64], pt.bvh()[0].shape[0])"
            % os.path.dirname(os.path.dirname(pbr.__file__)))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PTC_BVH="lbvh"), capture_output=True, text=True, check=True).stdout.split()
    assert out[0] == u_l.tobytes().hex()[:64] and int(out[1]) == u_l.shape[0]


def test_material_classes_follow_the_texture_sets(ora, pbr):
    """The class word of a triangle record (what k_shade sorts hits by): 0 untextured Lambert, 1 untextured GGX, 2 + set % 5 for textured
    materials, a set being a distinct (colour, normal, metal-rough) triple numbered in material order.  Checked on the triangle records
    of the product's tree dump and of the oracle's, against classes worked out here."""
    d = pbr.scenes.textured_objects()
    # materials: 0 (0,1,2) -> set 0; 1 (0,-1,-1) -> set 1; 2 untextured Lambert emitter (unused mesh dropped); 3 (-1,1,-1) -> set 2
    from pbr_amd.scene import Material
    d.materials.append(Material((0.5, 0.5, 0.5, 1.0), 0.0, 0.4, (0, 0, 0), 0, 1, 2))      # 4: same triple as material 0 -> set 0 again
    d.materials.append(Material((0.5, 0.5, 0.5, 1.0), 0.3, 1.0))                            # 5: untextured GGX (metallic > 0)
    d.materials.append(Material((0.5, 0.5, 0.5, 1.0), 0.0, 1.0))                            # 6: untextured Lambert
    qv, qi = pbr.scenes._quad((-1, 3, -1), (1, 3, -1), (1, 3, 1), (-1, 3, 1))
    from pbr_amd.scene import MeshDesc, InstanceDesc
    for k, mat in enumerate((4, 5, 6)):
        d.meshes.append(MeshDesc(qv, qi, mat))
        d.instances.append(InstanceDesc(len(d.meshes) - 1, (3.0 * k, 0.0, 0.0), (1.0, 0.0, 0.0, 0.0), (1.0, 1.0, 1.0)))
    want = {0: 2, 1: 3, 3: 4, 4: 2, 5: 1, 6: 0}
    pt = pbr.PathTracer(pbr.DEVICE_NONE).load_scene(d)
    o = ora.Oracle().load_scene(d)
    _, _, tri_mat = pt.flat_scene()
    _, t2 = o.bvh()
    prim = t2.view(np.uint32)[:, 3]
    cls = t2.view(np.uint32)[:, 7]
    assert len(prim) == len(tri_mat)
    for p, c in zip(prim, cls):
        assert c == want[int(tri_mat[p])], (p, tri_mat[p], c)


def _rotated(desc, k):
    """The viewer's App::update turns its nodes a little every frame (src/gltf_viewer/App.cpp:306-313): here every third instance gets
    a yaw of 0.05*k rad about its own origin and a small lift, the rest stay."""
    import copy
    import math
    out = []
    for i, it in enumerate(desc.instances):
        if i % 3 or getattr(it, "matrix", None) is not None:
            out.append(None)
            continue
        a = 0.05 * k + 0.01 * i
        q = (math.cos(a / 2), 0.0, math.sin(a / 2), 0.0)
        # compose with the instance's own rotation: q_new = q * q_old
        w1, x1, y1, z1 = q
        w2, x2, y2, z2 = it.q_wxyz
        qn = (w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2)
        out.append(((it.t[0], it.t[1] + 0.02 * k, it.t[2]), qn, it.s))
    return out


@pytest.mark.parametrize("name,kw,builder", [("cornell", {}, None), ("sphere10k", {}, "lbvh"), ("atrium", {"scale": 0.05}, None), ("textured_objects", {}, None)])
def test_host_share_of_the_device_refit(pbr, name, kw, builder):
    """csrc/pt_refit.hip's kernels need a GPU; what the host contributes to them does not: the plan of the committed scene (every 8-wide node's
    address once, children in earlier levels than their parents — the order the node kernel is launched in — and the vertex -> instance map) and
    the emitter table rebuilt from the emissive primitives alone, which must be the host refit's bit for bit after any move; a move that changes
    which triangles are emitters is recognised (the device path then hands over to the host refit)."""
    import copy
    d = pbr.scenes.by_name(name, **kw)
    if builder:
        d = copy.deepcopy(d)
        d.bvh_builder = builder
    pt = pbr.PathTracer(pbr.DEVICE_NONE).load_scene(d)
    st = pt.stats()
    r = pt.refit_host_parts()
    assert r["levels_ok"] == 1 and r["emitters_equal"] == 1 and r["transforms_finite"] == 1
    assert r["n_tris"] == st["n_triangles"] and r["nodes_listed"] == st["n_bvh_nodes"] and r["levels"] == st["bvh_max_depth"] + 1
    assert r["n_verts"] == pt.flat_scene()[0].shape[0]
    for k in (1, 2):
        for i, m in enumerate(_rotated(d, k)):
            if m is not None:
                pt.update_instance(i, m[0], m[1], tuple(np.float32(x) * np.float32(1.0 + 0.07 * k) for x in m[2]))
        pt.scene_refit()
        r = pt.refit_host_parts()
        assert r["levels_ok"] == 1 and r["emitters_equal"] == 1, (k, r)
    emissive = [i for i, it in enumerate(d.instances) if any(x > 0 for x in d.materials[d.meshes[it.mesh].material].emissive)]
    if emissive:
        assert r["emissive_prims"] > 0
        i = emissive[0]
        pt.update_instance(i, (0.0, 0.0, 0.0), (1.0, 0.0, 0.0, 0.0), (0.0, 0.0, 0.0))      # the emitter collapses to a point: it stops being one
        assert pt.refit_host_parts()["emitters_equal"] == 0                                # before the refit the committed emitter set is the old one
        pt.scene_refit()
        assert pt.stats()["n_emitters"] < st["n_emitters"] and pt.refit_host_parts()["emitters_equal"] == 1
    bad = np.eye(4, dtype=np.float32).reshape(16)
    bad[5] = np.inf
    pt.update_instance(0, matrix=bad)
    assert pt.refit_host_parts()["transforms_finite"] == 0


@pytest.mark.parametrize("name,kw", [("cornell", {}), ("sphere10k", {}), ("atrium", {"scale": 0.05}), ("textured_objects", {}), ("textured_atrium", {"scale": 0.03, "tex_size": 32, "env_size": (32, 16)})])
def test_host_share_of_the_device_commit(pbr, name, kw):
    """ptc_scene_commit with the LBVH builder on a device context lets the device flatten and build (csrc/ptc_api.cpp device_commit, tested on the GPU); what the host
    contributes needs none: ptc_build_skeleton — world vertex indices and material per primitive, the material table, the emitter index of every primitive with the emitter table
    and its cdf from the emissive primitives ALONE (each vertex through its instance's matrix, no flatten), textures, texture sets, environment tables — must be, bit for bit, what
    the full host build of the same description holds; also after the instances have moved and the scene was committed again, and with an emitter collapsed to zero area."""
    import copy
    d = copy.deepcopy(pbr.scenes.by_name(name, **kw))
    for builder in ("lbvh", "sah"):
        d.bvh_builder = builder
        pt = pbr.PathTracer(pbr.DEVICE_NONE).load_scene(d)
        r = pt.commit_host_parts()
        assert r["n_tris"] == pt.stats()["n_triangles"] and r["n_lights"] == pt.stats()["n_emitters"], r
        assert r["indices_ok"] == r["emitters_ok"] == r["materials_ok"] == r["tables_ok"] == r["sizes_ok"] == 1, r
    for i, m in enumerate(_rotated(d, 2)):
        if m is not None:
            d.instances[i].t, d.instances[i].q_wxyz, d.instances[i].s = m[0], m[1], tuple(float(np.float32(x) * np.float32(1.3)) for x in m[2])
    emissive = [i for i, it in enumerate(d.instances) if any(x > 0 for x in d.materials[d.meshes[it.mesh].material].emissive)]
    if len(emissive) > 1:
        d.instances[emissive[0]].s = (0.0, 1.0, 1.0)                       # zero area: emissive material, not an emitter
    pt = pbr.PathTracer(pbr.DEVICE_NONE).load_scene(d)
    r = pt.commit_host_parts()
    assert r["indices_ok"] == r["emitters_ok"] == r["materials_ok"] == r["tables_ok"] == r["sizes_ok"] == 1 and r["n_lights"] == pt.stats()["n_emitters"], r


@pytest.mark.parametrize("name,kw", [("sphere10k", {}), ("atrium", {"scale": 0.05}), ("textured_objects", {})])
def test_refit_equals_oracle_and_keeps_the_topology(ora, pbr, name, kw):
    """ptc_update_instance + ptc_scene_refit against the oracle's: the refitted trees are identical bit for bit (same topology and slots as
    the committed tree, boxes and records from the moved vertices); a refit without any change reproduces the committed tree; the flattened
    scene after the refit is the one a fresh commit of the same transforms flattens to."""
    d = pbr.scenes.by_name(name, **kw)
    pt = pbr.PathTracer(pbr.DEVICE_NONE).load_scene(d)
    o = ora.Oracle().load_scene(d)
    units0, nn0, nt0, grid0 = pt.bvh()
    pt.scene_refit()
    units1, nn1, nt1, grid1 = pt.bvh()
    assert nn0 == nn1 and np.array_equal(units0.view(np.uint32), units1.view(np.uint32)) and np.array_equal(np.asarray(grid0), np.asarray(grid1))
    n0, t0 = o.bvh()
    o.scene_refit()
    n1, t1 = o.bvh()
    assert np.array_equal(n0.view(np.uint32), n1.view(np.uint32)) and np.array_equal(t0.view(np.uint32), t1.view(np.uint32))
    for k in (1, 2):
        moved = _rotated(d, k)
        import copy
        d2 = copy.deepcopy(d)
        for i, m in enumerate(moved):
            if m is None:
                continue
            pt.update_instance(i, *m)
            o.update_instance(i, *m)
            d2.instances[i].t, d2.instances[i].q_wxyz, d2.instances[i].s = m
        pt.scene_refit()
        o.scene_refit()
        units, nn, nt, grid = pt.bvh()
        n2, t2 = o.bvh()
        assert nn == nn0 == n2.shape[0]
        assert np.array_equal(_canon(_decode_product(units, nn, nt, grid)), _canon(_decode_oracle(n2, t2)))
        # the topology stayed: child codes of the oracle's nodes are those of the commit
        assert np.array_equal(n2.view(np.uint32)[:, -8:], n0.view(np.uint32)[:, -8:])
        fresh = pbr.PathTracer(pbr.DEVICE_NONE).load_scene(d2)
        v1, i1, m1 = pt.flat_scene()
        v2, i2, m2 = fresh.flat_scene()
        assert np.array_equal(v1.view(np.uint32), v2.view(np.uint32)) and np.array_equal(i1, i2) and np.array_equal(m1, m2)
        v3, _, _ = o.flat_scene()
        assert np.array_equal(v1.view(np.uint32), v3.view(np.uint32))
        assert pt.stats()["seconds_refit"] > 0.0
    with pytest.raises(pbr.PtcError):
        pt.update_instance(10 ** 6, (0, 0, 0), (1, 0, 0, 0), (1, 1, 1))
    with pytest.raises(pbr.PtcError):
        pbr.PathTracer(pbr.DEVICE_NONE).scene_refit()                  # nothing committed


def test_refit_refuses_a_description_that_changed_since_the_commit(pbr):
    """ptc_add_instance* / ptc_add_mesh are accepted after a commit (they describe the next one); a refit of such a description would index the
    committed arrays out of bounds, so ptc_scene_refit refuses it before anything is computed — and a fresh commit takes it."""
    import ctypes as C
    d = pbr.scenes.by_name("cornell")
    pt = pbr.PathTracer(pbr.DEVICE_NONE).load_scene(d)
    f3 = lambda *v: (C.c_float * len(v))(*v)
    assert pt._L.ptc_add_instance(pt._h, 0, f3(0.0, 0.1, 0.0), f3(1.0, 0.0, 0.0, 0.0), f3(1.0, 1.0, 1.0)) >= 0
    with pytest.raises(pbr.PtcError, match="changed since the commit"):
        pt.scene_refit()
    pt.update_instance(len(d.instances), (0.0, 0.2, 0.0), (1.0, 0.0, 0.0, 0.0), (1.0, 1.0, 1.0))     # the new instance exists in the description
    with pytest.raises(pbr.PtcError, match="changed since the commit"):
        pt.scene_refit()
    assert pt._L.ptc_scene_commit(pt._h) == 0
    pt.scene_refit()
    assert pt.stats()["n_triangles"] > pbr.PathTracer(pbr.DEVICE_NONE).load_scene(d).stats()["n_triangles"]


def test_a_group_builds_its_scene_once(pbr):
    """ptc_group_scene_commit: device 0's context flattens and builds, the other contexts of the group take THAT build (one host build for N devices) — checked
    without a GPU on a description-only group (every id PTC_DEVICE_NONE: no RCCL, nothing uploaded) through the identity of the host build each context holds.
    A refit of the group leaves them sharing one (new) build again; a context that refits on its own gets its own copy."""
    d = pbr.scenes.by_name("cornell")
    g = pbr.ptc.Group([pbr.DEVICE_NONE, pbr.DEVICE_NONE, pbr.DEVICE_NONE]).load_scene(d)
    ids = [g.ctx(i).host_build_id() for i in range(3)]
    assert ids[0] != 0 and ids.count(ids[0]) == 3, ids
    assert [g.ctx(i).stats()["n_triangles"] for i in range(3)] == [12, 12, 12]
    u0 = g.ctx(0).bvh()[0].view(np.uint32).copy()
    g.ctx(0).update_instance(0, (0.1, 0.0, 0.0), (1.0, 0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
    g.scene_refit()
    ids2 = [g.ctx(i).host_build_id() for i in range(3)]
    assert ids2.count(ids2[0]) == 3 and ids2[0] != 0
    assert not np.array_equal(g.ctx(2).bvh()[0].view(np.uint32), u0) and np.array_equal(g.ctx(2).bvh()[0].view(np.uint32), g.ctx(0).bvh()[0].view(np.uint32))
    g.ctx(1).scene_refit()                          # on its own: copy-on-write
    assert g.ctx(1).host_build_id() != g.ctx(0).host_build_id() == g.ctx(2).host_build_id()
    with pytest.raises(pbr.PtcError):
        g.render(8, 8, 1)                           # no device


def test_launch_policy_is_reported_without_a_device(pbr):
    """ptc_launch_policy(NULL): the kernels' compile-time constants and the built-in defaults, as one string; a context's string starts with the same constants and
    carries its own knobs (here: the builder).  tests/test_profiles.py holds the committed kernel model against the first."""
    L = pbr.load_library()
    dflt = L.ptc_launch_policy(None).decode()
    for key in ("trace_block=", "ring=", "refill_idle=", "node_min=", "chunk=", "shade_block=", "nodelets=", "lanes=", "batch_paths=", "trace_overlap=", "bvh=sah"):
        assert key in dflt, (key, dflt)
    import copy
    d = copy.deepcopy(pbr.scenes.by_name("cornell")); d.bvh_builder = "lbvh"
    pt = pbr.PathTracer(pbr.DEVICE_NONE).load_scene(d)
    mine = pt.launch_policy()
    assert mine.split(" | ")[0] == dflt.split(" | ")[0] and "bvh=lbvh" in mine and "trace_blocks_per_cu" not in mine      # nothing was configured for a device



def test_a_refused_host_refit_leaves_the_host_build_as_it_was(pbr):
    """A transform that overflows to non-finite positions is refused by ptc_scene_refit; the host build (what the debug getters describe, and what a later
    refit starts from) must still be the committed one — vertices, tree and tables."""
    d = pbr.scenes.by_name("textured_objects")
    pt = pbr.PathTracer(pbr.DEVICE_NONE).load_scene(d)
    v0, i0, m0 = pt.flat_scene()
    u0 = pt.bvh()[0].view(np.uint32).copy()
    s0 = pt.shading_tables()[0].view(np.uint32).copy()
    big = np.eye(4, dtype=np.float32); big[0, 0] = 3e38; big[3, 0] = 3e38
    pt.update_instance(0, matrix=big.reshape(16))
    with pytest.raises(pbr.PtcError, match="non-finite"):
        pt.scene_refit()
    v1, _, _ = pt.flat_scene()
    assert np.array_equal(v0.view(np.uint32), v1.view(np.uint32)) and np.array_equal(pt.bvh()[0].view(np.uint32), u0)
    assert np.array_equal(pt.shading_tables()[0].view(np.uint32), s0)
    it = d.instances[0]
    if getattr(it, "matrix", None) is not None:
        pt.update_instance(0, matrix=it.matrix)
    else:
        pt.update_instance(0, it.t, it.q_wxyz, it.s)
    pt.scene_refit()                                   # the original transform again: the committed bytes
    assert np.array_equal(pt.bvh()[0].view(np.uint32), u0) and np.array_equal(pt.flat_scene()[0].view(np.uint32), v0.view(np.uint32))
