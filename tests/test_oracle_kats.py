"""Known-answer tests that pin the oracle's restatement of the reference's conventions (R3, R5, R9)
and of its own building blocks (RNG, sin/cos, pow, tile walk).  The reference has no test for any of
these: expected values are hand-computed or come from an independent float64 numpy
restatement of the published closed forms (glm 1.0.1 lookAtRH / perspectiveRH_NO / mat4_cast)."""
import math

import numpy as np
import pytest


# ---- R3: makeModelPushConstant (ModelPushConstant.hpp:33-46) -------------------------------------
def test_model_identity(ora):
    M, N = ora.make_model((0, 0, 0), (1, 0, 0, 0), (1, 1, 1))
    assert np.array_equal(M, np.eye(4, dtype=np.float32))
    assert np.array_equal(N, np.eye(3, dtype=np.float32))


def test_model_trs_hand_computed(ora):
    # 90° about +z: q = (cos45°, 0, 0, sin45°); scale (2,3,4); translate (5,6,7).
    # R = [[0,-1,0],[1,0,0],[0,0,1]] (row-major) → columns (0,1,0), (-1,0,0), (0,0,1); model col j = R col j * s_j.
    c = math.sqrt(0.5)
    M, N = ora.make_model((5, 6, 7), (c, 0, 0, c), (2, 3, 4))
    exp = np.array([[0, 2, 0, 0], [-3, 0, 0, 0], [0, 0, 4, 0], [5, 6, 7, 1]], np.float32)  # [col][row]
    assert np.allclose(M, exp, atol=1e-6)
    # normal matrix = inverse-transpose = R * diag(1/s)
    expN = np.array([[0, 0.5, 0], [-1 / 3, 0, 0], [0, 0, 0.25]], np.float32)
    assert np.allclose(N, expN, atol=1e-6)


def test_model_matches_float64_closed_form(ora):
    rng = np.random.default_rng(0)
    for _ in range(50):
        t = rng.normal(0, 3, 3)
        q = rng.normal(0, 1, 4)
        q /= np.linalg.norm(q)
        s = rng.uniform(0.3, 3.0, 3)
        M, N = ora.make_model(t, q, s)
        w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        A = R @ np.diag(s)
        assert np.allclose(M[:3, :3].T, A, atol=2e-5)  # M is [col][row]
        assert np.allclose(M[3, :3], t, atol=1e-6)
        assert np.allclose(N.T, np.linalg.inv(A).T, atol=1e-4)


# ---- R5: makeCameraData (CameraData.hpp:22-32) ----------------------------------------------------
def test_camera_default_hand_computed(ora):
    # reference defaults: position 0, direction -z, fovY pi/2 (CameraController.hpp:25-40), aspect 1
    V, P = ora.make_camera((0, 0, 0), (0, 0, -1), math.pi / 2, 1.0)
    # lookAtRH with up=(0,-1,0): s = f x up = (-1,0,0), u = s x f = (0,-1,0)  →  view = diag(-1,-1,1,1)
    assert np.allclose(V, np.diag([-1, -1, 1, 1]).astype(np.float32), atol=1e-7)
    n, f = 0.01, 1024.0
    exp = np.zeros((4, 4), np.float32)
    exp[0][0] = 1.0
    exp[1][1] = 1.0
    exp[2][2] = -(f + n) / (f - n)
    exp[2][3] = -1.0
    exp[3][2] = -(2 * f * n) / (f - n)
    assert np.allclose(P, exp, rtol=1e-6, atol=1e-7)


def test_camera_matches_host_mirror(ora, pbr):
    rng = np.random.default_rng(1)
    for _ in range(20):
        pos = rng.normal(0, 5, 3)
        tgt = pos + rng.normal(0, 1, 3)
        fov = rng.uniform(0.1, math.pi / 2)
        asp = rng.uniform(0.5, 2.5)
        V, P = ora.make_camera(pos, tgt, fov, asp)
        cd = pbr.scene.make_camera_data(pos, tgt, fov, asp)
        assert np.allclose(V, cd.view, atol=2e-5)
        assert np.allclose(P, cd.proj, rtol=1e-5, atol=1e-6)


# ---- R9: aces+gamma.glsl ----------------------------------------------------------------------------
def _tonemap_f64(rgba):
    """Independent float64 restatement with GLSL column-major matrix semantics."""
    a_in = np.array([[0.59719, 0.35458, 0.04823], [0.07600, 0.90834, 0.01566], [0.02840, 0.13383, 0.83777]])  # columns
    a_out = np.array([[1.60475, -0.53108, -0.07367], [-0.10208, 1.10813, -0.00605], [-0.00327, -0.07276, 1.07602]])
    c = rgba[..., :3].astype(np.float64)
    x = c[..., 0:1] * a_in[0] + c[..., 1:2] * a_in[1] + c[..., 2:3] * a_in[2]
    x = (x * (x + 0.0245786) - 0.000090537) / (x * (0.983729 * x + 0.4329510) + 0.238081)
    y = x[..., 0:1] * a_out[0] + x[..., 1:2] * a_out[1] + x[..., 2:3] * a_out[2]
    y = np.power(np.maximum(y, 0.0), 1.0 / 2.2)
    out = np.concatenate([y, rgba[..., 3:4].astype(np.float64)], axis=-1)
    return np.floor(np.clip(out, 0.0, 1.0) * 255.0 + 0.5).astype(np.int32)


def test_tonemap_kat_table(ora):
    table = np.array([
        [0, 0, 0, 1], [1, 1, 1, 1], [0.18, 0.18, 0.18, 1], [1, 0, 0, 1], [0, 1, 0, 0.5], [0, 0, 1, 0],
        [4, 2, 1, 1], [0.01, 0.02, 0.03, 1], [15, 15, 15, 1], [0.5, 0.25, 0.125, 0.25], [1e-4, 1e-4, 1e-4, 1],
        [100, 0.1, 0.1, 1], [0.05, 0.4, 0.9, 2.0], [0.73, 0.73, 0.73, 1], [0.65, 0.05, 0.05, 1], [0.12, 0.45, 0.15, -1],
    ], np.float32)
    got = ora.tonemap_rgba8(table).astype(np.int32)
    exp = _tonemap_f64(table)
    assert np.abs(got - exp).max() <= 1
    assert (got[0, :3] == 0).all() and got[0, 3] == 255          # black stays black (negative fit clamped before pow)
    # radiance 15 does NOT go to white: the transposed ACES matrices leave a magenta cast
    assert got[8, 0] >= 250 and got[8, 2] >= 250 and got[8, 1] < 200
    assert got[15, 3] == 0 and got[12, 3] == 255                 # alpha clamps
    rnd = np.random.default_rng(3).uniform(0, 3, (4096, 4)).astype(np.float32)
    assert np.abs(ora.tonemap_rgba8(rnd).astype(np.int32) - _tonemap_f64(rnd)).max() <= 1


# ---- building blocks of the path tracer ----------------------------------------------------------------
def test_sincos_accuracy(ora):
    us = np.concatenate([np.linspace(0, 1, 4001, endpoint=False), np.random.default_rng(5).uniform(0, 1, 4000)]).astype(np.float32)
    err = 0.0
    for u in us:
        s, c = ora.sincos2pi(float(u))
        err = max(err, abs(s - math.sin(2 * math.pi * float(u))), abs(c - math.cos(2 * math.pi * float(u))))
    assert err < 4e-7


def test_pow_accuracy(ora):
    rng = np.random.default_rng(6)
    xs = np.exp(rng.uniform(-12, 6, 3000)).astype(np.float32)
    for x in xs:
        for y in (1 / 2.2, 2.0, 0.5, 5.0):
            ref = float(x) ** y
            # the float32 rounding of y*log2(x) bounds the relative error: ~ ulp * |y log2 x|
            assert abs(ora.powf(float(x), y) - ref) <= 2e-6 * (2.0 + abs(y * math.log2(float(x)))) * ref
    assert ora.powf(0.0, 0.4545) == 0.0 and ora.powf(-1.0, 2.0) == 0.0


def _pcg(v):
    s = (v * 747796405 + 2891336453) & 0xFFFFFFFF
    w = (((s >> ((s >> 28) + 4)) ^ s) * 277803737) & 0xFFFFFFFF
    return ((w >> 22) ^ w) & 0xFFFFFFFF


def test_rng_matches_python_restatement(ora):
    rng = np.random.default_rng(8)
    for _ in range(200):
        seed = int(rng.integers(0, 2**63))
        pixel, sample, bounce, dim = (int(rng.integers(0, 2**21)), int(rng.integers(0, 4096)), int(rng.integers(0, 10)), int(rng.integers(0, 8)))
        key = _pcg((pixel + _pcg((sample + _pcg(((seed & 0xFFFFFFFF) + _pcg(seed >> 32)) & 0xFFFFFFFF)) & 0xFFFFFFFF)) & 0xFFFFFFFF)
        exp = _pcg(_pcg((bounce * 8 + dim) & 0xFFFFFFFF) ^ key)
        assert ora.rng_u32(seed, pixel, sample, bounce, dim) == exp


def test_rng_uniformity(ora):
    xs = np.array([ora.rng_u32(1, p, s, 1, 4) for p in range(64) for s in range(64)], np.float64) / 2**32
    assert abs(xs.mean() - 0.5) < 0.02 and abs(xs.var() - 1 / 12) < 0.01


@pytest.mark.parametrize("w,h,n", [(64, 64, 2), (100, 70, 3), (33, 1, 2), (256, 135, 8)])
def test_tile_owner_matches_python_walk(ora, pbr, w, h, n):
    masks = [pbr.dist.owned_mask(w, h, r, n) for r in range(n)]
    assert np.array_equal(sum(m.astype(int) for m in masks), np.ones((h, w), int))  # disjoint cover
    rng = np.random.default_rng(9)
    for _ in range(200):
        x, y = int(rng.integers(0, w)), int(rng.integers(0, h))
        assert masks[ora.tile_owner(w, h, x, y, n)][y, x]


def test_float16_rounding_equals_numpy(ora):
    """fp32 -> fp16 of the RGBA16F contract (PbrRenderSystem.hpp:21, GBuffer.hpp:13-14): round to nearest even, like numpy's
    float16 — every half round-trips, every midpoint between neighbouring halves ties to even, random floats and bit patterns."""
    import numpy as np

    for h in range(0, 65536, 1):
        ref = np.array([h], np.uint16).view(np.float16).astype(np.float32)[0]
        f = ora.f16_to_f32(h)
        if np.isnan(ref):
            assert np.isnan(f)
            continue
        assert np.float32(f).view(np.uint32) == ref.view(np.uint32) and ora.f32_to_f16(f) == h
    with np.errstate(over="ignore"):
        for h in range(0, 0x7C00, 7):
            a, b = np.float64(ora.f16_to_f32(h)), np.float64(ora.f16_to_f32(h + 1))
            mid = np.float32((a + b) / 2)
            if np.float64(mid) == (a + b) / 2:
                assert ora.f32_to_f16(mid) == int(mid.astype(np.float16).view(np.uint16)), h
        rng = np.random.default_rng(1)
        xs = (rng.standard_normal(20000) * 10.0 ** rng.uniform(-9, 6, 20000)).astype(np.float32)
        bits = rng.integers(0, 2 ** 32, 20000, dtype=np.uint64).astype(np.uint32).view(np.float32)
        for arr in (xs, bits):
            want = arr.astype(np.float16).view(np.uint16)
            for x, r in zip(arr, want):
                g = ora.f32_to_f16(x)
                if np.isnan(x):
                    assert (g & 0x7C00) == 0x7C00 and (g & 0x3FF)
                else:
                    assert g == r, (x, hex(g), hex(int(r)))


def test_raster_gbuffer16_is_the_quantised_lighting_pass(ora, pbr):
    """Integrator 2 = lighting.glsl:19-29 fed from the reference's G-buffer formats (GBuffer.hpp:13-16): recomputed here in numpy
    float64 from the fp32 raster-compat inputs of one flat-shaded pixel row: P and N rounded to fp16, albedo to UNORM16."""
    import numpy as np

    d = pbr.scenes.two_triangles_and_sphere()
    o = ora.Oracle().load_scene(d)
    a = o.render(64, 64, 1, integrator=1)
    b = o.render(64, 64, 1, integrator=2)
    hit = a[..., 3] > 0
    assert (b[~hit] == 0).all() and hit.sum() > 500
    assert not np.array_equal(a, b)
    err = np.abs(a[hit] - b[hit]).max()
    assert 0 < err < 2e-2                                       # quantisation of P (|P| < 8 -> ulp16 <= 2^-8), N and albedo: small but there
    # a material whose albedo is exactly representable in UNORM16 and a head-on plane: only P and N rounding remain
    sc = pbr.scene
    v, i = pbr.scenes._quad((-4, -4, -2), (4, -4, -2), (4, 4, -2), (-4, 4, -2))
    for base in ((1.0, 0.0, 1.0, 1.0), (0.2, 0.4, 0.6, 1.0)):
        dd = sc.SceneDesc([sc.Material(base, 0.0, 1.0)], [sc.MeshDesc(v, i, 0)], [sc.InstanceDesc(0, (0, 0, 0), (1, 0, 0, 0), (1, 1, 1))],
                          sc.CameraDesc((0, 0, 0), (0, 0, -1), 1.0, 1.0))
        oo = ora.Oracle().load_scene(dd)
        img = oo.render(9, 9, 1, integrator=2)
        c = img[4, 4]                                           # centre pixel: P = (0,0,-2) exactly, N = (0,0,1): NdotV = 1, spec = 1
        q = np.round(np.clip(np.asarray(base, np.float64), 0, 1) * 65535) / 65535
        assert np.allclose(c, q * 1.0 + 1.0, rtol=0, atol=2e-7), (c, q)


def test_r6_front_face_is_the_vulkan_clockwise_rule(ora, pbr):
    """R6: the reference culls back faces with frontFace = eClockwise (PbrRenderSystem.cpp:186).  Vulkan decides facing from the
    sign of the polygon's area in FRAMEBUFFER coordinates (y down, un-flipped viewport): a = -1/2 sum(x_i y_{i+1} - x_{i+1} y_i);
    clockwise-front means a < 0 is front.  The raster-compat integrator must show exactly the triangles that rule calls front,
    with the framebuffer positions coming from the reference's own view / projection matrices (KAT'd above), not from the
    path tracer's determinant test."""
    sc = pbr.scene
    rng = np.random.default_rng(21)
    n_front = n_back = 0
    for k in range(60):
        pos = rng.uniform(-2, 2, 3)
        tgt = pos + rng.normal(size=3)
        fov, asp = 0.9, 1.0
        V, P = ora.make_camera(tuple(pos), tuple(tgt), fov, asp)
        fwd = (tgt - pos) / np.linalg.norm(tgt - pos)
        ctr = pos + 4.0 * fwd                                             # a triangle around the view axis, 4 units ahead
        tri = ctr + rng.normal(scale=0.6, size=(3, 3))
        clip = np.concatenate([tri, np.ones((3, 1))], 1) @ V.astype(np.float64) @ P.astype(np.float64)
        assert (clip[:, 3] > 0).all()
        ndc = clip[:, :2] / clip[:, 3:4]
        fb = (ndc + 1.0) * 0.5 * 64.0                                     # y-down framebuffer, no flip (PbrRenderSystem.cpp:425-430)
        area = -0.5 * sum(fb[i, 0] * fb[(i + 1) % 3, 1] - fb[(i + 1) % 3, 0] * fb[i, 1] for i in range(3))
        if abs(area) < 20.0 or (np.abs(ndc) > 0.95).any():                # skip slivers and triangles leaving the frame
            continue
        front = area < 0.0                                                # VK_FRONT_FACE_CLOCKWISE
        v = np.zeros(3, sc.MESH_VERTEX)
        v["position"] = tri.astype(np.float32)
        nrm = np.cross(tri[1] - tri[0], tri[2] - tri[0]); nrm /= np.linalg.norm(nrm)
        v["normal"] = nrm.astype(np.float32)
        v["tangent"] = (1, 0, 0, 1)
        d = sc.SceneDesc([sc.Material((0.8, 0.8, 0.8, 1.0))], [sc.MeshDesc(v, np.array([0, 1, 2], np.uint32), 0)], [sc.InstanceDesc(0)],
                         sc.CameraDesc(tuple(pos), tuple(tgt), fov, asp))
        img = ora.Oracle().load_scene(d).render(64, 64, 1, integrator=1)
        visible = bool((img[..., 3] > 0).any())
        assert visible == front, (k, area)
        n_front += front; n_back += (not front)
    assert n_front >= 8 and n_back >= 8


@pytest.mark.parametrize("tilt_deg", [0.0, 20.0, 45.0, 70.0])
def test_r8_blinn_phong_with_the_light_at_the_camera(ora, pbr, tilt_deg):
    """R8: lighting.glsl:25-28 sets L = V, so BlinnPhong.lib.glsl:4-10 reduces to  out = albedo * NdotV + NdotV^64  on all four
    channels.  A plane through (0,0,-3) tilted about y, seen by the centre pixel along -z: NdotV = cos(tilt), in float64."""
    sc = pbr.scene
    a = math.radians(tilt_deg)
    n = np.array([math.sin(a), 0.0, math.cos(a)])
    t, b, c = np.array([math.cos(a), 0.0, -math.sin(a)]), np.array([0.0, 1.0, 0.0]), np.array([0.0, 0.0, -3.0])
    base = (0.7, 0.35, 0.15, 0.6)
    shown = None
    for order in ((0, 1, 2, 3), (3, 2, 1, 0)):                       # one of the two windings faces the camera (R6)
        corners = [c - 4 * t - 4 * b, c + 4 * t - 4 * b, c + 4 * t + 4 * b, c - 4 * t + 4 * b]
        v, i = pbr.scenes._quad(*[tuple(corners[k]) for k in order])
        v["normal"][:] = n.astype(np.float32)
        d = sc.SceneDesc([sc.Material(base, 0.0, 1.0)], [sc.MeshDesc(v, i, 0)], [sc.InstanceDesc(0)], sc.CameraDesc((0, 0, 0), (0, 0, -1), 0.8, 1.0))
        img = ora.Oracle().load_scene(d).render(9, 9, 1, integrator=1)
        if img[4, 4, 3] > 0:
            shown = img[4, 4].astype(np.float64)
    assert shown is not None
    ndv = math.cos(a)
    expect = np.asarray(base, np.float64) * ndv + ndv ** 64
    assert np.allclose(shown, expect, rtol=2e-5, atol=2e-6), (shown, expect)
