"""Textures and the environment light (BASELINE config 5) in the oracle: closed-form checks of the new estimator
pieces, on the CPU.  (GPU-vs-oracle bit parity of the same scenes is in test_gpu_parity.py.)"""
import math

import numpy as np
import pytest


def _plane_scene(pbr, material, textures=(), env=None, uv_scale=1.0):
    sc = pbr.scene
    v, i = pbr.scenes._quad((-50, 0, 50), (50, 0, 50), (50, 0, -50), (-50, 0, -50))      # normal +y
    v["texCoords"] = v["texCoords"] * uv_scale
    return sc.SceneDesc([material], [sc.MeshDesc(v, i, 0)], [sc.InstanceDesc(0)], sc.CameraDesc((0, 3, 0), (0, 0, -0.001), 0.6, 1.0),
                        textures=list(textures), env=env)


def test_atan2_accuracy(ora):
    rng = np.random.default_rng(0)
    for y, x in rng.normal(0, 1, (4000, 2)):
        assert abs(ora.lib().ora_atan2f(float(y), float(x)) - math.atan2(np.float32(y), np.float32(x))) < 2e-5
    assert ora.lib().ora_atan2f(0.0, 0.0) == 0.0 and abs(ora.lib().ora_atan2f(0.0, -1.0) - math.pi) < 1e-6


def test_constant_environment_on_a_diffuse_plane(ora, pbr):
    """Radiance 1 from every direction on an infinite-ish Lambert plane of albedo rho (convex: no interreflection) gives exactly rho;
    NEE and BSDF sampling are combined by MIS, so this checks pdfs and weights of both environment strategies."""
    env = np.ones((16, 32, 3), np.float32)
    rho = 0.6
    d = _plane_scene(pbr, pbr.scene.Material((rho, rho, rho, 1.0), 0.0, 1.0), env=env)
    img = ora.Oracle().load_scene(d).render(32, 32, 64, seed=1, max_bounces=4)
    assert abs(float(img[..., :3].mean()) - rho) < 0.01
    # a GGX dielectric plane under the same furnace stays <= 1 and > the diffuse part
    d2 = _plane_scene(pbr, pbr.scene.Material((rho, rho, rho, 1.0), 0.0, 0.4), env=env)
    m2 = float(ora.Oracle().load_scene(d2).render(32, 32, 64, seed=1, max_bounces=4)[..., :3].mean())
    assert rho - 0.02 < m2 < 1.05


def test_small_bright_sun_needs_and_gets_importance_sampling(ora, pbr):
    """A 6-degree sun carrying most of the power: the estimate at 64 spp is already within a few % of the 1024 spp one
    (BSDF sampling alone would be far noisier), and equals the analytic irradiance of the map."""
    env = pbr.scenes.analytic_sky(128, 64)
    d = _plane_scene(pbr, pbr.scene.Material((0.5, 0.5, 0.5, 1.0), 0.0, 1.0), env=env)
    o = ora.Oracle().load_scene(d)
    a = o.render(24, 24, 64, seed=3, max_bounces=1)[..., :3].mean((0, 1))
    b = o.render(24, 24, 1024, seed=4, max_bounces=1)[..., :3].mean((0, 1))
    assert np.allclose(a, b, rtol=0.05)
    # analytic: E = sum over texels of L * cos(theta) * d_omega (upper hemisphere), radiance = rho/pi * E
    h, w = env.shape[:2]
    th = np.pi * (np.arange(h) + 0.5) / h
    dom = (2 * np.pi / w) * (np.pi / h) * np.sin(th)
    E = (env * (np.clip(np.cos(th), 0, None) * dom)[:, None, None]).sum((0, 1))
    assert np.allclose(b, 0.5 / np.pi * E, rtol=0.04)


def test_base_colour_texture_nearest_repeat(ora, pbr):
    """2x2 texture, NEAREST/REPEAT (the reference's default sampler, gltf/Asset.cpp:116-117): under a constant environment each
    visible texel shows base_color_factor * texel, and uv outside [0,1) wraps."""
    tex = np.array([[[255, 0, 0, 255], [0, 255, 0, 255]], [[0, 0, 255, 255], [255, 255, 255, 255]]], np.uint8)
    env = np.ones((8, 16, 3), np.float32)
    mat = pbr.scene.Material((0.5, 1.0, 1.0, 1.0), 0.0, 1.0, (0, 0, 0), 0, -1, -1)
    d = _plane_scene(pbr, mat, [tex], env, uv_scale=3.0)                      # uv in [0,3): the 2x2 pattern repeats three times
    img = ora.Oracle().load_scene(d).render(48, 48, 16, seed=2, max_bounces=2)[..., :3]
    cols = {tuple(np.round(c, 2)) for c in img.reshape(-1, 3)[::7]}
    expect = [(0.5, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0), (0.5, 1.0, 1.0)]
    for e in expect:
        assert any(np.allclose(c, e, atol=0.08) for c in cols), e
    flat = img.reshape(-1, 3)
    ok = sum(any(np.allclose(p, e, atol=0.12) for e in expect) for p in flat[::5])
    assert ok > 0.9 * len(flat[::5])                                          # every pixel is one of the four texel colours (pixel-footprint mixing aside)


def test_normal_map_tilts_the_shading_normal(ora, pbr):
    """Raster-compat (fragment.glsl:24-27): with a constant tangent-space normal (sx, 0, sqrt(1-sx^2)) the shaded N·V changes as
    TBN * texel predicts; a flat (0.5,0.5,1) texel reproduces the untextured result bit for bit."""
    flat = np.full((4, 4, 4), (128, 128, 255, 255), np.uint8)                 # not exactly (0,0,1): 128/255*2-1 = 0.0039
    exact = np.full((2, 2, 4), (255, 255, 255, 255), np.uint8)
    base = pbr.scene.Material((0.8, 0.8, 0.8, 1.0), 0.0, 1.0)
    r0 = ora.Oracle().load_scene(_plane_scene(pbr, base)).render(16, 16, 1, integrator=1)
    white = pbr.scene.Material((0.8, 0.8, 0.8, 1.0), 0.0, 1.0, (0, 0, 0), 0, -1, -1)
    r1 = ora.Oracle().load_scene(_plane_scene(pbr, white, [exact])).render(16, 16, 1, integrator=1)
    assert np.array_equal(r0, r1)                                            # albedo * 1.0 and no normal map: identical
    nm = pbr.scene.Material((0.8, 0.8, 0.8, 1.0), 0.0, 1.0, (0, 0, 0), -1, 0, -1)
    r2 = ora.Oracle().load_scene(_plane_scene(pbr, nm, [flat])).render(16, 16, 1, integrator=1)
    assert np.abs(r2 - r0).max() < 0.1 and not np.array_equal(r2, r0)   # 128/255*2-1 = 0.004: almost, not exactly, flat (pow(.,64) amplifies)
    tilt = np.full((2, 2, 4), (230, 128, 200, 255), np.uint8)
    r3 = ora.Oracle().load_scene(_plane_scene(pbr, nm, [tilt])).render(16, 16, 1, integrator=1)
    assert np.abs(r3 - r0).max() > 0.15


def test_environment_argument_errors(ora, pbr):
    o = ora.Oracle()
    L = ora.lib()
    assert L.ora_set_env_latlong_rgb32f(o._h, np.zeros(3, np.float32).ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_float)), 0, 1) < 0
    pt = pbr.PathTracer(pbr.DEVICE_NONE)
    with pytest.raises(pbr.PtcError, match="texture id out of range"):
        pt.load_scene(_plane_scene(pbr, pbr.scene.Material((1, 1, 1, 1), 0.0, 1.0, (0, 0, 0), 3, -1, -1)))


def test_bilinear_filter_option(ora, pbr):
    """PTC_FILTER_LINEAR (an option beyond the reference, whose samplers are NEAREST): a 2-texel-wide black→white texture under a
    white furnace shows a linear ramp between the texel centres, wraps around (REPEAT), and the NEAREST result stays a step."""
    tex = np.zeros((1, 2, 4), np.uint8)
    tex[0, 1] = 255
    tex[..., 3] = 255
    env = np.ones((8, 16, 3), np.float32)
    mat = pbr.scene.Material((1.0, 1.0, 1.0, 1.0), 0.0, 1.0, (0, 0, 0), 0, -1, -1)
    d = _plane_scene(pbr, mat, [tex], env, uv_scale=1.0)
    d.camera = pbr.scene.CameraDesc((0, 60, 0), (0, 0, -0.001), 1.2, 1.0)      # sees most of the 100-unit plane = one uv period
    o = ora.Oracle()
    near = o.load_scene(d).render(64, 64, 32, seed=2, max_bounces=2)[..., 0]
    d.texture_filter = "linear"
    lin = o.load_scene(d).render(64, 64, 32, seed=2, max_bounces=2)[..., 0]
    row_n, row_l = near.mean(0), lin.mean(0)
    assert set(np.round(row_n, 1)) <= {0.0, 1.0}                                # a step
    # between the two texel centres (u = 1/4 and 3/4) the ramp is linear in u; u maps linearly to the pixel column here
    inner = row_l[20:44]
    k = np.arange(inner.size)
    slope, icpt = np.polyfit(k, inner, 1)
    assert abs(slope) > 0.02 and np.abs(inner - (slope * k + icpt)).max() < 0.02, (slope, inner)      # a straight ramp (Monte-Carlo noise aside)
    assert 0.4 < row_l[32] < 0.6 and abs(row_l.mean() - 0.5) < 0.03                 # symmetric wrap-around: the mean stays 1/2
    # exact texels: sampling at a texel centre returns the texel
    assert min(row_l) < 0.05 and max(row_l) > 0.95
    d.texture_filter = "nearest"                                                  # scene_begin resets the filter
    assert np.array_equal(o.load_scene(d).render(64, 64, 32, seed=2, max_bounces=2)[..., 0], near)


def test_environment_azimuth_and_polar_convention(ora, pbr):
    """The lat-long convention of include/ptc.h — row 0 = +y, u = atan2(d.z, d.x)/(2 pi) + 1/2 — pinned against physics: one bright
    texel block; a small Lambert quad facing direction n receives E = L * d_omega * max(0, n.d) from it, with d the direction the
    DOCUMENTED formula gives for that block.  Facing it: full; 60 degrees off: half; facing away or edge-on: nothing."""
    sc = pbr.scene
    h, w = 32, 64
    r0, c0 = 9, 41                                             # a 2x2 block well away from poles and seam
    env = np.zeros((h, w, 3), np.float32)
    L = 4000.0
    env[r0:r0 + 2, c0:c0 + 2] = L
    theta = np.pi * (r0 + 1.0) / h                             # block centre: v = (r0 + 1)/h, theta from +y
    a = 2.0 * np.pi * (c0 + 1.0) / w                           # u = (c0 + 1)/w = atan2(d.z, d.x)/(2 pi) + 1/2  =>  atan2(d.z, d.x) = a - pi
    dsun = np.array([np.sin(theta) * np.cos(a - np.pi), np.cos(theta), np.sin(theta) * np.sin(a - np.pi)])
    dom = sum((2 * np.pi / w) * (np.pi / h) * np.sin(np.pi * (r + 0.5) / h) * 2 for r in (r0, r0 + 1))   # two columns per row
    rho = 0.8

    def radiance(n):
        n = np.asarray(n, np.float64) / np.linalg.norm(n)
        t = np.cross(n, [0.3, 0.5, 0.8]); t /= np.linalg.norm(t)
        b = np.cross(n, t)
        k = 0.5
        v, i = pbr.scenes._quad(tuple(-k * t - k * b), tuple(k * t - k * b), tuple(k * t + k * b), tuple(-k * t + k * b))
        nn = np.cross(np.asarray(v["position"][1]) - np.asarray(v["position"][0]), np.asarray(v["position"][2]) - np.asarray(v["position"][0]))
        if np.dot(nn, n) < 0:                                  # wind the quad so that its front faces n
            i = i.reshape(-1, 3)[:, ::-1].reshape(-1).copy()
        v["normal"][:] = n.astype(np.float32)
        cam = sc.CameraDesc(tuple(3.0 * n), (0.0, 0.0, 0.0), 0.05, 1.0)
        d = sc.SceneDesc([sc.Material((rho, rho, rho, 1.0), 0.0, 1.0)], [sc.MeshDesc(v, i, 0)], [sc.InstanceDesc(0)], cam, env=env)
        img = ora.Oracle().load_scene(d).render(8, 8, 256, seed=5, max_bounces=1)
        return float(img[..., :3].mean())

    full = rho / np.pi * L * dom
    assert abs(radiance(dsun) - full) < 0.03 * full
    # 60 degrees off: rotate dsun about an axis perpendicular to it
    ax = np.cross(dsun, [0.0, 1.0, 0.0]); ax /= np.linalg.norm(ax)
    n60 = dsun * 0.5 + np.cross(ax, dsun) * np.sqrt(0.75)
    assert abs(radiance(n60) - 0.5 * full) < 0.03 * full
    assert radiance(-dsun) < 1e-3 * full
    # a quad facing the z-mirrored direction (what the formula with the other sign of z would call the sun) gets the cosine only
    wrong = dsun * np.array([1.0, 1.0, -1.0])
    assert abs(radiance(wrong) - max(0.0, float(np.dot(wrong, dsun))) * full) < 0.03 * full and float(np.dot(wrong, dsun)) < 0.9


@pytest.mark.parametrize("g,b", [(128, 255), (64, 0), (200, 128)])
def test_metal_rough_texture_channels_are_gltf(ora, pbr, g, b):
    """glTF metallicRoughnessTexture: roughness in G, metallic in B, both multiplied by the factors.  A uniform texel (R = 77,
    G = g, B = b) on a material with factors 1 must render, bit for bit, what the untextured material with roughness g/255 and
    metallic b/255 renders (R is ignored); with factors 1/2 the products are what counts."""
    env = np.ones((8, 16, 3), np.float32)
    tex = np.full((4, 4, 4), 255, np.uint8)
    tex[..., 0], tex[..., 1], tex[..., 2] = 77, g, b
    base = (0.9, 0.6, 0.3, 1.0)
    f = lambda x: float(np.float32(x) / np.float32(255))
    for fm, fr in ((1.0, 1.0), (0.5, 0.5)):
        textured = _plane_scene(pbr, pbr.scene.Material(base, fm, fr, (0, 0, 0), -1, -1, 0), [tex], env)
        plain = _plane_scene(pbr, pbr.scene.Material(base, float(np.float32(fm) * np.float32(f(b))), float(np.float32(fr) * np.float32(f(g)))), env=env)
        a = ora.Oracle().load_scene(textured).render(24, 24, 8, seed=9, max_bounces=3)
        c = ora.Oracle().load_scene(plain).render(24, 24, 8, seed=9, max_bounces=3)
        assert np.array_equal(a.view(np.uint32), c.view(np.uint32))
    swapped = _plane_scene(pbr, pbr.scene.Material(base, f(g), f(b)), env=env)          # G and B read the other way round: a different image
    w = ora.Oracle().load_scene(swapped).render(24, 24, 8, seed=9, max_bounces=3)
    if g != b:
        assert not np.array_equal(w.view(np.uint32), c.view(np.uint32))
