"""Analytic known answers for the path-tracer specification itself.  The reference has no ray tracer, so the oracle is
"parity unpinned" against it; these tests pin it against physics instead: closed forms and quadrature
that any correct unidirectional path tracer with next-event estimation, MIS and Russian roulette must reproduce in the
mean.  The HIP path equals the oracle bit for bit (tests/test_gpu_parity.py), so they hold for the product too."""
import math

import numpy as np
import pytest


def _box(pbr, mat_ids, lo=(-1, -1, -1), hi=(1, 1, 1)):
    """Closed axis-aligned box, all faces looking inward (the six quads of scenes.cornell_box plus the front)."""
    q = pbr.scenes._quad
    (x0, y0, z0), (x1, y1, z1) = lo, hi
    quads = [q((x0, y0, z1), (x1, y0, z1), (x1, y0, z0), (x0, y0, z0)),    # y = y0, normal +y
             q((x0, y1, z0), (x1, y1, z0), (x1, y1, z1), (x0, y1, z1)),    # y = y1, normal -y
             q((x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0)),    # z = z0, normal +z
             q((x1, y0, z1), (x0, y0, z1), (x0, y1, z1), (x1, y1, z1)),    # z = z1, normal -z
             q((x0, y0, z1), (x0, y0, z0), (x0, y1, z0), (x0, y1, z1)),    # x = x0, normal +x
             q((x1, y0, z0), (x1, y0, z1), (x1, y1, z1), (x1, y1, z0))]    # x = x1, normal -x
    return [pbr.scene.MeshDesc(v, i, m) for (v, i), m in zip(quads, mat_ids)]


@pytest.mark.parametrize("rho,max_bounces", [(0.5, 8), (0.8, 8), (0.8, 3), (0.3, 0), (0.95, 12)])
def test_emitting_furnace_geometric_series(ora, pbr, rho, max_bounces):
    """A closed box whose every wall emits Le and reflects rho (Lambert): radiance seen from inside is
    Le * sum_{k=0..B} rho^k for bounce limit B.  Exercises emission + NEE with MIS on every bounce, the throughput
    update and Russian roulette (from bounce 3 on) — a biased roulette or a wrong MIS weight breaks the series."""
    sc = pbr.scene
    le = 2.0
    mat = sc.Material((rho, rho, rho, 1.0), 0.0, 1.0, (le, le, le))
    meshes = _box(pbr, [0] * 6)
    d = sc.SceneDesc([mat], meshes, [sc.InstanceDesc(k) for k in range(6)], sc.CameraDesc((0.1, -0.2, 0.3), (0.4, 0.1, -1.0), 1.2, 1.0))
    img = ora.Oracle().load_scene(d).render(48, 48, 64, seed=5, max_bounces=max_bounces)[..., :3]
    want = le * sum(rho ** k for k in range(max_bounces + 1))
    assert abs(float(img.mean()) / want - 1.0) < 0.01, (float(img.mean()), want)
    assert np.allclose(img.mean((0, 1)), want, rtol=0.015)


def test_furnace_with_ggx_walls_conserves_energy(ora, pbr):
    """Same furnace with rough-dielectric and metal walls: the BSDF (single-scattering GGX + Lambert) may lose energy but
    never creates it, so the radiance stays below the Lambert series for albedo 1 and above the pure emission."""
    sc = pbr.scene
    le = 1.0
    for metallic, rough in ((0.0, 0.5), (1.0, 0.3), (1.0, 0.8), (0.0, 0.1)):
        mat = sc.Material((0.9, 0.9, 0.9, 1.0), metallic, rough, (le, le, le))
        d = sc.SceneDesc([mat], _box(pbr, [0] * 6), [sc.InstanceDesc(k) for k in range(6)], sc.CameraDesc((0.1, -0.2, 0.3), (0.4, 0.1, -1.0), 1.2, 1.0))
        m = float(ora.Oracle().load_scene(d).render(32, 32, 64, seed=2, max_bounces=6)[..., :3].mean())
        assert le * 1.5 < m < le * 7.0 * 1.01, (metallic, rough, m)          # 7 = sum_{k=0..6} 1^k


def test_direct_light_from_a_rectangular_emitter_matches_quadrature(ora, pbr):
    """One-bounce radiance on a Lambert floor under a square emitter: L = rho/pi * integral over the emitter of
    Le cos(theta_floor) cos(theta_light) / r^2 dA (one-sided emitter facing the floor), evaluated here by brute-force
    quadrature in float64.  Checks area-light sampling pdfs, the geometry term and MIS against BSDF hits."""
    sc = pbr.scene
    q = pbr.scenes._quad
    rho, le, hgt, e = 0.7, 10.0, 1.5, 0.5
    floor = q((-50, 0, 50), (50, 0, 50), (50, 0, -50), (-50, 0, -50))               # y = 0, normal +y
    light = q((-e, hgt, -e), (e, hgt, -e), (e, hgt, e), (-e, hgt, e))               # y = hgt, normal -y (faces the floor)
    mats = [sc.Material((rho, rho, rho, 1.0), 0.0, 1.0), sc.Material((0, 0, 0, 1.0), 0.0, 1.0, (le, le, le))]
    cam = sc.CameraDesc((0.0, 6.0, 0.0), (0.0, 0.0, 0.0001), 0.35, 1.0)             # looking straight down at the origin
    d = sc.SceneDesc(mats, [sc.MeshDesc(*floor, 0), sc.MeshDesc(*light, 1)], [sc.InstanceDesc(0), sc.InstanceDesc(1)], cam)
    o = ora.Oracle().load_scene(d)
    img = o.render(64, 64, 256, seed=3, max_bounces=1)[..., :3]

    def radiance_at(x, z, n=400):
        u = (np.arange(n) + 0.5) / n * 2 * e - e
        X, Z = np.meshgrid(u, u)
        dx, dz = X - x, Z - z
        r2 = dx * dx + dz * dz + hgt * hgt
        return rho / math.pi * le * float((hgt * hgt / (r2 * r2)).sum()) * (2 * e / n) ** 2

    # the camera sees |x|,|z| <= 6 tan(0.175) ~ 1.06 on the floor; the emitter hides the centre (it is one-sided: black from above)
    half = 6.0 * math.tan(0.175)
    for (px, py) in ((4, 4), (60, 8), (10, 56), (58, 58), (2, 32)):
        blk = img[py - 2 : py + 3, px - 2 : px + 3].mean((0, 1))
        # pixel centre → floor point; the image is x-mirrored (lookAtRH with up = -y) but the scene is symmetric in x and z
        fx, fz = ((px + 0.5) / 64 * 2 - 1) * half, ((py + 0.5) / 64 * 2 - 1) * half
        want = radiance_at(abs(fx), abs(fz))
        assert np.allclose(blk, want, rtol=0.03), ((px, py), blk, want)
    centre = img[28:36, 28:36]
    assert float(centre.max()) == 0.0                                                # back of the one-sided emitter


def test_more_bounces_than_russian_roulette_start_stay_unbiased(ora, pbr):
    """Russian roulette (from bounce 3) must not change the mean: the furnace value with rho = 0.9 at B = 20 is within 1 %
    of the series although most paths are terminated early."""
    sc = pbr.scene
    rho, le, B = 0.9, 1.0, 20
    mat = sc.Material((rho, rho, rho, 1.0), 0.0, 1.0, (le, le, le))
    d = sc.SceneDesc([mat], _box(pbr, [0] * 6), [sc.InstanceDesc(k) for k in range(6)], sc.CameraDesc((0.0, 0.0, 0.0), (0.3, 0.2, -1.0), 1.0, 1.0))
    o = ora.Oracle().load_scene(d)
    img = o.render(48, 48, 128, seed=11, max_bounces=B)[..., :3]
    want = le * (1 - rho ** (B + 1)) / (1 - rho)
    assert abs(float(img.mean()) / want - 1.0) < 0.01
    st = o.stats()
    assert st["segments"] / st["paths"] < B * 0.8          # roulette really did end paths early


def _ggx_albedo(base, metallic, roughness, mu, n_th=1500, n_ph=720):
    """Directional albedo  ∫ f(wo, wi) cos(theta_i) dwi  of the specified BSDF (Lambert (1-m)·base/pi + F·D·G1(v)·G1(l) /
    (4 nov nol), D GGX, separable Smith, Schlick) by midpoint quadrature in float64."""
    a2 = max(roughness * roughness, 1e-3) ** 2
    cd = np.asarray(base, np.float64) * (1.0 - metallic)
    f0 = 0.04 * (1.0 - metallic) + np.asarray(base, np.float64) * metallic
    wo = np.array([math.sqrt(1 - mu * mu), 0.0, mu])
    ct = (np.arange(n_th) + 0.5) / n_th                       # cos(theta_i), uniform in cos → dw = dcos dphi
    ph = (np.arange(n_ph) + 0.5) / n_ph * 2 * math.pi
    CT, PH = np.meshgrid(ct, ph, indexing="ij")
    ST = np.sqrt(1 - CT * CT)
    wi = np.stack([ST * np.cos(PH), ST * np.sin(PH), CT], -1)
    h = wi + wo
    h /= np.linalg.norm(h, axis=-1, keepdims=True)
    noh, voh = h[..., 2], (h * wo).sum(-1)
    D = a2 / (math.pi * (noh * noh * (a2 - 1) + 1) ** 2)
    g1 = lambda x: 2 * x / (x + np.sqrt((1 - a2) * x * x + a2))
    sp = D * g1(mu) * g1(CT) / (4 * mu * CT)
    m5 = np.clip(1 - voh, 0, None) ** 5
    dw = (1.0 / n_th) * (2 * math.pi / n_ph)
    out = []
    for c in range(3):
        F = f0[c] + (1 - f0[c]) * m5
        out.append(float(((cd[c] / math.pi + F * sp) * CT).sum() * dw))
    return np.array(out)


@pytest.mark.parametrize("metallic,roughness,base", [(1.0, 0.5, (0.9, 0.6, 0.2)), (0.0, 0.5, (0.8, 0.8, 0.8)), (1.0, 0.25, (1.0, 1.0, 1.0)), (0.0, 0.3, (0.2, 0.5, 0.9)),
                                                     (0.5, 0.7, (0.7, 0.7, 0.3))])
@pytest.mark.parametrize("height", [3.0, 0.6])
def test_ggx_plane_under_a_white_sky_equals_the_brdf_albedo(ora, pbr, metallic, roughness, base, height):
    """A convex GGX surface under radiance 1 from every direction shows its directional albedo.  The estimator mixes VNDF
    sampling, cosine sampling and environment NEE by MIS; all of it has to agree with the quadrature of the BSDF itself."""
    sc = pbr.scene
    env = np.ones((16, 32, 3), np.float32)
    v, i = pbr.scenes._quad((-200, 0, 200), (200, 0, 200), (200, 0, -200), (-200, 0, -200))      # normal +y
    cam = sc.CameraDesc((0.0, height, 2.0), (0.0, 0.0, 0.0), 0.05, 1.0)                           # narrow view: one angle of incidence
    d = sc.SceneDesc([sc.Material((*base, 1.0), metallic, roughness)], [sc.MeshDesc(v, i, 0)], [sc.InstanceDesc(0)], cam, env=env)
    img = ora.Oracle().load_scene(d).render(24, 24, 512, seed=7, max_bounces=1)[..., :3]
    mu = height / math.hypot(height, 2.0)
    want = _ggx_albedo(base, metallic, roughness, mu)
    got = img.mean((0, 1))
    assert np.allclose(got, want, rtol=0.02, atol=2e-3), (got, want)


def test_ray_generation_agrees_with_the_reference_projection(ora, pbr):
    """P1 against R5: a small emitter at a known world position must light the pixel that the reference's own matrices
    (glm::lookAtRH with up = (0,-1,0), perspectiveRH_NO, un-flipped Vulkan viewport: CameraData.hpp:22-32,
    PbrRenderSystem.cpp:425-430) send it to:  pixel = ((ndc.x + 1)/2 · W, (ndc.y + 1)/2 · H)."""
    sc = pbr.scene
    q = pbr.scenes._quad
    W, H = 96, 64
    cam = sc.CameraDesc((0.5, -0.3, 4.0), (0.2, 0.1, 0.0), 0.9, W / H)
    cd = sc.make_camera_data(cam.position, cam.target, cam.fov_y, cam.aspect)
    view, proj = cd.view.astype(np.float64).T, cd.proj.astype(np.float64).T          # stored column-major (glm): transpose → row-major math
    for world in ((1.0, 0.5, 0.0), (-1.2, -0.8, 0.5), (0.3, 0.9, -1.0), (-0.6, 0.2, 1.5)):
        e = 0.03
        x, y, z = world
        light = q((x - e, y - e, z), (x + e, y - e, z), (x + e, y + e, z), (x - e, y + e, z))      # faces +z, towards the camera
        d = sc.SceneDesc([sc.Material((0, 0, 0, 1), 0.0, 1.0, (50.0, 50.0, 50.0))], [sc.MeshDesc(*light, 0)], [sc.InstanceDesc(0)], cam)
        img = ora.Oracle().load_scene(d).render(W, H, 16, seed=1, max_bounces=0)[..., 0]
        clip = proj @ view @ np.array([x, y, z, 1.0])
        ndc = clip[:3] / clip[3]
        px, py = (ndc[0] + 1) / 2 * W, (ndc[1] + 1) / 2 * H
        ys, xs = np.nonzero(img > 0)
        assert len(xs) > 0
        cx, cy = (xs + 0.5).mean(), (ys + 0.5).mean()
        assert abs(cx - px) < 0.75 and abs(cy - py) < 0.75, (world, (cx, cy), (px, py))


def test_light_transport_is_linear_in_the_emitters(ora, pbr):
    """Rendering with the environment only, plus rendering with the emissive triangles only, equals rendering with both
    (in the mean): pins the environment-vs-area selection probability (RNG dim 7, p = 1/2) and both MIS pairs together."""
    import copy

    both = pbr.scenes.by_name("textured_atrium", scale=0.05, tex_size=64, env_size=(64, 32))
    assert both.env is not None and any(max(m.emissive) > 0 for m in both.materials)
    env_only = copy.deepcopy(both)
    for m in env_only.materials:
        m.emissive = (0.0, 0.0, 0.0)
    area_only = copy.deepcopy(both)
    area_only.env = None
    r = {}
    for name, d in (("both", both), ("env", env_only), ("area", area_only)):
        r[name] = ora.Oracle().load_scene(d).render(40, 24, 256, seed=21, max_bounces=4)[..., :3].astype(np.float64)
    total, parts = r["both"].mean((0, 1)), (r["env"] + r["area"]).mean((0, 1))
    assert r["env"].mean() > 0 and r["area"].mean() > 0
    assert np.allclose(total, parts, rtol=0.03), (total, parts)
