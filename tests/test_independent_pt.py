"""An INDEPENDENT estimate of the same integral, to catch what a shared specification cannot: the oracle and the kernels are two writings of one
algorithm (BVH, MIS-weighted next-event estimation, cosine / VNDF sampling, Russian roulette, counter-based RNG), so a mistake in the algorithm itself —
a missing cosine, a wrong MIS normalisation, emission from the wrong side, a camera convention — would pass every bit-exact test.  Here the Cornell box
(BASELINE configs[0]) is rendered by a different estimator of the rendering equation with nothing in common but the scene description:
float64 numpy, brute force over the 12 triangles, UNIFORM hemisphere sampling, no next-event estimation, no MIS, no Russian roulette, numpy's PCG64, the
camera from glm's lookAtRH definition.  Both are unbiased for the same truncated path integral (emission gathered at path vertices 0..max_bounces), so
block means agree within Monte-Carlo error; the test fails beyond 4 standard errors + 1.5 %.  (The reference has no renderer to take this role: SURVEY §0.)"""
import math

import numpy as np
import pytest


def _independent_cornell(pbr, w, h, spp, max_bounces, seed):
    d = pbr.scenes.cornell_box()
    tris, alb, emi = [], [], []
    for inst in d.instances:                                            # identity transforms in this scene: positions are world positions
        m = d.meshes[inst.mesh]
        P = np.asarray(m.vertices["position"], np.float64)
        I = np.asarray(m.indices, np.int64).reshape(-1, 3)
        mat = d.materials[m.material]
        for a, b, c in I:
            tris.append((P[a], P[b], P[c])); alb.append(mat.base_color[:3]); emi.append(mat.emissive)
    A = np.array([t[0] for t in tris]); E1 = np.array([t[1] - t[0] for t in tris]); E2 = np.array([t[2] - t[0] for t in tris])
    NG = np.cross(E1, E2); NG /= np.linalg.norm(NG, axis=1, keepdims=True)
    alb, emi = np.asarray(alb, np.float64), np.asarray(emi, np.float64)
    cam = d.camera
    eye, tgt = np.asarray(cam.position, np.float64), np.asarray(cam.target, np.float64)
    f = (tgt - eye) / np.linalg.norm(tgt - eye)                         # glm::lookAtRH(eye, target, up = (0,-1,0)): CameraData.hpp:22-32
    s = np.cross(f, [0.0, -1.0, 0.0]); s /= np.linalg.norm(s)
    u = np.cross(s, f)
    th = math.tan(cam.fov_y / 2)
    rng = np.random.Generator(np.random.PCG64(seed))
    img = np.zeros((h, w, 3)); img2 = np.zeros((h, w, 3))
    py, px = np.mgrid[0:h, 0:w]
    px, py = px.reshape(-1).astype(np.float64), py.reshape(-1).astype(np.float64)
    n = px.size
    for _ in range(spp):
        x = (2 * (px + rng.random(n)) / w - 1) * th * 1.0               # aspect 1
        y = (2 * (py + rng.random(n)) / h - 1) * th
        dirs = x[:, None] * s + y[:, None] * u + f
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        org = np.repeat(eye[None], n, 0)
        T = np.ones((n, 3)); L = np.zeros((n, 3)); alive = np.ones(n, bool)
        for b in range(max_bounces + 1):
            # brute force: every ray against every triangle (two-sided), closest t > 1e-9
            pv = np.cross(dirs[:, None, :], E2[None])
            det = (E1[None] * pv).sum(2)
            with np.errstate(divide="ignore", invalid="ignore"):
                inv = 1.0 / det
                tv = org[:, None, :] - A[None]
                uu = (tv * pv).sum(2) * inv
                qv = np.cross(tv, E1[None])
                vv = (dirs[:, None, :] * qv).sum(2) * inv
                tt = (E2[None] * qv).sum(2) * inv
            ok = (det != 0) & (uu >= 0) & (vv >= 0) & (uu + vv <= 1) & (tt > 1e-9)
            tt = np.where(ok, tt, np.inf)
            k = tt.argmin(1)
            th_ = tt[np.arange(n), k]
            hit = alive & np.isfinite(th_)
            ng = NG[k]
            front = (ng * -dirs).sum(1) > 0
            L += np.where((hit & front)[:, None], T * emi[k], 0.0)       # one-sided emission
            alive = hit
            if b == max_bounces:
                break
            nn = np.where(front[:, None], ng, -ng)
            with np.errstate(invalid="ignore"):                          # rays that left the box: t = inf, masked by `alive`
                P = org + dirs * np.where(hit, th_, 0.0)[:, None] + nn * 1e-7
            # uniform hemisphere about nn: pdf 1/(2 pi); f = albedo/pi; weight = albedo * cos * 2
            z, phi = rng.random(n), 2 * math.pi * rng.random(n)
            r = np.sqrt(np.maximum(0.0, 1 - z * z))
            a_ = np.where(np.abs(nn[:, 0:1]) > 0.9, [[0.0, 1.0, 0.0]], [[1.0, 0.0, 0.0]])
            t1 = np.cross(a_, nn); t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
            t2 = np.cross(nn, t1)
            wi = t1 * (r * np.cos(phi))[:, None] + t2 * (r * np.sin(phi))[:, None] + nn * z[:, None]
            T = T * alb[k] * (2 * z)[:, None]
            org, dirs = P, wi
            alive &= T.max(1) > 0
            T = np.where(alive[:, None], T, 0.0)
        img += L.reshape(h, w, 3); img2 += (L * L).reshape(h, w, 3)
    mean = img / spp
    var = np.maximum(img2 / spp - mean * mean, 0.0) / spp              # variance of the per-pixel mean
    return mean, var


def test_an_independent_estimator_agrees_with_the_oracle_on_the_cornell_box(ora, pbr):
    w = h = 16
    mb = 4
    ind, var = _independent_cornell(pbr, w, h, 1500, mb, seed=2026)
    o = ora.Oracle().load_scene(pbr.scenes.cornell_box())
    ref = o.render(w, h, 512, seed=77, max_bounces=mb)[..., :3].astype(np.float64)
    ref2 = o.render(w, h, 512, seed=78, max_bounces=mb)[..., :3].astype(np.float64)
    ora_var = ((ref - ref2) ** 2) / 2 / 2                                # crude per-pixel variance of the mean of the two renders
    ref = 0.5 * (ref + ref2)
    assert ind.mean() > 0.05 and ref.mean() > 0.05
    # whole image: the two estimators' means agree to 1.5 % + 4 standard errors
    se = math.sqrt(var.sum() + ora_var.sum()) / var.size
    assert abs(ind.mean() - ref.mean()) <= 4 * se + 0.015 * ref.mean(), (ind.mean(), ref.mean(), se)
    # 4x4-pixel blocks, per colour channel
    B = 4
    for c in range(3):
        a = ind[..., c].reshape(h // B, B, w // B, B).mean((1, 3))
        b = ref[..., c].reshape(h // B, B, w // B, B).mean((1, 3))
        s = np.sqrt((var[..., c] + ora_var[..., c]).reshape(h // B, B, w // B, B).sum((1, 3))) / (B * B)
        bad = np.abs(a - b) > 4 * s + 0.03 * np.maximum(b, 0.02)
        assert not bad.any(), (c, a[bad], b[bad], s[bad])
    # the walls are where glm::lookAtRH(eye, target, up = (0,-1,0)) puts them: s = f x up = -x, so image x runs towards world -x and the GREEN wall
    # (world x = +1) is on the image's left, the red one on its right; image y runs down (u = s x f = -y): ceiling light in the upper rows
    for im in (ref, ind):
        assert im[:, :3, 1].mean() > 2 * im[:, :3, 0].mean() and im[:, -3:, 0].mean() > 2 * im[:, -3:, 1].mean()
        assert im[:4].mean() > im[-4:].mean() * 0.5
