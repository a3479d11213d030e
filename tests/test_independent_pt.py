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
    return _independent(pbr.scenes.cornell_box(), w, h, spp, max_bounces, seed)


def _agree(ind, var, ref, ovar, w, h, block_tol=0.04):
    """True when the two estimates agree: whole image to 1.5 % + 4 standard errors, 4x4 blocks per channel to block_tol + 4 standard errors."""
    se = math.sqrt(var.sum() + ovar.sum()) / var.size
    if abs(ind.mean() - ref.mean()) > 4 * se + 0.015 * ref.mean():
        return False, se
    B = 4
    for c in range(3):
        a = ind[..., c].reshape(h // B, B, w // B, B).mean((1, 3))
        b = ref[..., c].reshape(h // B, B, w // B, B).mean((1, 3))
        s = np.sqrt((var[..., c] + ovar[..., c]).reshape(h // B, B, w // B, B).sum((1, 3))) / (B * B)
        if (np.abs(a - b) > 4 * s + block_tol * np.maximum(b, 0.02)).any():
            return False, se
    return True, se


def _model_matrix(inst):
    """The instance's model matrix from ITS definition (ModelPushConstant.hpp:40-46): translate(t) * toMat4(q) * scale(s), q = (w, x, y, z), the rotation matrix
    of a unit quaternion from the textbook formula — float64, row-major 4x4 acting on column vectors.  (An instance given by a matrix: glm's column-major 16 floats.)"""
    if getattr(inst, "matrix", None) is not None:
        return np.asarray(inst.matrix, np.float64).reshape(4, 4).T
    w_, x, y, z = (float(c) for c in inst.q_wxyz)
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w_), 2 * (x * z + y * w_)],
                  [2 * (x * y + z * w_), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w_)],
                  [2 * (x * z - y * w_), 2 * (y * z + x * w_), 1 - 2 * (x * x + y * y)]])
    M = np.eye(4)
    M[:3, :3] = R @ np.diag([float(c) for c in inst.s])
    M[:3, 3] = [float(c) for c in inst.t]
    return M


def _independent(d, w, h, spp, max_bounces, seed, mutation=None):
    """mutation: None, or one of the mistakes the transformed-scene test must catch — "normal_matrix_is_model" (normals transformed by the model matrix instead of
    its inverse transpose), "no_renormalisation" (the transformed normal / tangent / bitangent left un-normalised)."""
    tris, alb, emi, met, rough, vattr, texid = [], [], [], [], [], [], []
    for inst in d.instances:
        m = d.meshes[inst.mesh]
        M = _model_matrix(inst)                                         # vertex.glsl:26: worldPos = model * vec4(position, 1)
        NM = np.linalg.inv(M[:3, :3]).T                                 # ModelPushConstant.hpp:36: normalModel = mat3(transpose(inverse(model)))
        if mutation == "normal_matrix_is_model":
            NM = M[:3, :3]
        P = np.asarray(m.vertices["position"], np.float64) @ M[:3, :3].T + M[:3, 3]
        I = np.asarray(m.indices, np.int64).reshape(-1, 3)
        mat = d.materials[m.material]
        N0, T0, UV = (np.asarray(m.vertices[k], np.float64) for k in ("normal", "tangent", "texCoords"))
        unit = (lambda v: v) if mutation == "no_renormalisation" else (lambda v: v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-30))
        Nv = unit(N0 @ NM.T)                                            # vertex.glsl:33-35: each of N, T, B = normalize(normalModel * ...),
        Tv = np.concatenate([unit(T0[:, :3] @ NM.T), T0[:, 3:4]], 1)    # B from the OBJECT-space cross(N, T.xyz) * T.w
        Bv = unit((np.cross(N0, T0[:, :3]) * T0[:, 3:4]) @ NM.T)
        for a, b, c in I:
            tris.append((P[a], P[b], P[c])); alb.append(mat.base_color[:3]); emi.append(mat.emissive); met.append(mat.metallic); rough.append(mat.roughness)
            vattr.append([np.concatenate([UV[i], Nv[i], Tv[i, :3], Bv[i]]) for i in (a, b, c)])
            texid.append((mat.tex_color, mat.tex_normal, mat.tex_mr))
    vattr, texid = np.asarray(vattr, np.float64), np.asarray(texid, np.int64)      # (tri, 3 vertices, 11), (tri, 3)
    textures = [np.asarray(t, np.float64) / 255.0 for t in getattr(d, "textures", [])]
    A = np.array([t[0] for t in tris]); E1 = np.array([t[1] - t[0] for t in tris]); E2 = np.array([t[2] - t[0] for t in tris])
    NG = np.cross(E1, E2); NG /= np.linalg.norm(NG, axis=1, keepdims=True)
    alb, emi, met, rough = np.asarray(alb, np.float64), np.asarray(emi, np.float64), np.asarray(met, np.float64), np.asarray(rough, np.float64)
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
        x = (2 * (px + rng.random(n)) / w - 1) * th * cam.aspect
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
            env = getattr(d, "env", None)
            if env is not None:                                          # a ray that leaves the scene sees the lat-long environment: row 0 = +y,
                miss = alive & ~np.isfinite(th_)                         # u = atan2(d.z, d.x) / 2 pi + 1/2, v = acos(d.y) / pi, piecewise-constant texels
                eh, ew = env.shape[:2]
                uu_ = np.arctan2(dirs[:, 2], dirs[:, 0]) / (2 * math.pi) + 0.5
                vv_ = np.arccos(np.clip(dirs[:, 1], -1.0, 1.0)) / math.pi
                ex = np.clip((uu_ * ew).astype(np.int64), 0, ew - 1); ey = np.clip((vv_ * eh).astype(np.int64), 0, eh - 1)
                L += np.where(miss[:, None], T * np.asarray(env, np.float64)[ey, ex], 0.0)
            alive = hit
            if b == max_bounces:
                break
            nn = np.where(front[:, None], ng, -ng)
            with np.errstate(invalid="ignore"):                          # rays that left the box: t = inf, masked by `alive`
                P = org + dirs * np.where(hit, th_, 0.0)[:, None] + nn * 1e-7
            # the surface the BRDF sees: vertex attributes interpolated with the hit's barycentrics, textures looked up NEAREST / REPEAT (the reference's
            # default sampler), glTF channels (roughness = G, metallic = B), normal map through the interpolated tangent frame (fragment.glsl:19-31)
            base_k, met_k, rough_k, ns = alb[k].copy(), met[k].copy(), rough[k].copy(), nn.copy()
            if (texid >= 0).any():
                bu, bv_ = uu[np.arange(n), k], vv[np.arange(n), k]
                bw = 1.0 - bu - bv_
                at = vattr[k]
                ip = at[:, 0] * bw[:, None] + at[:, 1] * bu[:, None] + at[:, 2] * bv_[:, None]
                uv, Ni, Ti, Bi = ip[:, 0:2], ip[:, 2:5], ip[:, 5:8], ip[:, 8:11]
                fu = uv - np.floor(uv)
                def tap(ids):
                    out = np.ones((n, 4))
                    for t_id in np.unique(ids[ids >= 0]):
                        tx = textures[t_id]; th_px, tw_px = tx.shape[:2]
                        sel = ids == t_id
                        x = np.minimum((fu[sel, 0] * tw_px).astype(np.int64), tw_px - 1); y = np.minimum((fu[sel, 1] * th_px).astype(np.int64), th_px - 1)
                        out[sel] = tx[y, x]
                    return out
                tc, tn_, tm = texid[k, 0], texid[k, 1], texid[k, 2]
                base_k = np.where((tc >= 0)[:, None], base_k * tap(tc)[:, :3], base_k)
                mr = tap(tm)
                rough_k = np.where(tm >= 0, rough_k * mr[:, 1], rough_k); met_k = np.where(tm >= 0, met_k * mr[:, 2], met_k)
                nm = 2.0 * tap(tn_)[:, :3] - 1.0
                nmap = Ti * nm[:, 0:1] + Bi * nm[:, 1:2] + Ni * nm[:, 2:3]
                nmap /= np.maximum(np.linalg.norm(nmap, axis=1, keepdims=True), 1e-30)
                ns = np.where((tn_ >= 0)[:, None], nmap, Ni / np.maximum(np.linalg.norm(Ni, axis=1, keepdims=True), 1e-30))
                ns = np.where(((ns * ng).sum(1) < 0)[:, None], -ns, ns)           # the shading normal on the geometric normal's side,
                ns = np.where(front[:, None], ns, -ns)                            # both towards the viewer,
                ns = np.where(((ns * -dirs).sum(1) > 0)[:, None], ns, nn)         # and the geometric normal where the viewer is below the shading normal
            # uniform hemisphere about nn: pdf 1/(2 pi); f = albedo/pi; weight = albedo * cos * 2
            z, phi = rng.random(n), 2 * math.pi * rng.random(n)
            r = np.sqrt(np.maximum(0.0, 1 - z * z))
            a_ = np.where(np.abs(nn[:, 0:1]) > 0.9, [[0.0, 1.0, 0.0]], [[1.0, 0.0, 0.0]])
            t1 = np.cross(a_, nn); t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
            t2 = np.cross(nn, t1)
            wi = t1 * (r * np.cos(phi))[:, None] + t2 * (r * np.sin(phi))[:, None] + nn * z[:, None]
            # BRDF from its definition: Lambert base(1-m)/pi + Cook-Torrance D G F / (4 n.v n.l), GGX D, separable Smith G1,
            # Schlick F, alpha = max(r^2, 1e-3), F0 = 0.04 (1-m) + base m; materials with metallic 0 and roughness >= 1 are Lambert only
            m_, r_ = met_k[:, None], rough_k
            cl = (ns * wi).sum(1)                                          # the BRDF's cosine is the shading normal's; below it there is no contribution
            cd = base_k * (1 - m_)
            fr = cd / math.pi
            ggx = ~((met_k == 0) & (r_ >= 1))
            if ggx.any():
                wo = -dirs
                nv = np.maximum((ns * wo).sum(1), 1e-4)
                hv = wo + wi
                hv /= np.maximum(np.linalg.norm(hv, axis=1, keepdims=True), 1e-30)
                nh, vh = (ns * hv).sum(1), (wo * hv).sum(1)
                al = np.maximum(r_ * r_, 1e-3); a2 = al * al
                D = a2 / (math.pi * (nh * nh * (a2 - 1) + 1) ** 2)
                g1 = lambda x: 2 * x / (x + np.sqrt(a2 + (1 - a2) * x * x))
                F0 = 0.04 * (1 - m_) + base_k * m_
                F = F0 + (1 - F0) * (np.maximum(1 - vh, 0.0) ** 5)[:, None]
                cz = np.maximum(cl, 1e-9)
                spec = (D * g1(nv) * g1(cz) / (4 * nv * cz))[:, None] * F
                fr = fr + np.where(ggx[:, None], spec, 0.0)
            T = T * fr * (2 * math.pi * np.maximum(cl, 0.0))[:, None]     # f cos / pdf, pdf = 1 / (2 pi) over the geometric hemisphere
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


def test_an_independent_estimator_agrees_on_a_ggx_surface(ora, pbr):
    """The same cross-check where the oracle's path is at its most intricate: a rough metal-dielectric GGX floor (VNDF sampling, lobe selection, Fresnel, MIS
    between BSDF sampling and light sampling) under an emissive quad, seen at a grazing-ish angle.  The independent estimator evaluates the BRDF from its
    definition and samples the hemisphere uniformly."""
    sc = pbr.scene
    quad = pbr.scenes._quad
    mats = [sc.Material((0.8, 0.6, 0.4, 1.0), 0.3, 0.5), sc.Material((0.0, 0.0, 0.0, 1.0), 0.0, 1.0, (6.0, 6.0, 6.0)), sc.Material((0.5, 0.5, 0.7, 1.0), 0.0, 1.0)]
    q = [(quad((-2, 0, 2), (2, 0, 2), (2, 0, -2), (-2, 0, -2)), 0),                     # floor, normal +y: GGX
         (quad((-0.8, 1.5, -0.8), (0.8, 1.5, -0.8), (0.8, 1.5, 0.8), (-0.8, 1.5, 0.8)), 1),   # light, normal -y
         (quad((-2, 0, -2), (2, 0, -2), (2, 2.5, -2), (-2, 2.5, -2)), 2)]               # back wall, normal +z: Lambert
    meshes = [sc.MeshDesc(v, i, m) for (v, i), m in q]
    inst = [sc.InstanceDesc(k, (0.0, 0.0, 0.0), (1.0, 0.0, 0.0, 0.0), (1.0, 1.0, 1.0)) for k in range(3)]
    d = sc.SceneDesc(mats, meshes, inst, sc.CameraDesc((0.0, 0.9, 3.2), (0.0, 0.2, 0.0), math.radians(50.0), 1.0), "ggx_floor")
    w = h = 16
    mb = 3
    ind, var = _independent(d, w, h, 3000, mb, seed=4052)
    o = ora.Oracle().load_scene(d)
    r1 = o.render(w, h, 512, seed=5, max_bounces=mb)[..., :3].astype(np.float64)
    r2 = o.render(w, h, 512, seed=6, max_bounces=mb)[..., :3].astype(np.float64)
    ref, ovar = 0.5 * (r1 + r2), ((r1 - r2) ** 2) / 4
    se = math.sqrt(var.sum() + ovar.sum()) / var.size
    assert ref.mean() > 0.05
    assert abs(ind.mean() - ref.mean()) <= 4 * se + 0.015 * ref.mean(), (ind.mean(), ref.mean(), se)
    B = 4
    for c in range(3):
        a = ind[..., c].reshape(h // B, B, w // B, B).mean((1, 3))
        b = ref[..., c].reshape(h // B, B, w // B, B).mean((1, 3))
        s = np.sqrt((var[..., c] + ovar[..., c]).reshape(h // B, B, w // B, B).sum((1, 3))) / (B * B)
        bad = np.abs(a - b) > 4 * s + 0.04 * np.maximum(b, 0.02)
        assert not bad.any(), (c, a[bad], b[bad], s[bad])


def test_an_independent_estimator_agrees_under_an_environment_light(ora, pbr):
    """... and with the lat-long environment as the only light: the oracle importance-samples it (row / column cdfs, MIS against BSDF sampling), the independent
    estimator only looks it up, from the mapping's definition, where a path leaves the scene."""
    sc = pbr.scene
    quad = pbr.scenes._quad
    mats = [sc.Material((0.7, 0.7, 0.6, 1.0), 0.0, 0.6), sc.Material((0.6, 0.3, 0.3, 1.0), 0.0, 1.0)]
    q = [(quad((-2, 0, 2), (2, 0, 2), (2, 0, -2), (-2, 0, -2)), 0), (quad((-2, 0, -1), (0.5, 0, -1), (0.5, 1.6, -1), (-2, 1.6, -1)), 1)]
    meshes = [sc.MeshDesc(v, i, m) for (v, i), m in q]
    inst = [sc.InstanceDesc(k, (0.0, 0.0, 0.0), (1.0, 0.0, 0.0, 0.0), (1.0, 1.0, 1.0)) for k in range(2)]
    d = sc.SceneDesc(mats, meshes, inst, sc.CameraDesc((0.3, 1.0, 3.0), (0.0, 0.3, 0.0), math.radians(50.0), 1.0), "env_floor")
    eh, ew = 16, 32
    yy, xx = np.mgrid[0:eh, 0:ew]
    env = np.zeros((eh, ew, 3), np.float32)                              # smooth and asymmetric: brighter up and towards +x / +z, tinted
    env[..., 0] = 0.3 + 1.2 * np.cos(math.pi * (yy + 0.5) / eh / 2) ** 2 + 0.5 * (xx / ew)
    env[..., 1] = 0.3 + 1.0 * np.cos(math.pi * (yy + 0.5) / eh / 2) ** 2
    env[..., 2] = 0.4 + 0.8 * np.cos(math.pi * (yy + 0.5) / eh / 2) ** 2 + 0.6 * (1 - xx / ew)
    d.env = env
    w = h = 16
    mb = 3
    ind, var = _independent(d, w, h, 2000, mb, seed=99)
    o = ora.Oracle().load_scene(d)
    r1 = o.render(w, h, 512, seed=15, max_bounces=mb)[..., :3].astype(np.float64)
    r2 = o.render(w, h, 512, seed=16, max_bounces=mb)[..., :3].astype(np.float64)
    ref, ovar = 0.5 * (r1 + r2), ((r1 - r2) ** 2) / 4
    se = math.sqrt(var.sum() + ovar.sum()) / var.size
    assert ref.mean() > 0.05
    assert abs(ind.mean() - ref.mean()) <= 4 * se + 0.015 * ref.mean(), (ind.mean(), ref.mean(), se)
    B = 4
    for c in range(3):
        a = ind[..., c].reshape(h // B, B, w // B, B).mean((1, 3))
        b = ref[..., c].reshape(h // B, B, w // B, B).mean((1, 3))
        s = np.sqrt((var[..., c] + ovar[..., c]).reshape(h // B, B, w // B, B).sum((1, 3))) / (B * B)
        bad = np.abs(a - b) > 4 * s + 0.04 * np.maximum(b, 0.02)
        assert not bad.any(), (c, a[bad], b[bad], s[bad])


def test_an_independent_estimator_agrees_on_a_textured_normal_mapped_surface(ora, pbr):
    """... and with R7's textures in play: base colour x texel, the glTF metal-rough channels (G = roughness, B = metallic), a tangent-space normal map
    through the per-vertex T / B / N frame, NEAREST + REPEAT lookups at uv beyond [0, 1) — each taken from its definition in the reference's
    fragment shader (geometry_pass/fragment.glsl:19-31, vertex.glsl:25-36), not from the oracle.  A swapped channel, a flipped bitangent, a
    transposed texture or a wrong wrap changes the image far beyond Monte-Carlo error."""
    sc = pbr.scene
    quad = pbr.scenes._quad
    rng = np.random.default_rng(21)
    T_ = 8
    albedo = np.concatenate([rng.integers(60, 256, (T_, T_, 3)), np.full((T_, T_, 1), 255)], 2).astype(np.uint8)
    albedo[:, : T_ // 2, 0] //= 3                                                    # left half less red: a transposed or mirrored lookup shows
    mr = np.zeros((T_, T_, 4), np.uint8); mr[..., 3] = 255
    mr[..., 1] = rng.integers(90, 256, (T_, T_)); mr[..., 2] = rng.integers(0, 2, (T_, T_)) * 200        # G roughness, B metallic
    yy, xx = np.mgrid[0:T_, 0:T_]
    nx, ny = 0.45 * np.sin(2 * math.pi * xx / T_), 0.35 * np.cos(2 * math.pi * yy / T_) + 0.2           # asymmetric in y: the bitangent's sign matters
    nz = np.sqrt(1 - nx * nx - ny * ny)
    nmap = np.stack([np.round((nx * 0.5 + 0.5) * 255), np.round((ny * 0.5 + 0.5) * 255), np.round((nz * 0.5 + 0.5) * 255), np.full((T_, T_), 255)], 2).astype(np.uint8)
    mats = [sc.Material((0.9, 0.8, 0.7, 1.0), 1.0, 0.8, (0.0, 0.0, 0.0), 0, 1, 2), sc.Material((0.0, 0.0, 0.0, 1.0), 0.0, 1.0, (7.0, 6.0, 5.0)),
            sc.Material((0.5, 0.6, 0.7, 1.0), 0.0, 1.0)]
    fv, fi = quad((-2, 0, 2), (2, 0, 2), (2, 0, -2), (-2, 0, -2))
    fv["texCoords"] = np.array([(-0.3, -0.2), (1.9, -0.2), (1.9, 1.6), (-0.3, 1.6)], np.float32)       # beyond [0, 1): REPEAT
    q = [((fv, fi), 0), (quad((-0.9, 1.6, -0.9), (0.9, 1.6, -0.9), (0.9, 1.6, 0.9), (-0.9, 1.6, 0.9)), 1), (quad((-2, 0, -2), (2, 0, -2), (2, 2.5, -2), (-2, 2.5, -2)), 2)]
    meshes = [sc.MeshDesc(v, i, m) for (v, i), m in q]
    inst = [sc.InstanceDesc(k, (0.0, 0.0, 0.0), (1.0, 0.0, 0.0, 0.0), (1.0, 1.0, 1.0)) for k in range(3)]
    d = sc.SceneDesc(mats, meshes, inst, sc.CameraDesc((0.4, 1.0, 3.2), (0.0, 0.2, 0.0), math.radians(50.0), 1.0), "textured_floor", textures=[albedo, nmap, mr])
    w = h = 16
    mb = 3
    ind, var = _independent(d, w, h, 3000, mb, seed=777)
    o = ora.Oracle().load_scene(d)
    r1 = o.render(w, h, 512, seed=25, max_bounces=mb)[..., :3].astype(np.float64)
    r2 = o.render(w, h, 512, seed=26, max_bounces=mb)[..., :3].astype(np.float64)
    ref, ovar = 0.5 * (r1 + r2), ((r1 - r2) ** 2) / 4
    se = math.sqrt(var.sum() + ovar.sum()) / var.size
    assert ref.mean() > 0.05
    assert abs(ind.mean() - ref.mean()) <= 4 * se + 0.015 * ref.mean(), (ind.mean(), ref.mean(), se)
    B = 4
    for c in range(3):
        a = ind[..., c].reshape(h // B, B, w // B, B).mean((1, 3))
        b = ref[..., c].reshape(h // B, B, w // B, B).mean((1, 3))
        s = np.sqrt((var[..., c] + ovar[..., c]).reshape(h // B, B, w // B, B).sum((1, 3))) / (B * B)
        bad = np.abs(a - b) > 4 * s + 0.04 * np.maximum(b, 0.02)
        assert not bad.any(), (c, a[bad], b[bad], s[bad])
    print("textured: independent %.4f oracle %.4f (se %.4f)" % (ind.mean(), ref.mean(), se))


def test_an_independent_estimator_agrees_on_a_rotated_non_uniformly_scaled_normal_mapped_instance(ora, pbr):
    """R3 / R4 by evidence the oracle did not write: the textured, normal-mapped floor of the previous test is now an INSTANCE with a rotation and a non-uniform
    scale (and the light a rotated, scaled instance too), so that world normals, tangents and bitangents only come out right through
    normalModel = transpose(inverse(model)) and the re-normalisation of vertex.glsl:33-35 — the independent estimator forms both from their definitions
    (numpy's inverse, the textbook quaternion matrix).  The same scene after ptc_update_instance + a refit (oracle: ora_scene_refit) must agree as well.
    And the two classic mistakes — normals through the model matrix itself, no re-normalisation — must NOT agree."""
    import copy
    sc = pbr.scene
    quad = pbr.scenes._quad
    rng = np.random.default_rng(21)
    T_ = 8
    albedo = np.concatenate([rng.integers(60, 256, (T_, T_, 3)), np.full((T_, T_, 1), 255)], 2).astype(np.uint8)
    albedo[:, : T_ // 2, 0] //= 3
    mr = np.zeros((T_, T_, 4), np.uint8); mr[..., 3] = 255
    mr[..., 1] = rng.integers(90, 256, (T_, T_)); mr[..., 2] = rng.integers(0, 2, (T_, T_)) * 200
    yy, xx = np.mgrid[0:T_, 0:T_]
    nx, ny = 0.55 * np.sin(2 * math.pi * xx / T_), 0.45 * np.cos(2 * math.pi * yy / T_) + 0.2        # a strong map: the frame's errors show
    nz = np.sqrt(1 - nx * nx - ny * ny)
    nmap = np.stack([np.round((nx * 0.5 + 0.5) * 255), np.round((ny * 0.5 + 0.5) * 255), np.round((nz * 0.5 + 0.5) * 255), np.full((T_, T_), 255)], 2).astype(np.uint8)
    mats = [sc.Material((0.9, 0.8, 0.7, 1.0), 1.0, 0.8, (0.0, 0.0, 0.0), 0, 1, 2), sc.Material((0.0, 0.0, 0.0, 1.0), 0.0, 1.0, (9.0, 8.0, 7.0)),
            sc.Material((0.5, 0.6, 0.7, 1.0), 0.0, 1.0)]
    fv, fi = quad((-1, -0.7, 1), (1, 0.7, 1), (1, 0.7, -1), (-1, -0.7, -1))             # a floor that is TILTED in object space (normal ~ (-0.57, 0.82, 0)) ...
    fv["texCoords"] = np.array([(-0.3, -0.2), (1.9, -0.2), (1.9, 1.6), (-0.3, 1.6)], np.float32)
    meshes = [sc.MeshDesc(fv, fi, 0), sc.MeshDesc(*quad((-1, 0, -1), (1, 0, -1), (1, 0, 1), (-1, 0, 1)), 1), sc.MeshDesc(*quad((-2, 0, -2), (2, 0, -2), (2, 2.5, -2), (-2, 2.5, -2)), 2)]
    a1, a2 = -0.263, -0.35                                                             # ... stretched 2.6 x 1 x 1.7 and turned back to about level: its normal is not an axis of the
                                                                                        # scale, so model * n (45 degrees off) and transpose(inverse(model)) * n part ways
    inst = [sc.InstanceDesc(0, (0.1, 0.15, 0.0), (math.cos(a1 / 2), 0.0, 0.0, math.sin(a1 / 2)), (2.6, 1.0, 1.7)),
            sc.InstanceDesc(1, (0.0, 1.9, 0.0), (math.cos(a2 / 2), math.sin(a2 / 2), 0.0, 0.0), (0.9, 1.0, 0.6)),
            sc.InstanceDesc(2, (0.0, 0.0, 0.0), (1.0, 0.0, 0.0, 0.0), (1.0, 1.0, 1.0))]
    d = sc.SceneDesc(mats, meshes, inst, sc.CameraDesc((0.4, 1.3, 3.4), (0.0, 0.3, 0.0), math.radians(50.0), 1.0), "tilted_textured_floor", textures=[albedo, nmap, mr])
    w = h = 16
    mb = 3

    def oracle_estimate(o):
        r1 = o.render(w, h, 512, seed=35, max_bounces=mb)[..., :3].astype(np.float64)
        r2 = o.render(w, h, 512, seed=36, max_bounces=mb)[..., :3].astype(np.float64)
        return 0.5 * (r1 + r2), ((r1 - r2) ** 2) / 4

    o = ora.Oracle().load_scene(d)
    ref, ovar = oracle_estimate(o)
    assert ref.mean() > 0.05
    ind, var = _independent(d, w, h, 3000, mb, seed=31)
    ok, se = _agree(ind, var, ref, ovar, w, h)
    assert ok, (ind.mean(), ref.mean(), se)
    print("transformed: independent %.4f oracle %.4f (se %.4f)" % (ind.mean(), ref.mean(), se))
    for mutation in ("normal_matrix_is_model", "no_renormalisation"):
        bad, bvar = _independent(d, w, h, 3000, mb, seed=31, mutation=mutation)
        ok_bad, _ = _agree(bad, bvar, ref, ovar, w, h)
        assert not ok_bad, f"the mutation {mutation} passes: the test does not see the normal matrix ({bad.mean():.4f} vs {ref.mean():.4f})"
    # the same through the dynamics path: new transforms for the committed instances, then a refit (R3 / R4 run again inside it)
    d2 = copy.deepcopy(d)
    b1 = -0.4
    d2.instances[0].q_wxyz = (math.cos(b1 / 2), math.sin(b1 / 2) * 0.6, 0.0, math.sin(b1 / 2) * 0.8)
    d2.instances[0].s = (1.5, 1.0, 2.8)
    d2.instances[0].t = (-0.1, 0.2, 0.1)
    o.update_instance(0, d2.instances[0].t, d2.instances[0].q_wxyz, d2.instances[0].s)
    o.scene_refit()
    ref2, ovar2 = oracle_estimate(o)
    ind2, var2 = _independent(d2, w, h, 3000, mb, seed=32)
    ok2, se2 = _agree(ind2, var2, ref2, ovar2, w, h)
    assert ok2, (ind2.mean(), ref2.mean(), se2)
    assert not _agree(ind, var, ref2, ovar2, w, h)[0], "the refitted scene's image equals the unmoved scene's: the move was not applied"
    print("transformed + refit: independent %.4f oracle %.4f (se %.4f)" % (ind2.mean(), ref2.mean(), se2))

