"""ctypes loader for the CPU oracle (oracle/_build/libptc_oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product
package (physically-based-renderer_amd/pbr_amd), which is HIP-only.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libptc_oracle.so")


class OraStats(C.Structure):
    _fields_ = [
        ("paths", C.c_uint64), ("segments", C.c_uint64), ("shadow_rays", C.c_uint64), ("hits", C.c_uint64),
        ("node_visits_closest", C.c_uint64), ("tri_tests_closest", C.c_uint64),
        ("node_visits_any", C.c_uint64), ("tri_tests_any", C.c_uint64),
        ("algorithmic_bytes", C.c_uint64), ("seconds_render", C.c_double),
        ("n_triangles", C.c_uint32), ("n_bvh_nodes", C.c_uint32), ("n_emitters", C.c_uint32), ("bvh_max_depth", C.c_uint32),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ptc_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("ORA_LIB") or _LIB_PATH      # ORA_LIB: an experimental build of the oracle (tools/tree_quality.py)
        if path == _LIB_PATH and not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(path)
        vp, fp, u32p, i32p, u8p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
        L.ora_create.restype = vp
        L.ora_destroy.argtypes = [vp]
        L.ora_last_error.restype = C.c_char_p
        L.ora_last_error.argtypes = [vp]
        L.ora_scene_begin.argtypes = [vp]
        L.ora_add_material.argtypes = [vp, fp, C.c_float, C.c_float, fp, C.c_int, C.c_int, C.c_int]
        L.ora_add_mesh.argtypes = [vp, vp, C.c_uint32, u32p, C.c_uint32, C.c_int]
        L.ora_add_texture_rgba8.argtypes = [vp, u8p, C.c_int, C.c_int]
        L.ora_set_env_latlong_rgb32f.argtypes = [vp, fp, C.c_int, C.c_int]
        L.ora_set_texture_filter.argtypes = [vp, C.c_int]
        L.ora_set_bvh_builder.argtypes = [vp, C.c_int]
        L.ora_atan2f.argtypes = [C.c_float, C.c_float]
        L.ora_atan2f.restype = C.c_float
        L.ora_add_instance.argtypes = [vp, C.c_int, fp, fp, fp]
        L.ora_update_instance.argtypes = [vp, C.c_int, fp, fp, fp]
        L.ora_update_instance_matrix.argtypes = [vp, C.c_int, fp]
        L.ora_scene_refit.argtypes = [vp]
        L.ora_add_instance_matrix.argtypes = [vp, C.c_int, fp]
        L.ora_set_camera.argtypes = [vp, fp, fp, C.c_float, C.c_float]
        L.ora_scene_commit.argtypes = [vp]
        L.ora_render.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp]
        L.ora_get_stats.argtypes = [vp, C.POINTER(OraStats)]
        L.ora_trace_closest.argtypes = [vp, fp, fp, C.c_uint32, fp, i32p, fp]
        L.ora_trace_any.argtypes = [vp, fp, fp, fp, C.c_uint32, u8p]
        L.ora_get_flat_scene.argtypes = [vp, u32p, u32p, vp, u32p, i32p]
        L.ora_get_bvh.argtypes = [vp, u32p, u32p, fp, fp]
        L.ora_make_model.argtypes = [fp, fp, fp, fp, fp]
        L.ora_make_model.restype = None
        L.ora_make_camera.argtypes = [fp, fp, C.c_float, C.c_float, fp, fp]
        L.ora_make_camera.restype = None
        L.ora_tonemap_rgba8.argtypes = [fp, C.c_uint32, u8p]
        L.ora_tonemap_rgba8.restype = None
        L.ora_sincos2pi.argtypes = [C.c_float, fp, fp]
        L.ora_sincos2pi.restype = None
        L.ora_powf.argtypes = [C.c_float, C.c_float]
        L.ora_powf.restype = C.c_float
        L.ora_f32_to_f16.argtypes = [C.c_float]
        L.ora_f32_to_f16.restype = C.c_uint16
        L.ora_f16_to_f32.argtypes = [C.c_uint16]
        L.ora_f16_to_f32.restype = C.c_float
        L.ora_rng_u32.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.ora_rng_u32.restype = C.c_uint32
        L.ora_tile_owner.argtypes = [C.c_int] * 5
        L.ora_hw_threads.restype = C.c_int
        _lib = L
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


class Oracle:
    """Same call sequence as pbr_amd.PathTracer, CPU scalar."""

    def __init__(self):
        self._L = lib()
        self._h = self._L.ora_create()
        if not self._h:
            raise RuntimeError("ora_create failed (host CPU lacks FMA?)")

    def close(self):
        if self._h:
            self._L.ora_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc < 0:
            raise RuntimeError(self._L.ora_last_error(self._h).decode())
        return rc

    def load_scene(self, desc):
        L, h = self._L, self._h
        self._ck(L.ora_scene_begin(h))
        for t in getattr(desc, "textures", []):
            t = np.ascontiguousarray(t, np.uint8)
            self._ck(L.ora_add_texture_rgba8(h, t.ctypes.data_as(C.POINTER(C.c_uint8)), t.shape[1], t.shape[0]))
        self._ck(L.ora_set_texture_filter(h, 1 if getattr(desc, "texture_filter", "nearest") == "linear" else 0))
        self._ck(L.ora_set_bvh_builder(h, 1 if getattr(desc, "bvh_builder", None) == "lbvh" else 0))
        env = getattr(desc, "env", None)
        if env is not None:
            e = np.ascontiguousarray(env, np.float32)
            self._ck(L.ora_set_env_latlong_rgb32f(h, e.ctypes.data_as(C.POINTER(C.c_float)), e.shape[1], e.shape[0]))
        for m in desc.materials:
            _, b = _f(m.base_color)
            _, e = _f(m.emissive)
            self._ck(L.ora_add_material(h, b, m.metallic, m.roughness, e, m.tex_color, m.tex_normal, m.tex_mr))
        for me in desc.meshes:
            v = np.ascontiguousarray(me.vertices)
            i = np.ascontiguousarray(me.indices, np.uint32)
            self._ck(L.ora_add_mesh(h, v.ctypes.data, v.size, i.ctypes.data_as(C.POINTER(C.c_uint32)), i.size, me.material))
        for it in desc.instances:
            if getattr(it, "matrix", None) is not None:
                self._ck(L.ora_add_instance_matrix(h, it.mesh, _f(np.asarray(it.matrix, np.float32).reshape(16))[1]))
            else:
                self._ck(L.ora_add_instance(h, it.mesh, _f(it.t)[1], _f(it.q_wxyz)[1], _f(it.s)[1]))
        c = desc.camera
        self._ck(L.ora_set_camera(h, _f(c.position)[1], _f(c.target)[1], c.fov_y, c.aspect))
        self._ck(L.ora_scene_commit(h))
        return self

    def update_instance(self, instance, t=None, q_wxyz=None, s=None, matrix=None):
        if matrix is not None:
            self._ck(self._L.ora_update_instance_matrix(self._h, instance, _f(np.asarray(matrix, np.float32).reshape(16))[1]))
        else:
            self._ck(self._L.ora_update_instance(self._h, instance, _f(t)[1], _f(q_wxyz)[1], _f(s)[1]))
        return self

    def scene_refit(self):
        self._ck(self._L.ora_scene_refit(self._h))
        return self

    def render(self, w, h, spp, seed=1, max_bounces=8, integrator=0, tile_rank=0, tile_count=1, n_threads=0):
        out = np.zeros((h, w, 4), np.float32)
        self._ck(self._L.ora_render(self._h, w, h, spp, seed, max_bounces, integrator, tile_rank, tile_count, n_threads,
                                    out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def stats(self):
        s = OraStats()
        self._ck(self._L.ora_get_stats(self._h, C.byref(s)))
        return s.as_dict()

    def trace_closest(self, origins, dirs):
        o, op = _f(origins)
        d, dp = _f(dirs)
        n = o.shape[0]
        t = np.zeros(n, np.float32)
        prim = np.zeros(n, np.int32)
        uv = np.zeros((n, 2), np.float32)
        self._ck(self._L.ora_trace_closest(self._h, op, dp, n, t.ctypes.data_as(C.POINTER(C.c_float)),
                                           prim.ctypes.data_as(C.POINTER(C.c_int32)), uv.ctypes.data_as(C.POINTER(C.c_float))))
        return t, prim, uv

    def trace_any(self, origins, dirs, tmax):
        o, op = _f(origins)
        d, dp = _f(dirs)
        tm, tp = _f(tmax)
        n = o.shape[0]
        occ = np.zeros(n, np.uint8)
        self._ck(self._L.ora_trace_any(self._h, op, dp, tp, n, occ.ctypes.data_as(C.POINTER(C.c_uint8))))
        return occ

    def flat_scene(self):
        from numpy import dtype
        nv, nt = C.c_uint32(), C.c_uint32()
        self._ck(self._L.ora_get_flat_scene(self._h, C.byref(nv), C.byref(nt), None, None, None))
        verts = np.zeros((nv.value, 12), np.float32)
        idx = np.zeros((nt.value, 3), np.uint32)
        tm = np.zeros(nt.value, np.int32)
        self._ck(self._L.ora_get_flat_scene(self._h, None, None, verts.ctypes.data, idx.ctypes.data_as(C.POINTER(C.c_uint32)),
                                            tm.ctypes.data_as(C.POINTER(C.c_int32))))
        return verts, idx, tm


def _bvh(self):
    nn, nt = C.c_uint32(), C.c_uint32()
    self._ck(self._L.ora_get_bvh(self._h, C.byref(nn), C.byref(nt), None, None))
    nodes = np.zeros((nn.value, 65), np.float32)
    tris = np.zeros((nt.value, 12), np.float32)
    self._ck(self._L.ora_get_bvh(self._h, None, None, nodes.ctypes.data_as(C.POINTER(C.c_float)), tris.ctypes.data_as(C.POINTER(C.c_float))))
    return nodes, tris


Oracle.bvh = _bvh


def make_model(t, q, s):
    M = np.zeros(16, np.float32)
    N = np.zeros(9, np.float32)
    lib().ora_make_model(_f(t)[1], _f(q)[1], _f(s)[1], M.ctypes.data_as(C.POINTER(C.c_float)), N.ctypes.data_as(C.POINTER(C.c_float)))
    return M.reshape(4, 4), N.reshape(3, 3)  # [col][row]


def make_camera(pos, target, fov, aspect):
    V = np.zeros(16, np.float32)
    P = np.zeros(16, np.float32)
    lib().ora_make_camera(_f(pos)[1], _f(target)[1], fov, aspect, V.ctypes.data_as(C.POINTER(C.c_float)), P.ctypes.data_as(C.POINTER(C.c_float)))
    return V.reshape(4, 4), P.reshape(4, 4)


def tonemap_rgba8(rgba):
    a, ap = _f(rgba)
    n = a.size // 4
    out = np.zeros((n, 4), np.uint8)
    lib().ora_tonemap_rgba8(ap, n, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out.reshape(a.shape[:-1] + (4,))


def sincos2pi(u):
    s, c = C.c_float(), C.c_float()
    lib().ora_sincos2pi(float(u), C.byref(s), C.byref(c))
    return s.value, c.value


def f32_to_f16(x) -> int:
    return int(lib().ora_f32_to_f16(float(x)))


def f16_to_f32(h) -> float:
    return float(lib().ora_f16_to_f32(int(h)))


def powf(x, y):
    return lib().ora_powf(float(x), float(y))


def rng_u32(seed, pixel, sample, bounce, dim):
    return lib().ora_rng_u32(seed, pixel, sample, bounce, dim)


def tile_owner(w, h, x, y, n):
    return lib().ora_tile_owner(w, h, x, y, n)


def hw_threads():
    return lib().ora_hw_threads()


# ---------------------------------------------------------------------------------------------------------------
# oracle/_ref: the reference's own image decoder (stb_image, compiled from /root/reference by `make -C oracle ref`).
_REF_STB = os.path.join(_HERE, "_ref", "libstb_image_ref.so")


def have_ref_stb() -> bool:
    return os.path.exists(_REF_STB)


def ref_stb_decode(data: bytes) -> np.ndarray:
    """What image::loadImage2D hands to Vulkan for this file: stbi_load_from_memory(..., 4) of the reference's stb build
    (src/pbr_engine/image/pbr/image/LoadImage.cpp:66-72).  Returns (h, w, 4) uint8; raises ValueError with stb's reason."""
    L = C.CDLL(_REF_STB)
    L.stbi_load_from_memory.restype = C.c_void_p
    L.stbi_load_from_memory.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    L.stbi_failure_reason.restype = C.c_char_p
    L.stbi_image_free.argtypes = [C.c_void_p]
    w, h, ch = C.c_int(0), C.c_int(0), C.c_int(0)
    p = L.stbi_load_from_memory(data, len(data), C.byref(w), C.byref(h), C.byref(ch), 4)
    if not p:
        raise ValueError((L.stbi_failure_reason() or b"stb_image failed").decode())
    out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (h.value, w.value, 4)).copy()
    L.stbi_image_free(p)
    return out


def ref_stb_decode_float(data: bytes) -> np.ndarray:
    """stbi_loadf_from_memory(..., 3) of the reference's stb build: what its image path yields for a Radiance .hdr file.  Returns (h, w, 3) float32."""
    L = C.CDLL(_REF_STB)
    L.stbi_loadf_from_memory.restype = C.c_void_p
    L.stbi_loadf_from_memory.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    L.stbi_failure_reason.restype = C.c_char_p
    L.stbi_image_free.argtypes = [C.c_void_p]
    w, h, ch = C.c_int(0), C.c_int(0), C.c_int(0)
    p = L.stbi_loadf_from_memory(data, len(data), C.byref(w), C.byref(h), C.byref(ch), 3)
    if not p:
        raise ValueError((L.stbi_failure_reason() or b"stb_image failed").decode())
    out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), (h.value, w.value, 3)).copy()
    L.stbi_image_free(p)
    return out

