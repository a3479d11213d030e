"""ctypes binding of the C-ABI in include/ptc.h (physically-based-renderer_amd/lib/libptc.so).

`PathTracer` is the drop-in for the reference's render seam: where the reference calls
`PbrRenderSystem::render(cmd, scene, gBuffer, renderTarget, extent)`
(src/pbr_engine/engine/pbr/PbrRenderSystem.hpp:46-47, called at src/gltf_viewer/App.cpp:387-388) after
`gltf::Asset::loadScene` (src/pbr_engine/gltf/pbr/gltf/Asset.hpp:76-78), a caller here does
`PathTracer(device).load_scene(desc).render(w, h, spp, ...)`.

HIP only.  There is no CPU fallback: if libptc.so is missing or no gfx950 device is usable this
module raises — loudly — instead of computing anything on the host.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PTC_LIB selects another build of the same library (the -DPT_STAMP / -DPT_DIAG diagnostic builds of tools/)
LIB_PATH = os.environ.get("PTC_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libptc.so")

DEVICE_NONE = -1  # PTC_DEVICE_NONE: description-only context (host flatten + BVH build; renders nothing)
INTEGRATOR_PATH = 0
INTEGRATOR_RASTER_COMPAT = 1
INTEGRATOR_RASTER_GBUFFER16 = 2   # raster-compat lit from the reference's G-buffer formats (RGBA16F P/N, UNORM16 albedo)
COMM_ID_BYTES = 128
ABI_VERSION = 4

# every symbol include/ptc.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "ptc_create", "ptc_destroy", "ptc_last_error", "ptc_abi_version", "ptc_build_info", "ptc_launch_policy", "ptc_scene_begin", "ptc_add_material",
    "ptc_add_texture_rgba8", "ptc_add_mesh", "ptc_add_instance", "ptc_add_instance_matrix", "ptc_update_instance", "ptc_update_instance_matrix", "ptc_scene_refit", "ptc_scene_rebuild", "ptc_set_camera", "ptc_set_env_latlong_rgb32f", "ptc_set_texture_filter", "ptc_set_bvh_builder", "ptc_scene_commit", "ptc_render",
    "ptc_frame_begin", "ptc_frame_add_samples", "ptc_frame_reserve", "ptc_frame_resolve", "ptc_frame_checkpoint", "ptc_frame_restore", "ptc_frame_set_sample_range", "ptc_sync", "ptc_read_radiance_rgba32f",
    "ptc_radiance_device_ptr", "ptc_write_radiance_rgba32f", "ptc_tonemap_rgba8", "ptc_get_stats",
    "ptc_debug_trace_closest", "ptc_debug_trace_any", "ptc_debug_get_flat_scene", "ptc_debug_get_bvh", "ptc_debug_get_counters",
    "ptc_debug_get_description", "ptc_debug_get_material", "ptc_debug_get_texture", "ptc_debug_get_internals", "ptc_debug_host_build_id", "ptc_debug_get_shading_tables", "ptc_debug_refit_host_parts", "ptc_debug_commit_host_parts",
    "ptc_read_radiance_rgba16f", "ptc_radiance_rgba16f_device_ptr",
    "ptc_comm_unique_id", "ptc_comm_init", "ptc_comm_reduce_radiance", "ptc_comm_destroy",
    "ptc_group_create", "ptc_group_size", "ptc_group_scene_commit", "ptc_group_scene_refit", "ptc_group_ctx", "ptc_group_render", "ptc_group_last_error", "ptc_group_destroy",
]


class PtcStats(C.Structure):
    _fields_ = [
        ("paths", C.c_uint64), ("segments", C.c_uint64), ("shadow_rays", C.c_uint64), ("hits", C.c_uint64),
        ("node_visits_closest", C.c_uint64), ("tri_tests_closest", C.c_uint64),
        ("node_visits_any", C.c_uint64), ("tri_tests_any", C.c_uint64), ("algorithmic_bytes", C.c_uint64),
        ("seconds_render", C.c_double), ("seconds_trace_closest", C.c_double), ("seconds_trace_any", C.c_double),
        ("seconds_shade", C.c_double), ("seconds_commit", C.c_double), ("seconds_reduce", C.c_double), ("seconds_refit", C.c_double),
        ("launches_trace_closest", C.c_uint32), ("launches_trace_any", C.c_uint32),
        ("n_triangles", C.c_uint32), ("n_bvh_nodes", C.c_uint32), ("n_emitters", C.c_uint32), ("bvh_max_depth", C.c_uint32),
        ("bvh_sa_cost", C.c_double), ("bvh_sa_cost_built", C.c_double), ("seconds_rebuild", C.c_double),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class PtcError(RuntimeError):
    pass


_lib = None


def load_library():
    """Load libptc.so and declare the prototypes.  Raises PtcError when the HIP library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PtcError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, fp = C.c_void_p, C.POINTER(C.c_float)
    u32p, i32p, u8p = C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    L.ptc_create.restype = vp
    L.ptc_create.argtypes = [C.c_int]
    L.ptc_destroy.argtypes = [vp]
    L.ptc_destroy.restype = None
    L.ptc_last_error.restype = C.c_char_p
    L.ptc_last_error.argtypes = [vp]
    L.ptc_abi_version.restype = C.c_int
    L.ptc_build_info.restype = C.c_char_p
    L.ptc_launch_policy.restype = C.c_char_p
    L.ptc_launch_policy.argtypes = [vp]
    L.ptc_scene_begin.argtypes = [vp]
    L.ptc_add_material.argtypes = [vp, fp, C.c_float, C.c_float, fp, C.c_int, C.c_int, C.c_int]
    L.ptc_add_texture_rgba8.argtypes = [vp, u8p, C.c_int, C.c_int]
    L.ptc_add_mesh.argtypes = [vp, vp, C.c_uint32, u32p, C.c_uint32, C.c_int]
    L.ptc_add_instance.argtypes = [vp, C.c_int, fp, fp, fp]
    L.ptc_add_instance_matrix.argtypes = [vp, C.c_int, fp]
    L.ptc_update_instance.argtypes = [vp, C.c_int, fp, fp, fp]
    L.ptc_update_instance_matrix.argtypes = [vp, C.c_int, fp]
    L.ptc_scene_refit.argtypes = [vp]
    L.ptc_scene_rebuild.argtypes = [vp]
    L.ptc_set_camera.argtypes = [vp, fp, fp, C.c_float, C.c_float]
    L.ptc_set_env_latlong_rgb32f.argtypes = [vp, fp, C.c_int, C.c_int]
    L.ptc_set_texture_filter.argtypes = [vp, C.c_int]
    L.ptc_set_bvh_builder.argtypes = [vp, C.c_int]
    L.ptc_scene_commit.argtypes = [vp]
    L.ptc_render.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int]
    L.ptc_frame_begin.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int]
    L.ptc_frame_add_samples.argtypes = [vp, C.c_int]
    L.ptc_frame_resolve.argtypes = [vp]
    L.ptc_frame_reserve.argtypes = [vp]
    L.ptc_frame_checkpoint.argtypes = [vp, fp, C.POINTER(C.c_uint64), u32p]
    L.ptc_frame_restore.argtypes = [vp, fp, C.c_uint64, C.c_uint32]
    L.ptc_frame_set_sample_range.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.ptc_sync.argtypes = [vp]
    L.ptc_read_radiance_rgba32f.argtypes = [vp, fp]
    L.ptc_radiance_device_ptr.argtypes = [vp]
    L.ptc_radiance_device_ptr.restype = vp
    L.ptc_write_radiance_rgba32f.argtypes = [vp, fp]
    L.ptc_tonemap_rgba8.argtypes = [vp, u8p]
    L.ptc_get_stats.argtypes = [vp, C.POINTER(PtcStats)]
    L.ptc_debug_trace_closest.argtypes = [vp, fp, fp, C.c_uint32, fp, i32p, fp]
    L.ptc_debug_trace_any.argtypes = [vp, fp, fp, fp, C.c_uint32, u8p]
    L.ptc_debug_get_flat_scene.argtypes = [vp, u32p, u32p, vp, u32p, i32p]
    L.ptc_debug_get_bvh.argtypes = [vp, u32p, u32p, u32p, fp, fp]
    L.ptc_debug_get_counters.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int]
    L.ptc_debug_get_description.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ptc_debug_get_material.argtypes = [vp, C.c_int, fp, C.POINTER(C.c_int)]
    L.ptc_debug_get_texture.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), vp]
    L.ptc_debug_get_internals.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.ptc_debug_host_build_id.argtypes = [vp]
    L.ptc_debug_host_build_id.restype = C.c_uint64
    L.ptc_debug_get_shading_tables.argtypes = [vp, u32p, fp, u32p, fp, fp]
    L.ptc_debug_refit_host_parts.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.ptc_debug_commit_host_parts.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.ptc_read_radiance_rgba16f.argtypes = [vp, C.POINTER(C.c_uint16)]
    L.ptc_radiance_rgba16f_device_ptr.argtypes = [vp]
    L.ptc_radiance_rgba16f_device_ptr.restype = vp
    L.ptc_comm_unique_id.argtypes = [u8p]
    L.ptc_comm_init.argtypes = [vp, u8p, C.c_int, C.c_int]
    L.ptc_comm_reduce_radiance.argtypes = [vp, C.c_int]
    L.ptc_comm_destroy.argtypes = [vp]
    L.ptc_group_create.argtypes = [C.POINTER(C.c_int), C.c_int]
    L.ptc_group_create.restype = vp
    L.ptc_group_size.argtypes = [vp]
    L.ptc_group_scene_commit.argtypes = [vp]
    L.ptc_group_scene_refit.argtypes = [vp]
    L.ptc_group_ctx.argtypes = [vp, C.c_int]
    L.ptc_group_ctx.restype = vp
    L.ptc_group_render.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int]
    L.ptc_group_last_error.argtypes = [vp]
    L.ptc_group_last_error.restype = C.c_char_p
    L.ptc_group_destroy.argtypes = [vp]
    L.ptc_group_destroy.restype = None
    _lib = L
    return L


def _f(a):
    a = np.ascontiguousarray(a, np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def comm_unique_id() -> bytes:
    """ptc_comm_unique_id: the 128 bytes rank 0 ships to the other ranks before ptc_comm_init."""
    L = load_library()
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    if L.ptc_comm_unique_id(buf) < 0:
        raise PtcError(L.ptc_last_error(None).decode())
    return bytes(buf)


class PathTracer:
    def __init__(self, device: int = 0, _handle=None):
        self._L = load_library()
        self._owned = _handle is None
        self._h = self._L.ptc_create(int(device)) if _handle is None else _handle
        if not self._h:
            raise PtcError(self._L.ptc_last_error(None).decode())
        self.device = int(device)
        self._w = self._h_px = 0

    def close(self):
        if getattr(self, "_h", None):
            if self._owned:
                self._L.ptc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc < 0:
            raise PtcError(f"ptc error {rc}: {self._L.ptc_last_error(self._h).decode()}")
        return rc

    # ---- scene ------------------------------------------------------------------------------------
    def load_scene(self, desc):
        L, h = self._L, self._h
        self._ck(L.ptc_scene_begin(h))
        for t in getattr(desc, "textures", []):
            t = np.ascontiguousarray(t, np.uint8)
            assert t.ndim == 3 and t.shape[2] == 4, "textures are (h, w, 4) uint8"
            self._ck(L.ptc_add_texture_rgba8(h, t.ctypes.data_as(C.POINTER(C.c_uint8)), t.shape[1], t.shape[0]))
        self._ck(L.ptc_set_texture_filter(h, 1 if getattr(desc, "texture_filter", "nearest") == "linear" else 0))
        if getattr(desc, "bvh_builder", None) is not None:      # None: the context's default (SAH, or what PTC_BVH says)
            self._ck(L.ptc_set_bvh_builder(h, {"sah": 0, "lbvh": 1}[desc.bvh_builder]))
        env = getattr(desc, "env", None)
        if env is not None:
            e = np.ascontiguousarray(env, np.float32)
            assert e.ndim == 3 and e.shape[2] == 3, "env is (h, w, 3) float32"
            self._ck(L.ptc_set_env_latlong_rgb32f(h, e.ctypes.data_as(C.POINTER(C.c_float)), e.shape[1], e.shape[0]))
        for m in desc.materials:
            self._ck(L.ptc_add_material(h, _f(m.base_color)[1], m.metallic, m.roughness, _f(m.emissive)[1], m.tex_color, m.tex_normal, m.tex_mr))
        for me in desc.meshes:
            v = np.ascontiguousarray(me.vertices)
            i = np.ascontiguousarray(me.indices, np.uint32)
            self._ck(L.ptc_add_mesh(h, v.ctypes.data, v.size, i.ctypes.data_as(C.POINTER(C.c_uint32)), i.size, me.material))
        for it in desc.instances:
            if getattr(it, "matrix", None) is not None:
                self._ck(L.ptc_add_instance_matrix(h, it.mesh, _f(np.asarray(it.matrix, np.float32).reshape(16))[1]))
            else:
                self._ck(L.ptc_add_instance(h, it.mesh, _f(it.t)[1], _f(it.q_wxyz)[1], _f(it.s)[1]))
        c = desc.camera
        self._ck(L.ptc_set_camera(h, _f(c.position)[1], _f(c.target)[1], c.fov_y, c.aspect))
        self._ck(L.ptc_scene_commit(h))
        return self

    def set_camera(self, position, target, fov_y, aspect):
        self._ck(self._L.ptc_set_camera(self._h, _f(position)[1], _f(target)[1], fov_y, aspect))

    def update_instance(self, instance, t=None, q_wxyz=None, s=None, matrix=None):
        """New transform for a committed instance (ptc_update_instance / ptc_update_instance_matrix); scene_refit() applies it."""
        if matrix is not None:
            self._ck(self._L.ptc_update_instance_matrix(self._h, instance, _f(np.asarray(matrix, np.float32).reshape(16))[1]))
        else:
            self._ck(self._L.ptc_update_instance(self._h, instance, _f(t)[1], _f(q_wxyz)[1], _f(s)[1]))
        return self

    def scene_refit(self):
        self._ck(self._L.ptc_scene_refit(self._h))
        return self

    def host_build_id(self):
        return int(self._L.ptc_debug_host_build_id(self._h))

    def launch_policy(self):
        """ptc_launch_policy of this context (after a commit: with the launch configuration that followed)."""
        return self._L.ptc_launch_policy(self._h).decode()

    def scene_rebuild(self):
        """ptc_scene_rebuild: pending transforms + a new LBVH for the moved geometry, built on the device."""
        self._ck(self._L.ptc_scene_rebuild(self._h))
        return self

    # ---- rendering --------------------------------------------------------------------------------
    def render(self, w, h, spp, seed=1, max_bounces=8, integrator=INTEGRATOR_PATH):
        self._ck(self._L.ptc_render(self._h, w, h, spp, seed, max_bounces, integrator))
        self._w, self._h_px = w, h
        return self.read_radiance()

    def frame_begin(self, w, h, spp_total, seed=1, max_bounces=8, integrator=INTEGRATOR_PATH, tile_rank=0, tile_count=1):
        self._ck(self._L.ptc_frame_begin(self._h, w, h, spp_total, seed, max_bounces, integrator, tile_rank, tile_count))
        self._w, self._h_px = w, h

    def frame_checkpoint(self):
        """(per-pixel sums as an opaque (n_owned, 4) float32 array, samples in them) of the frame in progress (ptc_frame_checkpoint)."""
        n, k = C.c_uint64(0), C.c_uint32(0)
        self._ck(self._L.ptc_frame_checkpoint(self._h, None, C.byref(n), C.byref(k)))
        acc = np.zeros((n.value, 4), np.float32)
        self._ck(self._L.ptc_frame_checkpoint(self._h, acc.ctypes.data_as(C.POINTER(C.c_float)), C.byref(n), C.byref(k)))
        return acc, int(k.value)

    def frame_restore(self, accum, samples_done):
        a = np.ascontiguousarray(accum, np.float32)
        self._ck(self._L.ptc_frame_restore(self._h, a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[0], samples_done))

    def frame_set_sample_range(self, first_sample, resolve_divisor=0):
        self._ck(self._L.ptc_frame_set_sample_range(self._h, first_sample, resolve_divisor))

    def frame_add_samples(self, n):
        self._ck(self._L.ptc_frame_add_samples(self._h, n))

    def frame_reserve(self):
        """Allocate the frame's queues for full batches now (offline renders; see ptc_frame_reserve)."""
        self._ck(self._L.ptc_frame_reserve(self._h))

    def frame_resolve(self):
        self._ck(self._L.ptc_frame_resolve(self._h))

    def sync(self):
        self._ck(self._L.ptc_sync(self._h))

    def read_radiance(self):
        out = np.empty((self._h_px, self._w, 4), np.float32)
        self._ck(self._L.ptc_read_radiance_rgba32f(self._h, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def write_radiance(self, img):
        a, p = _f(img)
        assert a.size == self._w * self._h_px * 4
        self._ck(self._L.ptc_write_radiance_rgba32f(self._h, p))

    def radiance_device_ptr(self) -> int:
        return int(self._L.ptc_radiance_device_ptr(self._h) or 0)

    def read_radiance_f16(self):
        """The radiance buffer as the reference's RGBA16F HdrImage holds it: (h, w, 4) float16."""
        out = np.empty((self._h_px, self._w, 4), np.uint16)
        self._ck(self._L.ptc_read_radiance_rgba16f(self._h, out.ctypes.data_as(C.POINTER(C.c_uint16))))
        return out.view(np.float16)

    def radiance_f16_device_ptr(self) -> int:
        return int(self._L.ptc_radiance_rgba16f_device_ptr(self._h) or 0)

    # ---- multi-GPU (RCCL through the C-ABI) ---------------------------------------------------------
    def comm_init(self, unique_id: bytes, rank: int, n_ranks: int):
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        self._ck(self._L.ptc_comm_init(self._h, buf, rank, n_ranks))

    def comm_reduce_radiance(self, root: int = 0):
        self._ck(self._L.ptc_comm_reduce_radiance(self._h, root))

    def comm_destroy(self):
        self._ck(self._L.ptc_comm_destroy(self._h))

    def tonemap(self):
        out = np.empty((self._h_px, self._w, 4), np.uint8)
        self._ck(self._L.ptc_tonemap_rgba8(self._h, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def stats(self):
        s = PtcStats()
        self._ck(self._L.ptc_get_stats(self._h, C.byref(s)))
        return s.as_dict()

    # ---- test hooks -------------------------------------------------------------------------------
    def trace_closest(self, origins, dirs):
        o, op = _f(origins)
        d, dp = _f(dirs)
        n = o.shape[0]
        t = np.zeros(n, np.float32)
        prim = np.zeros(n, np.int32)
        uv = np.zeros((n, 2), np.float32)
        self._ck(self._L.ptc_debug_trace_closest(self._h, op, dp, n, t.ctypes.data_as(C.POINTER(C.c_float)),
                                                 prim.ctypes.data_as(C.POINTER(C.c_int32)), uv.ctypes.data_as(C.POINTER(C.c_float))))
        return t, prim, uv

    def trace_any(self, origins, dirs, tmax):
        o, op = _f(origins)
        d, dp = _f(dirs)
        tm, tp = _f(tmax)
        n = o.shape[0]
        occ = np.zeros(n, np.uint8)
        self._ck(self._L.ptc_debug_trace_any(self._h, op, dp, tp, n, occ.ctypes.data_as(C.POINTER(C.c_uint8))))
        return occ

    def flat_scene(self):
        nv, nt = C.c_uint32(), C.c_uint32()
        self._ck(self._L.ptc_debug_get_flat_scene(self._h, C.byref(nv), C.byref(nt), None, None, None))
        verts = np.zeros((nv.value, 12), np.float32)
        idx = np.zeros((nt.value, 3), np.uint32)
        tm = np.zeros(nt.value, np.int32)
        self._ck(self._L.ptc_debug_get_flat_scene(self._h, None, None, verts.ctypes.data, idx.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                  tm.ctypes.data_as(C.POINTER(C.c_int32))))
        return verts, idx, tm

    def description(self):
        """Materials and textures as the context received them: ([(factors9, (tex_color, tex_normal, tex_mr))], [rgba arrays])."""
        nm, nt = C.c_int(0), C.c_int(0)
        self._ck(self._L.ptc_debug_get_description(self._h, C.byref(nm), C.byref(nt)))
        mats, texs = [], []
        for i in range(nm.value):
            f = np.zeros(9, np.float32)
            t = (C.c_int * 3)()
            self._ck(self._L.ptc_debug_get_material(self._h, i, f.ctypes.data_as(C.POINTER(C.c_float)), t))
            mats.append((f, tuple(t)))
        for i in range(nt.value):
            w, h = C.c_int(0), C.c_int(0)
            self._ck(self._L.ptc_debug_get_texture(self._h, i, C.byref(w), C.byref(h), None))
            px = np.zeros((h.value, w.value, 4), np.uint8)
            self._ck(self._L.ptc_debug_get_texture(self._h, i, C.byref(w), C.byref(h), px.ctypes.data))
            texs.append(px)
        return mats, texs

    def bvh(self):
        """(units, n_nodes, n_tris, grid): the BVH's unit array as (n_units, 4) float32 (see include/ptc.h), node / triangle-record
        counts and the origin grid (scene_lo.xyz, step.xyz)."""
        nn, nt, nu = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._ck(self._L.ptc_debug_get_bvh(self._h, C.byref(nn), C.byref(nt), C.byref(nu), None, None))
        units = np.zeros((nu.value, 4), np.float32)
        grid = np.zeros(6, np.float32)
        self._ck(self._L.ptc_debug_get_bvh(self._h, None, None, None, units.ctypes.data_as(C.POINTER(C.c_float)), grid.ctypes.data_as(C.POINTER(C.c_float))))
        return units, nn.value, nt.value, grid

    def internals(self):
        buf = (C.c_uint64 * 8)()
        self._ck(self._L.ptc_debug_get_internals(self._h, buf))
        keys = ("events_created", "spans_waiting", "queue_cap", "per_batch", "pending", "trace_blocks_per_cu", "stack_lds", "refit_on_device")
        d = {k: int(buf[i]) for i, k in enumerate(keys)}
        d["commit_on_device"] = (d["refit_on_device"] >> 1) & 1
        d["refit_on_device"] &= 1
        return d

    def commit_host_parts(self):
        """The host's share of a commit on the device, checked against the host build (ptc_debug_commit_host_parts)."""
        buf = (C.c_uint64 * 8)()
        self._ck(self._L.ptc_debug_commit_host_parts(self._h, buf))
        keys = ("n_tris", "n_lights", "indices_ok", "emitters_ok", "materials_ok", "tables_ok", "sizes_ok")
        return {k: int(buf[i]) for i, k in enumerate(keys)}

    def refit_host_parts(self):
        """The host's share of a refit on the device, checked against the host build (ptc_debug_refit_host_parts)."""
        buf = (C.c_uint64 * 8)()
        self._ck(self._L.ptc_debug_refit_host_parts(self._h, buf))
        keys = ("n_verts", "n_tris", "nodes_listed", "levels", "emissive_prims", "levels_ok", "emitters_equal", "transforms_finite")
        return {k: int(buf[i]) for i, k in enumerate(keys)}

    def shading_tables(self):
        """(shade, lights, cdf): the per-primitive shading records as (n_tris, 4 * stride) float32, the emitter table (n, 20) and its cdf,
        as they lie in HBM (after a refit on the device they are read back first)."""
        stride, nl = C.c_uint32(), C.c_uint32()
        self._ck(self._L.ptc_debug_get_shading_tables(self._h, C.byref(stride), None, C.byref(nl), None, None))
        n = self.stats()["n_triangles"]
        shade = np.zeros((n, 4 * stride.value), np.float32)
        lights = np.zeros((max(nl.value, 1), 20), np.float32)
        cdf = np.zeros(max(nl.value, 1), np.float32)
        f = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        self._ck(self._L.ptc_debug_get_shading_tables(self._h, None, f(shade), None, f(lights), f(cdf)))
        return shade, lights[:max(nl.value, 1)], cdf

    def raw_counters(self):
        buf = (C.c_uint64 * 32)()
        n = self._ck(self._L.ptc_debug_get_counters(self._h, buf, 32))
        return [int(buf[i]) for i in range(n)]


class Group:
    """ptc_group: one process driving several GPUs (n contexts + ncclCommInitAll).  `ctx(i)` is a PathTracer view of
    device i's context; load the same scene into each, then `render` (tiles sharded, RCCL reduce onto device 0)."""

    def __init__(self, device_ids):
        self._L = load_library()
        ids = (C.c_int * len(device_ids))(*[int(d) for d in device_ids])
        self._g = self._L.ptc_group_create(ids, len(device_ids))
        if not self._g:
            raise PtcError(self._L.ptc_group_last_error(None).decode())
        self.device_ids = [int(d) for d in device_ids]
        self._ctx = [PathTracer(d, _handle=self._L.ptc_group_ctx(self._g, i)) for i, d in enumerate(self.device_ids)]

    def __len__(self):
        return int(self._L.ptc_group_size(self._g))

    def ctx(self, i) -> PathTracer:
        return self._ctx[i]

    def load_scene(self, desc):
        """Describe the scene on device 0 and commit it to every device with ONE host build (ptc_group_scene_commit)."""
        self._ctx[0].load_scene(desc)
        rc = self._L.ptc_group_scene_commit(self._g)
        if rc < 0:
            raise PtcError(f"ptc error {rc}: {self._L.ptc_group_last_error(self._g).decode()}")
        return self

    def scene_refit(self):
        """After ctx(0).update_instance(...): one host refit, uploaded to every device (ptc_group_scene_refit)."""
        rc = self._L.ptc_group_scene_refit(self._g)
        if rc < 0:
            raise PtcError(f"ptc error {rc}: {self._L.ptc_group_last_error(self._g).decode()}")
        return self

    def render(self, w, h, spp, seed=1, max_bounces=8, integrator=INTEGRATOR_PATH):
        rc = self._L.ptc_group_render(self._g, w, h, spp, seed, max_bounces, integrator)
        if rc < 0:
            raise PtcError(f"ptc error {rc}: {self._L.ptc_group_last_error(self._g).decode()}")
        for c in self._ctx:
            c._w, c._h_px = w, h
        return self._ctx[0].read_radiance()

    def close(self):
        if getattr(self, "_g", None):
            for c in self._ctx:
                c._h = None
            self._L.ptc_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
