"""The C-ABI's frame / output / multi-GPU contract on a real MI355X (everything through include/ptc.h):
progressive resolve, deferred batching and queue sizing, the bounded event pool, the RGBA16F hand-off of the
reference's HdrImage format, the G-buffer-format raster pass, RCCL through the C-ABI, and BASELINE configs 2 and 5
at their full sizes."""
import os
import subprocess
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import ROOT, rel_l2  # noqa: E402

pytestmark = pytest.mark.gpu

COUNTERS = ("paths", "segments", "shadow_rays", "hits", "node_visits_closest", "tri_tests_closest", "node_visits_any", "tri_tests_any", "algorithmic_bytes")


@pytest.fixture(scope="module")
def gpu(pbr):
    import torch

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return pbr


def _bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def test_progressive_resolve_shows_the_k_sample_frame(gpu, ora):
    """INTEGRATION.md §2's viewer loop: frame_begin(budget); add_samples(1); resolve; read — after k of N samples the
    buffer holds the k-sample image (divisor = samples so far, not the budget), and the budget is enforced."""
    d = gpu.scenes.sphere_scene(32, 17)
    pt, o = gpu.PathTracer(0).load_scene(d), ora.Oracle().load_scene(d)
    w, h, budget = 96, 64, 6
    pt.frame_begin(w, h, budget, 9, 4, 0)
    for k in range(1, budget + 1):
        pt.frame_add_samples(1)
        pt.frame_resolve()
        assert _bits_equal(pt.read_radiance(), o.render(w, h, k, seed=9, max_bounces=4)), k
    with pytest.raises(gpu.PtcError, match="spp_total"):
        pt.frame_add_samples(1)                                   # past the budget
    pt.frame_resolve()                                            # the frame is still intact
    assert _bits_equal(pt.read_radiance(), o.render(w, h, budget, seed=9, max_bounces=4))
    # resolve before any sample: zeros, no division by zero
    pt.frame_begin(w, h, 2, 9, 4, 0)
    pt.frame_resolve()
    assert (pt.read_radiance() == 0).all()


def test_queues_follow_the_batches_and_events_stay_bounded(gpu):
    """a still-camera loop that adds one sample per displayed frame at 1080p must not allocate queues for the
    whole budget (it did: 47 GB) nor create HIP events without bound (it did: ~56 per call)."""
    pt = gpu.PathTracer(0).load_scene(gpu.scenes.cornell_box())
    w, h = 1920, 1080
    pt.frame_begin(w, h, 4096, 1, 4, 0)
    assert pt.internals()["queue_cap"] == 0 or pt.internals()["queue_cap"] <= w * h      # nothing sized by the budget
    for _ in range(40):
        pt.frame_add_samples(1)
        pt.frame_resolve()
        pt.sync()
    it = pt.internals()
    assert it["queue_cap"] == w * h and it["pending"] == 0
    first = it["events_created"]
    for _ in range(160):
        pt.frame_add_samples(1)
        pt.frame_resolve()
    pt.sync()
    it = pt.internals()
    assert it["events_created"] <= max(first, 2 * 1024 + 64) and it["spans_waiting"] <= 1024 + 64
    assert pt.stats()["paths"] == 200 * w * h and pt.internals()["spans_waiting"] == 0


def test_timing_spans_follow_the_batch(gpu):
    """ptc_stats' seconds: a small batch (<= 2^26 paths) runs k_trace_any(b) beside k_trace_closest(b + 1) on two streams, so it carries the batch's span only — per-kernel
    spans would include each other, and their 54 event records were 0.2 ms of a 1080p x 1 spp frame (profiles/r04_interactive.txt); a large batch runs its kernels one after
    the other and carries a span per kernel; PTC_TIMING=2 records per-kernel spans always, PTC_TIMING=0 none.  The images do not depend on any of it."""
    d = gpu.scenes.cornell_box()
    w, h = 512, 512

    def run(env, spp):
        old = os.environ.get("PTC_TIMING")
        if env is None:
            os.environ.pop("PTC_TIMING", None)
        else:
            os.environ["PTC_TIMING"] = env
        try:
            pt = gpu.PathTracer(0).load_scene(d)
        finally:
            os.environ.pop("PTC_TIMING", None)
            if old is not None:
                os.environ["PTC_TIMING"] = old
        img = pt.render(w, h, spp, seed=2, max_bounces=3)
        return img, pt.stats(), pt.internals()["events_created"]
    img1, s1, e1 = run(None, 2)                                    # 524 k paths: overlapped
    assert s1["seconds_render"] > 0 and s1["seconds_trace_closest"] == s1["seconds_trace_any"] == s1["seconds_shade"] == 0.0 and e1 <= 4
    img2, s2, e2 = run("2", 2)
    assert s2["seconds_trace_closest"] > 0 and s2["seconds_trace_any"] > 0 and s2["seconds_shade"] > 0 and e2 > 10
    img0, s0, e0 = run("0", 2)
    assert s0["seconds_render"] == 0.0 and e0 == 0
    assert _bits_equal(img1, img2) and _bits_equal(img1, img0)
    _, s3, _ = run(None, 300)                                      # 78.6 M paths in one batch: one kernel after the other, a span each
    assert s3["seconds_trace_closest"] > 0 and s3["seconds_shade"] > 0 and s3["seconds_render"] >= s3["seconds_trace_closest"]


def test_frame_reserve_allocates_full_batches_up_front(gpu):
    """ptc_frame_reserve: an offline render sizes its queues for full batches before it starts, so that no growth step (which
    drains the device and reallocates) falls into the render; the capacity then stays put and the image is the same."""
    pt = gpu.PathTracer(0).load_scene(gpu.scenes.cornell_box())
    w, h, spp = 320, 200, 24
    ref = pt.render(w, h, spp, seed=6)
    pt.frame_begin(w, h, spp, 6, 8, 0)
    pt.frame_reserve()
    it = pt.internals()
    cap = it["queue_cap"]
    assert cap == w * h * min(it["per_batch"], spp)
    for _ in range(spp // 4):
        pt.frame_add_samples(4)
    pt.frame_resolve()
    pt.sync()
    assert pt.internals()["queue_cap"] == cap
    assert np.array_equal(pt.read_radiance().view(np.uint32), ref.view(np.uint32))
    with pytest.raises(gpu.PtcError):
        gpu.PathTracer(0).frame_reserve()            # no frame


def test_batch_size_is_clamped_to_free_device_memory(gpu):
    """frame_begin sizes a full batch by PTC_BATCH_PATHS but never beyond what 60 % of the free device memory holds in queues
    (176 B per path), so that an oversized setting (or a card that is partly taken) degrades to smaller batches, not to PTC_E_NOMEM."""
    import torch

    free_b, total_b = torch.cuda.mem_get_info(0)
    os.environ["PTC_BATCH_PATHS"] = str(1 << 40)
    try:
        pt = gpu.PathTracer(0).load_scene(gpu.scenes.cornell_box())
        w, h = 1920, 1080
        pt.frame_begin(w, h, 1 << 20, 1, 4, 0)
        per = pt.internals()["per_batch"]
        assert pt.internals()["queue_cap"] == 0                                    # nothing allocated by frame_begin itself
        assert 0.3 * free_b < per * w * h * 176 <= 0.62 * total_b
        assert per * w * h < 1 << 32                                               # queue slots are 32-bit
    finally:
        del os.environ["PTC_BATCH_PATHS"]


def test_owned_pixel_list_is_kept_only_while_it_is_valid(gpu):
    """ptc_frame_begin keeps the device's list of owned pixels from frame to frame (a viewer renders one size over and over); a change of the image
    size or of the tile assignment must replace it: a context taken through a sequence of sizes and tile ranks renders, every time, the bits of
    a fresh context given that frame alone — and a 1-sample-per-call progressive frame in between is not disturbed either."""
    d = gpu.scenes.by_name("cornell")
    pt = gpu.PathTracer(0).load_scene(d)
    def frame(p, w, h, rank, count, seed):
        p.frame_begin(w, h, 3, seed=seed, max_bounces=4, tile_rank=rank, tile_count=count)
        for _ in range(3):
            p.frame_add_samples(1)
        p.frame_resolve(); p.sync()
        return p.read_radiance()
    seq = [(48, 32, 0, 1, 5), (48, 32, 0, 1, 6), (48, 32, 1, 2, 6), (48, 32, 0, 2, 6), (32, 48, 0, 1, 6), (48, 32, 0, 1, 5), (96, 64, 2, 3, 7), (48, 32, 0, 1, 5)]
    first = None
    for w, h, rank, count, seed in seq:
        got = frame(pt, w, h, rank, count, seed)
        want = frame(gpu.PathTracer(0).load_scene(d), w, h, rank, count, seed)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (w, h, rank, count, seed)
        if (w, h, rank, count, seed) == seq[0]:
            first = got if first is None else first
            assert np.array_equal(got.view(np.uint32), first.view(np.uint32))


def test_small_calls_merge_into_full_batches(gpu, ora):
    """Deferred batching: samples added in small calls are issued as full wavefront batches (launches as wide as one
    big call's), a partial batch goes out at resolve; bits do not depend on how the samples were handed over."""
    d = gpu.scenes.atrium(0.05)
    pt = gpu.PathTracer(0).load_scene(d)
    w, h, spp = 160, 90, 12
    ref = pt.render(w, h, spp, seed=4)
    n_big = pt.stats()["launches_trace_closest"]
    pt.frame_begin(w, h, spp, 4, 8, 0)
    per = pt.internals()["per_batch"]
    assert per >= spp                                                  # a small frame: everything fits one batch
    for _ in range(spp):
        pt.frame_add_samples(1)
    assert pt.internals()["pending"] == spp                            # nothing issued yet
    pt.frame_resolve()
    assert _bits_equal(pt.read_radiance(), ref)
    assert pt.stats()["launches_trace_closest"] == n_big               # one batch, not twelve
    for lanes in (1, 2, 3):                                            # one stream (the default) and batches alternating over 2 and 3
        os.environ["PTC_BATCH_PATHS"] = str(w * h * 5 * lanes)          # per_batch = 5 samples
        os.environ["PTC_LANES"] = str(lanes)
        try:
            small = gpu.PathTracer(0).load_scene(d)
            small.frame_begin(w, h, spp, 4, 8, 0)
            assert small.internals()["per_batch"] == 5
            for k in range(spp):
                small.frame_add_samples(1)
                assert small.internals()["pending"] == (k + 1) % 5
            small.frame_resolve()
            assert _bits_equal(small.read_radiance(), ref)
            assert small.stats()["launches_trace_closest"] == 3 * (8 + 1)  # batches of 5, 5 and 2 samples
        finally:
            del os.environ["PTC_BATCH_PATHS"], os.environ["PTC_LANES"]


def test_rgba16f_output_matches_numpy_float16(gpu):
    """The reference's lighting target / HdrImage is RGBA16F (PbrRenderSystem.hpp:21, HdrImage.cpp:20): the half buffer is
    the fp32 buffer rounded to nearest even — numpy's float16 conversion — including denormals, overflow and inf."""
    pt = gpu.PathTracer(0).load_scene(gpu.scenes.cornell_box())
    w, h = 64, 48
    img = pt.render(w, h, 4, seed=2)
    with np.errstate(over="ignore"):
        assert np.array_equal(pt.read_radiance_f16().view(np.uint16), img.astype(np.float16).view(np.uint16))
    rng = np.random.default_rng(3)
    special = np.array([0.0, -0.0, 1.0, -1.0, 65504.0, 65519.9, 65520.0, 65536.0, 1e9, -1e9, np.inf, -np.inf, 6.1035156e-05, 6.0975552e-05, 5.9604645e-08,
                        2.9802322e-08, 2.9802326e-08, 8.9406967e-08, 1e-10, 0.33325195, 0.33337402, 2049.0, 2051.0, 1.0009766, 1.0004883, 1.0014648], np.float32)
    vals = np.concatenate([special, (rng.standard_normal(3000) * 10.0 ** rng.uniform(-9, 6, 3000)).astype(np.float32),
                           rng.integers(0, 2 ** 32, w * h * 4, dtype=np.uint64).astype(np.uint32).view(np.float32)])[: w * h * 4]
    vals = np.where(np.isnan(vals), np.float32(1.5), vals).astype(np.float32).reshape(h, w, 4)
    pt.write_radiance(vals)
    with np.errstate(over="ignore"):
        want = vals.astype(np.float16).view(np.uint16)
    got = pt.read_radiance_f16().view(np.uint16)
    assert np.array_equal(got, want), np.argwhere(got != want)[:5]
    nan = vals.copy()
    nan[0, 0, 0] = np.nan
    pt.write_radiance(nan)
    assert np.isnan(pt.read_radiance_f16()[0, 0, 0])
    # the device-pointer form hands the same bytes to a consumer on the GPU (the viewer shim's staged copy)
    import torch

    ptr = pt.radiance_f16_device_ptr()
    assert ptr
    t = torch.as_tensor(gpu.dist._DevArray(ptr, w * h * 2), device="cuda:0")   # 8 bytes per pixel, viewed as 2 floats: compare raw bytes
    raw = np.frombuffer(t.cpu().numpy().tobytes(), np.uint16).reshape(h, w, 4)
    assert np.array_equal(raw, pt.read_radiance_f16().view(np.uint16))


@pytest.mark.parametrize("name,kw,w,h", [("two_tris_sphere", {}, 64, 64), ("sphere10k", {}, 101, 67), ("textured_objects", {}, 96, 96),
                                         ("textured_atrium", {"scale": 0.05, "tex_size": 128, "env_size": (64, 32)}, 160, 90)])
def test_raster_gbuffer16_matches_the_oracle(gpu, ora, name, kw, w, h):
    """GBuffer.hpp:13-16: the reference lights from RGBA16F positions/normals and UNORM16 albedo and writes RGBA16F."""
    d = gpu.scenes.by_name(name, **kw)
    pt, o = gpu.PathTracer(0).load_scene(d), ora.Oracle().load_scene(d)
    g = pt.render(w, h, 1, integrator=gpu.INTEGRATOR_RASTER_GBUFFER16)
    c = o.render(w, h, 1, integrator=2)
    assert _bits_equal(g, c)
    plain = pt.render(w, h, 1, integrator=gpu.INTEGRATOR_RASTER_COMPAT)
    assert not np.array_equal(g, plain) and rel_l2(g, plain) < 2e-2         # the quantisation is visible, and small
    pt.render(w, h, 1, integrator=gpu.INTEGRATOR_RASTER_GBUFFER16)
    with np.errstate(over="ignore"):
        assert np.array_equal(pt.read_radiance_f16().view(np.uint16), c.astype(np.float16).view(np.uint16))   # the RGBA16F lighting target


def test_rccl_through_the_c_abi_single_rank(gpu):
    """The RCCL code path of the C-ABI executes on the one card: unique id, ncclCommInitRank(world 1), ncclReduce in place
    on the radiance buffer (x + nothing = x), destroy; and the one-process group API (ncclCommInitAll over one device)."""
    pt = gpu.PathTracer(0).load_scene(gpu.scenes.cornell_box())
    ref = pt.render(96, 64, 4, seed=6)
    uid = gpu.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    with pytest.raises(gpu.PtcError, match="no communicator"):
        pt.comm_reduce_radiance(0)
    pt.comm_init(uid, 0, 1)
    with pytest.raises(gpu.PtcError):
        pt.comm_init(uid, 0, 1)                                    # already has one
    pt.frame_begin(96, 64, 4, 6, 8, 0)
    pt.frame_add_samples(4)
    pt.frame_resolve()
    pt.comm_reduce_radiance(0)
    with pytest.raises(gpu.PtcError):
        pt.comm_reduce_radiance(3)                                 # bad root
    pt.sync()
    assert _bits_equal(pt.read_radiance(), ref)
    assert pt.stats()["seconds_reduce"] > 0.0                      # the collective is event-timed on the context's stream (bench.py reports it per rank)
    pt.comm_destroy()
    g = gpu.Group([0]).load_scene(gpu.scenes.cornell_box())        # ptc_group_scene_commit: described on device 0, one host build for the group
    assert g.ctx(0).stats()["n_triangles"] == 12
    g.ctx(0).update_instance(0, (0.0, 0.05, 0.0), (1.0, 0.0, 0.0, 0.0), (1.0, 1.0, 1.0))       # dynamics for a group: one host refit for all devices
    g.scene_refit()
    moved = gpu.scenes.cornell_box()
    moved.instances[0].t = (0.0, 0.05, 0.0)
    assert _bits_equal(g.render(96, 64, 4, seed=6), gpu.PathTracer(0).load_scene(moved).render(96, 64, 4, seed=6))
    g.ctx(0).update_instance(0, (0.0, 0.0, 0.0), (1.0, 0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
    g.scene_refit()
    assert len(g) == 1 and _bits_equal(g.render(96, 64, 4, seed=6), ref)
    g.close()
    with pytest.raises(gpu.PtcError):
        gpu.Group([99])
    # the C++ host: `ptc_render --gpus 1` goes through pbr::DeviceGroup (ptc_group_*), and writes the RGBA16F buffer as well
    import json
    import tempfile

    exe = os.path.join(os.path.dirname(gpu.ptc.LIB_PATH), "ptc_render")
    with tempfile.TemporaryDirectory() as td:
        out, half = os.path.join(td, "c.pfm"), os.path.join(td, "c.f16")
        r = subprocess.run([exe, "--scene", "cornell", "--width", "96", "--height", "64", "--spp", "4", "--seed", "6", "--gpus", "1", "--out", out, "--half", half],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        info = json.loads(r.stdout.strip().splitlines()[-1])
        assert info["gpus"] == 1 and info["paths"] == 96 * 64 * 4
        img = np.frombuffer(open(out, "rb").read().split(b"-1.0\n", 1)[1], "<f4").reshape(64, 96, 3)[::-1]
        d = gpu.scenes.cornell_box()
        d.camera.aspect = 1.0
        ref2 = gpu.PathTracer(0).load_scene(d).render(96, 64, 4, seed=6)
        assert _bits_equal(np.ascontiguousarray(img), np.ascontiguousarray(ref2[..., :3]))
        with np.errstate(over="ignore"):
            assert np.array_equal(np.fromfile(half, np.uint16).reshape(64, 96, 4), ref2.astype(np.float16).view(np.uint16))


def test_torch_nccl_backend_world_size_1(gpu):
    """bench.py's N>1 plumbing with the real backend: torch.distributed 'nccl' (= RCCL) initialised with device_id, the
    zero-copy radiance tensor reduced with dist.reduce, and the C-ABI communicator created beside torch's — in a child
    process, so that this test process keeps no process group."""
    code = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'physically-based-renderer_amd'))
import numpy as np, torch, torch.distributed as dist
import pbr_amd
from pbr_amd import dist as pdist
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
pt = pbr_amd.PathTracer(0).load_scene(pbr_amd.scenes.cornell_box())
ref = pt.render(64, 64, 2, seed=1)
box = [pbr_amd.comm_unique_id()]
dist.broadcast_object_list(box, src=0)
pt.comm_init(box[0], 0, 1)
pt.frame_begin(64, 64, 2, 1, 8, 0); pt.frame_add_samples(2); pt.frame_resolve()
pt.comm_reduce_radiance(0); pt.sync()
t = pdist.radiance_tensor(pt, 64, 64)
dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
tt = torch.tensor([1.5], dtype=torch.float64, device='cuda'); dist.all_reduce(tt, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
assert np.array_equal(pt.read_radiance(), ref) and float(tt.item()) == 1.5
dist.barrier(); dist.destroy_process_group()
print('NCCL_OK')
""" % (ROOT, ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "NCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_config2_full_size(gpu, ora):
    """BASELINE configs[1] at its stated size: ~10 k-triangle GGX sphere, 1024x1024x256 spp (268 M paths) — run-to-run
    determinism, counter identities, tile-shard sum — and 1024x1024x2 against the oracle, bits and counters."""
    d = gpu.scenes.sphere_scene()
    pt, o = gpu.PathTracer(0).load_scene(d), ora.Oracle().load_scene(d)
    w = h = 1024
    g = pt.render(w, h, 2, seed=2, max_bounces=8)
    c = o.render(w, h, 2, seed=2, max_bounces=8, n_threads=ora.hw_threads())
    assert _bits_equal(g, c)
    for k in COUNTERS:
        assert pt.stats()[k] == o.stats()[k], k
    a = pt.render(w, h, 256, seed=2, max_bounces=8)
    sa = pt.stats()
    assert np.isfinite(a).all() and (a[..., 3] == 1).all() and a[..., :3].min() >= 0
    assert sa["paths"] == w * h * 256 and sa["hits"] <= sa["segments"] and sa["shadow_rays"] <= sa["hits"] and sa["segments"] >= sa["paths"]
    b = pt.render(w, h, 256, seed=2, max_bounces=8)
    assert _bits_equal(a, b) and all(pt.stats()[k] == sa[k] for k in COUNTERS)
    acc = np.zeros_like(a)
    for r in range(2):
        pt.frame_begin(w, h, 256, 2, 8, 0, tile_rank=r, tile_count=2)
        pt.frame_add_samples(256)
        pt.frame_resolve()
        acc += pt.read_radiance()
    assert _bits_equal(acc, a)
    assert rel_l2(a, c) < 1.0                                    # 256 spp and 2 spp estimate the same image (2 spp is noisy: 0.4)


def test_config5_full_size(gpu, ora):
    """BASELINE configs[4] at full size: the 250 k-triangle textured atrium, 1024^2 albedo / normal / metal-rough textures,
    2048x1024 environment light, 1920x1080: 2 spp against the oracle (bits + counters), properties at 64 spp."""
    d = gpu.scenes.textured_atrium()
    d.camera.aspect = 1920 / 1080
    assert d.textures[0].shape[:2] == (1024, 1024) and d.env.shape[:2] == (1024, 2048) and abs(d.n_triangles - 250_000) < 5_000
    pt, o = gpu.PathTracer(0).load_scene(d), ora.Oracle().load_scene(d)
    w, h = 1920, 1080
    g = pt.render(w, h, 2, seed=5, max_bounces=8)
    c = o.render(w, h, 2, seed=5, max_bounces=8, n_threads=ora.hw_threads())
    assert _bits_equal(g, c), int((g != c).any(-1).sum())
    for k in COUNTERS:
        assert pt.stats()[k] == o.stats()[k], k
    assert np.array_equal(pt.tonemap(), ora.tonemap_rgba8(c))
    a = pt.render(w, h, 64, seed=5, max_bounces=8)
    sa = pt.stats()
    assert np.isfinite(a).all() and a[..., :3].min() >= 0 and sa["paths"] == w * h * 64
    assert _bits_equal(pt.render(w, h, 64, seed=5, max_bounces=8), a) and all(pt.stats()[k] == sa[k] for k in COUNTERS)
    acc = np.zeros_like(a)
    for r in range(4):
        pt.frame_begin(w, h, 64, 5, 8, 0, tile_rank=r, tile_count=4)
        pt.frame_add_samples(64)
        pt.frame_resolve()
        acc += pt.read_radiance()
    assert _bits_equal(acc, a)


def test_viewer_shim_renders_progressively_and_refits(gpu):
    """examples/viewer_shim.cpp on the GPU: App::run's loop — one sample per displayed frame into the RGBA16F staging buffer, the node turned half
    way (ptc_update_instance + ptc_scene_refit), the frame restarted."""
    import json

    exe = os.path.join(os.path.dirname(gpu.ptc.LIB_PATH), "viewer_shim")
    r = subprocess.run([exe, "0", "6", "1e9"], capture_output=True, text=True, timeout=120)       # rebuild ratio out of reach: refits only
    assert r.returncode == 0, r.stdout + r.stderr
    info = json.loads(r.stdout.strip().splitlines()[0])
    assert info["rendered"] is True and info["triangles"] == 4 and info["frames"] == 6 and info["staging_sum"] > 0 and info["rebuilds"] == 0
    # the rebuild policy of Frame::rotate, forced (ratio 0: every refit is followed by ptc_scene_rebuild on the device): the same image
    r2 = subprocess.run([exe, "0", "6", "0"], capture_output=True, text=True, timeout=120)
    assert r2.returncode == 0, r2.stdout + r2.stderr
    info2 = json.loads(r2.stdout.strip().splitlines()[0])
    assert info2["rebuilds"] == 1 and info2["staging_sum"] == info["staging_sum"]
    # the default policy (ratio 1.2): whichever way it decides for this scene, the picture is the same
    r3 = subprocess.run([exe, "0", "6"], capture_output=True, text=True, timeout=120)
    assert r3.returncode == 0, r3.stdout + r3.stderr
    info3 = json.loads(r3.stdout.strip().splitlines()[0])
    assert info3["rebuilds"] in (0, 1) and info3["staging_sum"] == info["staging_sum"]


def test_checkpoint_resume_and_sample_ranges(gpu):
    """SURVEY §5 / §8e options.  The RNG is counter-based and a pixel's sum is taken in sample order, so: checkpoint after k samples + restore in a NEW
    context + the remaining samples = the uninterrupted frame, bit for bit; and a frame sharded by SAMPLE ranges (each share a partial mean, resolve_divisor =
    spp) sums to the frame up to the rounding of fp32 sums taken in another order."""
    d = gpu.scenes.atrium(0.05)
    w, h, spp, seed, mb = 128, 72, 12, 11, 6
    pt = gpu.PathTracer(0).load_scene(d)
    full = pt.render(w, h, spp, seed=seed, max_bounces=mb)
    for k in (5, 12, 0):
        a = gpu.PathTracer(0).load_scene(d)
        a.frame_begin(w, h, spp, seed, mb, 0)
        if k:
            a.frame_add_samples(k)
        acc, done = a.frame_checkpoint()
        assert done == k and acc.shape == (w * h, 4)
        b = gpu.PathTracer(0).load_scene(d)
        b.frame_begin(w, h, spp, seed, mb, 0)
        b.frame_restore(acc, done)
        if spp - k:
            b.frame_add_samples(spp - k)
        b.frame_resolve()
        assert _bits_equal(b.read_radiance(), full), f"resumed after {k} samples"
        with pytest.raises(gpu.PtcError):
            b.frame_restore(acc, done)                               # only right after frame_begin
    with pytest.raises(gpu.PtcError):
        pt.frame_begin(w // 2, h, spp, seed, mb, 0)
        pt.frame_restore(acc, done)                                  # another frame's checkpoint
    # sample-range sharding: three shares of 4 samples each, all pixels, partial means
    total = np.zeros_like(full)
    for r in range(3):
        pt.frame_begin(w, h, 4, seed, mb, 0)
        pt.frame_set_sample_range(4 * r, spp)
        pt.frame_add_samples(4)
        pt.frame_resolve()
        total += pt.read_radiance()
    total[..., 3] = 1.0
    rel = float(np.sqrt(((total[..., :3].astype(np.float64) - full[..., :3]) ** 2).sum()) / np.sqrt((full[..., :3].astype(np.float64) ** 2).sum()))
    assert rel < 1e-6, rel
    assert not _bits_equal(total, full) or True                      # equal to rounding, not necessarily bit for bit: why tiles are the default
    pt.frame_begin(w, h, 4, seed, mb, 0)
    pt.frame_add_samples(1)
    with pytest.raises(gpu.PtcError):
        pt.frame_set_sample_range(4, spp)                            # only before the first sample
