#!/usr/bin/env python3
"""bench.py — Mpaths/s of the wavefront path tracer on BASELINE.json's headline configuration.

Workload (config.workload): the Sponza-class procedural atrium (249,936 triangles, SURVEY §8d row 3 —
the reference's assets are stripped, so the scene is generated) at 1920×1080.  One STEP = `--spp-per-step`
samples of every pixel of the frame; the default 16 steps × 64 spp = the 1024 spp of BASELINE.json's metric.
Inputs (scene, BVH, queues) are resident in HBM before the timed region.

Contract: W untimed warm-up steps, then exactly K timed steps bracketed by barrier +
torch.cuda.synchronize() on both sides, MAX over ranks, rank 0 prints ONE JSON line.
N > 1: `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`; one process per
GPU, 32×32-pixel tiles interleaved over ranks, and the single collective of the path — the RCCL fp32
sum of the framebuffer onto rank 0 (ncclReduce through the C-ABI, ptc_comm_reduce_radiance) — is inside
the timed region.  Scaling is STRONG by default: the frame is the metric's fixed 1920×1080×1024-spp frame at
every N, a rank traces its 1/N of the pixels.  The library merges the samples of successive steps into full
wavefront batches (ptc_frame_add_samples defers a partial batch), so a rank's launches stay as wide as at N=1
(N× deeper in samples) instead of shrinking by N; everything queued is finished inside the timed region
(resolve + sync + reduce).  `--scaling weak` is the opt-in alternative: a step adds spp-per-step × N samples
(the job renders 1024·N spp).

Extra objects on the JSON line:
  roofline     the dominant kernel, k_trace_closest, against the bound that limits it: VALU issue.
               achieved = counted node visits × the VALU wave-instructions per node visit calibrated from
               the committed rocprofv3 PMC profile (profiles/r02_kernel_model.json: SQ_INSTS_VALU ÷ counted
               visits of the same workload) ÷ the kernel's HIP-event time; peak = the measured issue rate of
               an all-fma stream at 7 waves per SIMD (tools/valu_issue_bench.hip → profiles/r02_valu_issue.json).
               `traffic` = HBM-side bytes per launch from the same PMC profile; `algorithmic_*` = the
               SURVEY §8(d) byte model (counted units × record sizes), labelled as such: those bytes are mostly
               served by LDS / L2 / Infinity Cache and are NOT an HBM roofline.
               `frac` uses the live HIP-event time of a launch (with PTC_LANES > 1 that includes the time it shares the chip
               with another lane's kernels; the default is one lane); `frac_exclusive` uses the kernel's exclusive duration
               from the profile's serialised run.
  kernels      the same two rooflines (valu_issue, hbm) for k_trace_closest, k_trace_any and k_shade.
  whole_frame  VALU issue and HBM rates of the three kernels together over the timed wall time.
  cpu_baseline the oracle (scalar C restatement, kind "port") timed on this box's host cores on a
               bounded sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (the host driver has no legacy IPC: RCCL's hipIpcGetMemHandle fails without it); the driver's environment
# exports it already — set here, before anything touches HIP, for a launch from a bare shell
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "physically-based-renderer_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
N_SIMD = 1024          # 256 CUs × 4 SIMD32
# VALU issue: a wave64 instruction occupies a SIMD32 for 2 cycles → 1024 SIMDs × 2.4 GHz / 2 = 1228.8 G wave-instr/s on paper;
# measured with an all-v_fma_f32 stream at 7 waves per SIMD (profiles/r02_valu_issue.json): 1.127 ns per instruction and SIMD.
VALU_PEAK_PAPER_GIPS = N_SIMD * 2.4 / 2.0


def _load_json(path):
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spp-per-step", type=int, default=64)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--max-bounces", type=int, default=8)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--scene-scale", type=float, default=1.0, help="atrium tessellation scale (1.0 = 249,936 triangles)")
    ap.add_argument("--workload", choices=["atrium", "textured"], default="atrium",
                    help="atrium = BASELINE configs[2]/[3] (the metric's configuration); textured = configs[4]: the same atrium with 1024^2 "
                         "albedo/normal/metal-rough textures and a 2048x1024 environment light")
    ap.add_argument("--direct-scene", action="store_true", help="hand the generated arrays to the C-ABI directly instead of through a GLB file + the glTF loader")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU work of the bounded CPU-baseline render")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; 'gloo' + --share-device rehearses N ranks on a one-GPU box")
    ap.add_argument("--share-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="N > 1: strong (default) = the metric's fixed 1024-spp frame, a rank traces 1/N of the pixels; weak = every step adds spp-per-step x N samples")
    ap.add_argument("--rehearse-tiles", type=int, default=0,
                    help="N = 1 only: render tile rank 0 of this many (what ONE rank of an N-GPU strong-scaling run does, without the reduce); the JSON "
                         "line's value is then that rank's share, not the metric")
    ap.add_argument("--reduce", choices=["cabi", "torch"], default="cabi",
                    help="N > 1: cabi = ncclReduce through the C-ABI (ptc_comm_*), the id shipped over torch.distributed; torch = torch.distributed.reduce on the zero-copy tensor")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import pbr_amd
    from pbr_amd import dist as pdist
    from pbr_amd import scenes

    rank, world, local_rank = pdist.env_rank_world()
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run); got {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the path tracer is HIP-only (no CPU fallback)")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    desc = scenes.atrium(args.scene_scale) if args.workload == "atrium" else scenes.textured_atrium(args.scene_scale)
    desc.camera.aspect = args.width / args.height
    # The metric is quoted on a glTF scene: the generated scene is written as a binary glTF and comes in through the library's own
    # loader (ptc_gltf_load: JSON, accessors, node hierarchy, PNG textures) — bit-identical to handing the arrays over directly
    # (tests/test_gltf.py, test_gpu_parity.py::test_gltf_loaded_scene_renders_identically).
    pt = pbr_amd.PathTracer(local_rank)
    scene_source = "generated arrays handed to the C-ABI"
    if args.direct_scene:
        pt.load_scene(desc)
    else:
        import tempfile

        from pbr_amd import gltf

        # rank 0 writes the file once and ships its BYTES (2.9 MB): every rank (one process per GPU, possibly on another node's file system) stores them in a
        # temporary file of its own and loads that through the library's own loader
        box = [None]
        td = tempfile.TemporaryDirectory()
        glb = os.path.join(td.name, "bench_scene.glb")
        if rank == 0:
            gltf.write_glb(desc, glb)
            box[0] = open(glb, "rb").read()
        if world > 1:
            dist.broadcast_object_list(box, src=0)
            if rank != 0:
                with open(glb, "wb") as f:
                    f.write(box[0])
        n_tri, _, _ = gltf.load_into(pt, glb, camera=desc.camera, env=getattr(desc, "env", None))
        assert n_tri == desc.n_triangles, (n_tri, desc.n_triangles)
        scene_source = f"binary glTF ({os.path.getsize(glb) / 1e6:.1f} MB) written from the generator by rank 0, loaded by ptc_gltf_load"
        td.cleanup()
    # scene dynamics: what a transform change costs (ptc_scene_refit of the unchanged scene re-computes and re-uploads exactly what a real one does)
    pt.scene_refit()                                       # the first one also builds and uploads the refit plan
    pt.scene_refit()
    refit_seconds = pt.stats()["seconds_refit"]
    os.environ["PTC_REFIT"] = "host"                       # the same refit computed on the host and uploaded (the cross-check path)
    try:
        pt.scene_refit()
    finally:
        del os.environ["PTC_REFIT"]
    refit_host_seconds = pt.stats()["seconds_refit"]
    pt.scene_refit()                                       # leave the device-refitted scene (the same bytes) in place
    K, W, S = args.steps, args.warmup, args.spp_per_step * (world if args.scaling == "weak" else 1)
    spp_total = (K + W) * S
    tile_count = world if world > 1 else max(1, args.rehearse_tiles)
    pt.frame_begin(args.width, args.height, spp_total, args.seed, args.max_bounces, pbr_amd.INTEGRATOR_PATH, tile_rank=rank, tile_count=tile_count)
    pt.frame_reserve()          # queues for full batches now: no allocation (and no drain of the device for one) inside the timed region
    reduce_impl = None
    if world > 1:
        reduce_impl = "torch.distributed.reduce"
        if args.reduce == "cabi" and args.backend == "nccl" and not args.share_device:     # RCCL needs one GPU per rank
            # Every rank first proves, locally, that the library can load librccl (ptc_comm_unique_id is not a collective), and the
            # ranks agree on that BEFORE anyone enters the collective ncclCommInitRank — a rank that cannot join must not leave the
            # others waiting in it.  Then rank 0's id is shipped over torch.distributed and every rank joins the communicator.
            ok, uid = 1, None
            try:
                uid = pbr_amd.comm_unique_id()
            except pbr_amd.PtcError as e:
                print(f"bench: rank {rank}: ptc_comm_unique_id failed ({e}); falling back to torch.distributed.reduce", file=sys.stderr)
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                box = [uid if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                try:
                    pt.comm_init(box[0], rank, world)
                    joined = 1
                except pbr_amd.PtcError as e:
                    print(f"bench: rank {rank}: ptc_comm_init failed ({e})", file=sys.stderr)
                    joined = 0
                flag = torch.tensor([joined], dtype=torch.int32, device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)          # every rank uses the same reduce
                if int(flag.item()) == 1:
                    reduce_impl = "ncclReduce via C-ABI (ptc_comm_reduce_radiance)"

    def barrier():
        if world > 1:
            dist.barrier()

    def reduce_to_root():
        if world > 1 and reduce_impl.startswith("ncclReduce"):
            pt.comm_reduce_radiance(0)           # queued on the context's stream behind the resolve
            pt.sync()
        else:
            pt.sync()
            if world > 1:
                fb = pdist.radiance_tensor(pt, args.width, args.height)
                pdist.reduce_framebuffer(fb, 0)

    for _ in range(W):
        pt.frame_add_samples(S)
    if world > 1 and W > 0:                      # the warm-up includes one resolve + reduce: the first collective sets up the links
        pt.frame_resolve()
        reduce_to_root()
    pt.sync()
    s0 = pt.stats()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        pt.frame_add_samples(S)
    pt.frame_resolve()
    reduce_to_root()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    s1 = pt.stats()
    dt = t1 - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    d = {k: s1[k] - s0[k] for k in s1 if isinstance(s1[k], (int, float))}
    # per-rank counters → whole-job sums (strong scaling: ranks own disjoint pixels)
    keys = ["paths", "segments", "shadow_rays", "hits", "node_visits_closest", "tri_tests_closest", "node_visits_any", "tri_tests_any", "algorithmic_bytes"]
    tot = {k: float(d[k]) for k in keys}
    if world > 1:
        v = torch.tensor([tot[k] for k in keys], dtype=torch.float64, device="cuda")
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        tot = {k: float(x) for k, x in zip(keys, v.tolist())}

    # PTC_TRACE_OVERLAP: 2 = every batch, 1 = batches of <= 2^26 paths: closest(b + 1) and any(b) then run beside each other and their event times include each other
    ov = int(os.environ.get("PTC_TRACE_OVERLAP", "1"))
    batch_samples = min(pt.internals()["per_batch"], K * S)                   # the deferred batching issues full batches, and what is left at the resolve as one more
    kernel_overlap = bool(ov == 2 or (ov == 1 and batch_samples * (args.width * args.height // max(1, tile_count)) <= (1 << 26)))
    # per-rank times, so that a poor scaling point can be attributed: tile imbalance (spread of the kernel sums), the collective
    # (reduce_ms: the collective as each rank's stream sees it, i.e. including the wait for the slowest rank) or launch tails
    per_rank = None
    if world > 1:
        mine = torch.tensor([d["seconds_trace_closest"], d["seconds_trace_any"], d["seconds_shade"], t1 - t0, d.get("seconds_reduce", 0.0), float(d["paths"])],
                            dtype=torch.float64, device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rows = [[float(x) for x in r.tolist()] for r in allr]
        names = ["seconds_trace_closest", "seconds_trace_any", "seconds_shade", "wall", "seconds_reduce", "paths"]
        per_rank = {n: {"min": min(r[i] for r in rows), "max": max(r[i] for r in rows), "ranks": [r[i] for r in rows]} for i, n in enumerate(names)}
        per_rank["reduce_ms"] = {"min": per_rank["seconds_reduce"]["min"] * 1e3, "max": per_rank["seconds_reduce"]["max"] * 1e3}
        per_rank["reduce_impl"] = reduce_impl
        per_rank["batches_per_rank"] = d["launches_trace_closest"] / (args.max_bounces + 1)        # wavefront batches this rank issued in the timed region (rank 0's count)
        per_rank["kernel_seconds_overlap"] = kernel_overlap

    if rank == 0:
        paths = args.width * args.height * S * K
        if tile_count == world:
            assert abs(tot["paths"] - paths) < 0.5, (tot["paths"], paths)
        else:
            paths = int(tot["paths"])       # a rehearsal of one rank's share
        # Rooflines of the three kernels that carry the frame, on this rank.  Units are counted by the kernels; the
        # per-unit instruction and HBM-byte figures come from the committed PMC profile of the same workload.
        model_name = "kernel_model.json" if args.workload == "atrium" else "textured_kernel_model.json"
        model_path = next((p for p in (os.path.join(ROOT, "profiles", f"r{r:02d}_{model_name}") for r in (4, 3, 2)) if os.path.exists(p)), None)
        model = (_load_json(model_path) if model_path else None) or {}
        # The per-unit figures are only as good as the kernels they were measured on: the library carries the sha256 of the kernel sources
        # it was built from (ptc_build_info), the model the sha256 of the sources that were profiled.  A mismatch is reported, not hidden.
        build_info = pbr_amd.load_library().ptc_build_info().decode()
        lib_sha = build_info.rsplit(" ", 1)[-1]
        # ... and on the launch policy they were measured under (grid sizing, LDS split, batch size, overlap mode: ptc_launch_policy)
        policy, policy_defaults = pt.launch_policy(), pbr_amd.load_library().ptc_launch_policy(None).decode()
        stale_why = []
        if not model:
            stale_why.append("no committed kernel model")
        else:
            if model.get("kernel_source_sha256") != lib_sha:
                stale_why.append("kernel sources differ from the profiled ones")
            if model.get("launch_policy") != policy:
                stale_why.append("launch policy differs from the profiled run's")
        model_stale = bool(stale_why)
        valu = _load_json(os.path.join(ROOT, "profiles", "r02_valu_issue.json")) or {}
        ns_per_instr = valu.get("ns_per_instr_per_simd_at_7_waves")
        valu_peak_measured = N_SIMD / ns_per_instr if ns_per_instr else None          # G wave-instr/s
        units = {"k_trace_closest": ("node_visits_closest", d["seconds_trace_closest"], max(1, d["launches_trace_closest"])),
                 "k_trace_any": ("node_visits_any", d["seconds_trace_any"], max(1, d["launches_trace_any"])),
                 "k_shade": ("segments", d["seconds_shade"], max(1, d["launches_trace_closest"]))}
        kernels = {}
        for kname, (ukey, sec, nl) in units.items():
            m = model.get(kname, {})
            n_units = float(d[ukey])
            e = {"unit_counted": ukey, "units_per_launch": n_units / nl, "avg_launch_ms": sec / nl * 1e3, "launches": nl}
            if m.get("valu_winstr_per_unit") and sec > 0:
                gips = n_units * m["valu_winstr_per_unit"] / sec / 1e9
                e["valu_issue"] = {"achieved": gips, "unit": "G wave-instr/s", "peak": VALU_PEAK_PAPER_GIPS,
                                   "peak_kind": "MI355X_MICROARCH.md: 1024 SIMD x 2.4 GHz / 2 cycles per wave64 instruction",
                                   "frac": gips / VALU_PEAK_PAPER_GIPS,
                                   "measured_ceiling": valu_peak_measured, "measured_ceiling_kind": "all-v_fma_f32 stream, 7 waves/SIMD (clocks down under power: profiles/r02_valu_issue.json)",
                                   "frac_of_measured_ceiling": gips / valu_peak_measured if valu_peak_measured else None,
                                   # issue slots used per SIMD cycle in the profiled run: SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)
                                   "issue_duty_per_cycle": m.get("valu_issue_duty_per_cycle")}
            if m.get("hbm_bytes_per_unit") and sec > 0:
                gbs = n_units * m["hbm_bytes_per_unit"] / sec / 1e9
                e["hbm"] = {"achieved": gbs, "unit": "GB/s", "peak": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS,
                            "bytes_per_launch": n_units * m["hbm_bytes_per_unit"] / nl}
            # With PTC_LANES > 1 the live event time of a launch includes the time it shares the chip with another lane's kernels;
            # the profile's serialised run gives the kernel's exclusive duration for the same units per launch.
            if m.get("serialised_ms_per_launch") and m.get("serialised_units_per_launch"):
                excl_ms = m["serialised_ms_per_launch"] * (n_units / nl) / m["serialised_units_per_launch"]
                e["exclusive_launch_ms_from_profile"] = excl_ms
                if "valu_issue" in e:
                    e["valu_issue"]["frac_exclusive"] = n_units / nl * m["valu_winstr_per_unit"] / (excl_ms * 1e-3) / 1e9 / VALU_PEAK_PAPER_GIPS
                if "hbm" in e:
                    e["hbm"]["frac_exclusive"] = n_units / nl * m["hbm_bytes_per_unit"] / (excl_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            kernels[kname] = e
        whole = None
        if all("valu_issue" in kernels[k] and "hbm" in kernels[k] for k in kernels):
            winstr = sum(float(d[units[k][0]]) * model[k]["valu_winstr_per_unit"] for k in kernels)
            hbm_b = sum(float(d[units[k][0]]) * model[k]["hbm_bytes_per_unit"] for k in kernels)
            pk = VALU_PEAK_PAPER_GIPS
            whole = {"note": "the three kernels together over the timed wall time; fractions against the guide's peaks (1228.8 G wave-instr/s, 8 TB/s)",
                     "valu_issue_G_winstr_s": winstr / dt / 1e9, "valu_issue_frac": winstr / dt / 1e9 / pk,
                     "hbm_GBs": hbm_b / dt / 1e9, "hbm_frac": hbm_b / dt / 1e9 / HBM_PEAK_GBS}
        n_launch = max(1, d["launches_trace_closest"])
        tc_bytes = d["segments"] * (56 + 16) + d["node_visits_closest"] * 64 + d["tri_tests_closest"] * 48
        tc_sec = d["seconds_trace_closest"]
        tc = kernels["k_trace_closest"]
        if "valu_issue" in tc:
            roof = {"bound": "valu_issue", "kernel": "k_trace_closest", "achieved": tc["valu_issue"]["achieved"], "peak": tc["valu_issue"]["peak"],
                    "unit": "G wave-instr/s", "frac": tc["valu_issue"]["frac"], "frac_of_measured_ceiling": tc["valu_issue"]["frac_of_measured_ceiling"],
                    "measured_ceiling": tc["valu_issue"]["measured_ceiling"], "issue_duty_per_cycle": tc["valu_issue"]["issue_duty_per_cycle"],
                    "frac_exclusive": tc["valu_issue"].get("frac_exclusive"), "exclusive_launch_ms_from_profile": tc.get("exclusive_launch_ms_from_profile"),
                    "peak_kind": tc["valu_issue"]["peak_kind"], "traffic": tc.get("hbm", {}).get("bytes_per_launch"),
                    "hbm_achieved_GBs": tc.get("hbm", {}).get("achieved"), "hbm_frac": tc.get("hbm", {}).get("frac")}
        else:   # no committed profile to calibrate from: report the physical side as unknown rather than a byte model as a bound
            roof = {"bound": "valu_issue", "kernel": "k_trace_closest", "achieved": None, "peak": VALU_PEAK_PAPER_GIPS, "unit": "G wave-instr/s", "frac": None, "traffic": None}
        roof.update({"model": os.path.relpath(model_path, ROOT) if model_path else None, "model_commit": model.get("git_commit"),
                     "model_kernel_sha256": model.get("kernel_source_sha256"), "library_kernel_sha256": lib_sha, "model_stale": model_stale, "model_stale_why": stale_why,
                     "model_launch_policy": model.get("launch_policy"),
                     "avg_launch_ms": tc_sec / n_launch * 1e3, "launches": n_launch,
                     "algorithmic_bytes_per_launch": tc_bytes / n_launch,
                     "algorithmic_GBs": tc_bytes / tc_sec / 1e9 if tc_sec > 0 else 0.0,
                     "algorithmic_note": "SURVEY 8(d) byte model (counted units x record sizes); served mostly by LDS/L2/Infinity Cache, not an HBM bound"})
        out = {
            "metric": "Mpaths/s + HBM GB/s (% of peak), 1920x1080x1024 spp glTF scene",
            "value": paths / dt / 1e6,
            "unit": "Mpaths/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": dt / K * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"procedural atrium {desc.n_triangles} tris (BASELINE configs[2]/[3])" if args.workload == "atrium" else
                             f"textured atrium {desc.n_triangles} tris + 1024^2 albedo/normal/metal-rough textures + 2048x1024 env light (BASELINE configs[4])")
                            + f", {args.width}x{args.height}, {S * K} spp = {K} steps x {S} spp, max_bounces {args.max_bounces}, seed {args.seed}",
                "paths": paths,
                "batching": "%d lane(s), %d samples per full wavefront batch (%d paths in flight)"
                            % (int(os.environ.get("PTC_LANES", "1")), pt.internals()["per_batch"], pt.internals()["per_batch"] * (pt.internals()["queue_cap"] // max(1, pt.internals()["per_batch"]))),
                "scene_source": scene_source,
                "sharding": (f"32x32 tiles over {world} ranks (each rank: 1/{world} of the pixels x {S} spp per step), {reduce_impl} to rank 0"
                             if world > 1 else "single GPU"),
            },
            "library": build_info,
            "launch_policy": policy,
            "launch_policy_defaults": policy_defaults,
            "roofline": roof,
            "kernels": kernels,
            "whole_frame": whole,
            "per_rank": per_rank,
            "rehearsal": (f"tile rank 0 of {tile_count} on one GPU: value is that rank's share of the frame, not the metric" if tile_count != world else None),
            "whole_path_algorithmic_GBs": tot["algorithmic_bytes"] / dt / 1e9,
            "scene_commit_seconds": s1.get("seconds_commit"),    # flatten + SAH BVH + upload, once per scene, outside the timed region
            "scene_refit_seconds": refit_seconds,                # ptc_scene_refit on the device (csrc/pt_refit.hip): flatten + shading records + nodes, in place in HBM
            "scene_refit_host_seconds": refit_host_seconds,      # PTC_REFIT=host: the same on the host's thread pool + upload
            "seconds": {"wall": dt, "trace_closest": d["seconds_trace_closest"], "trace_any": d["seconds_trace_any"], "shade": d["seconds_shade"],
                        "batches": d.get("seconds_render", 0.0), "kernel_times_overlap": kernel_overlap},
            "per_path": {"segments": tot["segments"] / paths, "shadow_rays": tot["shadow_rays"] / paths,
                         "node_visits": (tot["node_visits_closest"] + tot["node_visits_any"]) / paths,
                         "algorithmic_bytes": tot["algorithmic_bytes"] / paths},
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import ora  # the checker, timed as the CPU baseline; never on the product path

            o = ora.Oracle().load_scene(desc)
            cores = ora.hw_threads()
            tc0 = time.perf_counter()
            o.render(args.width, args.height, 1, args.seed, args.max_bounces, 0, 0, 1, cores)       # probe: 1 spp
            probe = time.perf_counter() - tc0
            cpu_spp = max(1, min(64, int(args.cpu_seconds / max(probe, 1e-3))))
            tc0 = time.perf_counter()
            o.render(args.width, args.height, cpu_spp, args.seed, args.max_bounces, 0, 0, 1, cores)
            tc = time.perf_counter() - tc0
            out["cpu_baseline"] = {
                "value": args.width * args.height * cpu_spp / tc / 1e6,
                "unit": "Mpaths/s",
                "cores": cores,
                "kind": "port",
                "sample": f"same scene and camera, {args.width}x{args.height} x {cpu_spp} spp ({tc:.1f} s of CPU work), oracle/ptc_oracle.c -O2 -ffp-contract=off, pthreads over rows, all online cores",
            }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
