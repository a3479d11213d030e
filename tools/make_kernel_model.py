#!/usr/bin/env python3
"""profiles/r02_kernel_model.json from one tools/profile.sh run: per kernel, the VALU wave-instructions and HBM-side bytes per counted
unit (node visit for the trace kernels, shaded segment for k_shade).  bench.py multiplies these by the units it counts live.

usage: tools/make_kernel_model.py gpurun_out/prof_TAG profiles/r02_kernel_model.json
FETCH_SIZE is in KiB and, on gfx950, tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md §HBM): it is doubled here; the
correction is calibrated for wide streaming reads, not for 16-byte gathers, so the true read traffic of the trace kernels lies
between 0.5x and 1x of `fetch_bytes` (stated in the output)."""
import hashlib, json, os, re, subprocess, sys

prof, out = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "physically-based-renderer_amd", "csrc")


def kernel_source_sha256():
    """The recipe of csrc/Makefile's KERNEL_SHA: sha256 over its KERNEL_SRC list, in that order (every file that holds device code)."""
    h = hashlib.sha256()
    for f in "pt_kernels.hip pt_device.h ptc_internal.h pt_refit.hip pt_refit.h pt_build.hip pt_build.h".split():
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()


pmc = json.load(open(os.path.join(prof, "pmc_summary.json")))
bench = None
for line in open(os.path.join(prof, "pmc1.log")):
    if line.startswith("{") and '"metric"' in line:
        bench = json.loads(line)
assert bench, "no bench line in pmc1.log"
lane1 = json.load(open(os.path.join(prof, "bench_lane1.json")))      # PTC_LANES=1: kernels do not overlap, event times are exclusive
K, W = bench["steps"], bench["warmup"]
scale = (K + W) / K                                   # the counters cover warm-up + timed steps, the bench line the timed steps
units = {"k_trace_closest": bench["kernels"]["k_trace_closest"], "k_trace_any": bench["kernels"]["k_trace_any"], "k_shade": bench["kernels"]["k_shade"]}
profiled_sha = bench.get("library", "").rsplit(" ", 1)[-1]                 # the library that ran under the profiler says what it was built from
assert profiled_sha == kernel_source_sha256(), f"the profile was taken on kernels {profiled_sha[:12]}, the tree holds {kernel_source_sha256()[:12]}: profile again"
commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", CSRC], capture_output=True, text=True).stdout.strip()
assert bench.get("launch_policy") and bench.get("launch_policy_defaults"), "the bench line carries no launch policy: old bench.py?"
model = {"git_commit": commit + ("+uncommitted csrc changes" if dirty else ""), "kernel_source_sha256": profiled_sha,
         # what the launches of the profiled run looked like (ptc_launch_policy): grid sizing, LDS split, batch size, overlap mode ... — bench.py reports
         # model_stale when the library it runs launches differently, tests/test_profiles.py when the library's DEFAULTS have moved since
         "launch_policy": bench["launch_policy"], "launch_policy_defaults": bench["launch_policy_defaults"],
         "source": f"{prof}: rocprofv3 --pmc passes of `bench.py --steps {K} --warmup {W}` ({bench['config']['workload']})",
         "fetch_note": "fetch bytes = FETCH_SIZE KiB x 1024 x 2 (gfx950 tallies 128-B requests at 64 B; calibrated for streaming reads, uncalibrated for 16-B gathers: true value between 0.5x and 1x)"}
for kname, u in units.items():
    key = [k for k in pmc if kname in k and "<true" not in k]
    key = max(key, key=lambda k: pmc[k].get("dispatches", 0))
    p = pmc[key]
    n_units = u["units_per_launch"] * u["launches"] * scale
    fetch = p.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0
    write = p.get("WRITE_SIZE", 0.0) * 1024.0
    model[kname] = {
        "pmc_kernel": key, "dispatches": p.get("dispatches"), "unit": u["unit_counted"], "units_in_profile": n_units,
        "valu_winstr_per_unit": p["SQ_INSTS_VALU"] / n_units, "salu_instr_per_unit": p.get("SQ_INSTS_SALU", 0) / n_units,
        "vmem_rd_instr_per_unit": p.get("SQ_INSTS_VMEM_RD", 0) / n_units, "lds_instr_per_unit": p.get("SQ_INSTS_LDS", 0) / n_units,
        "hbm_bytes_per_unit": (fetch + write) / n_units, "fetch_bytes_per_unit": fetch / n_units, "write_bytes_per_unit": write / n_units,
        "valu_lane_utilisation": p.get("_valu_lane_util"), "wave_cycles_waiting_memory": p.get("_wait_any_frac"), "wave_cycles_waiting_issue": p.get("_wait_inst_frac"),
        "l2_hit_rate": (p["TCC_HIT_sum"] / (p["TCC_HIT_sum"] + p["TCC_MISS_sum"])) if p.get("TCC_HIT_sum") else None,
        # exclusive duration of a launch: HIP-event time in the one-lane run of the same workload (units per launch are the same there)
        "serialised_ms_per_launch": lane1["kernels"][kname]["avg_launch_ms"],
        "serialised_units_per_launch": lane1["kernels"][kname]["units_per_launch"],
        "gui_active_cycles_per_launch": p.get("GRBM_GUI_ACTIVE", 0) / 8.0 / p.get("dispatches", 1) if p.get("GRBM_GUI_ACTIVE") else None,
        # VALU issue slots used per SIMD cycle: a wave64 instruction takes 2 cycles of a SIMD32; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        "valu_issue_duty_per_cycle": (p["SQ_INSTS_VALU"] * 2.0 / (1024.0 * p["GRBM_GUI_ACTIVE"] / 8.0)) if p.get("GRBM_GUI_ACTIVE") else None,
    }
json.dump(model, open(out, "w"), indent=1)
print(json.dumps(model, indent=1))
