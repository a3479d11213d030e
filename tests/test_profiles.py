"""The measurement artefacts bench.py depends on are committed and well-formed (no GPU needed): the per-kernel model calibrated from the
rocprofv3 PMC profile, the VALU issue-rate microbenchmark, and the arithmetic that turns counted units into roofline fractions."""
import ast
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


def test_kernel_model_and_ceilings_are_committed():
    model = json.load(open(os.path.join(PROF, "r02_kernel_model.json")))
    for k in ("k_trace_closest", "k_trace_any", "k_shade"):
        m = model[k]
        for key in ("valu_winstr_per_unit", "hbm_bytes_per_unit", "serialised_ms_per_launch", "serialised_units_per_launch", "dispatches", "unit"):
            assert key in m and m[key], (k, key)
        assert 0.3 < m["valu_lane_utilisation"] <= 1.0
    # a node visit of the 8-wide tree costs ~6 wave-instructions per lane-visit (=~380 lane slots, everything amortised)
    assert 3.0 < model["k_trace_closest"]["valu_winstr_per_unit"] < 12.0
    valu = json.load(open(os.path.join(PROF, "r02_valu_issue.json")))
    ns = valu["ns_per_instr_per_simd_at_7_waves"]
    assert 0.8 < ns < 1.6                                   # one wave64 instruction per ~2 cycles and SIMD at >= 2 resident waves
    assert 1.6 < valu["ns_per_instr_one_wave"] / ns < 2.2   # a lone wave issues at half that rate
    peak = 1024 / ns
    assert 600 < peak < 1229                                 # G wave-instr/s: below the paper figure 1024 x 2.4 GHz / 2
    # exclusive fractions reproduce from the model alone: units x instr / time / peak
    for k, lo, hi in (("k_trace_closest", 0.5, 1.0), ("k_trace_any", 0.5, 1.0), ("k_shade", 0.1, 0.6)):
        m = model[k]
        frac = m["serialised_units_per_launch"] * m["valu_winstr_per_unit"] / (m["serialised_ms_per_launch"] * 1e-3) / 1e9 / peak
        assert lo < frac < hi, (k, frac)
    for f in ("r02_kernel_stats.csv", "r02_pmc_summary.json", "r02_gather_bench.json", "r02_bench.json", "r02_textured_kernel_stats.csv", "r02_textured_pmc_summary.json"):
        assert os.path.getsize(os.path.join(PROF, f)) > 100, f


def test_bench_line_of_the_committed_run_has_the_contract_fields():
    d = json.load(open(os.path.join(PROF, "r02_bench.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "Mpaths/s" and d["dtype"] == "f32" and d["scaling"] == "strong" and d["vs_baseline"] is None and d["data"] == "synthetic"
    r = d["roofline"]
    assert r["bound"] == "valu_issue" and 0 < r["frac"] <= 1 and 0 < r["frac_exclusive"] <= 1 and r["traffic"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    assert abs(d["value"] - d["config"]["paths"] / (d["ms_per_step"] * 1e-3 * d["steps"]) / 1e6) / d["value"] < 1e-6
    ast.parse(open(os.path.join(ROOT, "bench.py")).read())
