"""Cost of a node visit by bounce: the benchmark frame at max_bounces 0, 1, 2, 4, 8 — event time of the trace kernels per node visit, cumulative and per added bounce.
Primary rays of a wave are 64 neighbouring pixels (coherent), later bounces are not: the difference is what ray coherence is worth to these kernels (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "physically-based-renderer_amd"))
import pbr_amd
from pbr_amd import scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pt = pbr_amd.PathTracer(0).load_scene(scenes.atrium())
prev = None
for mb in (0, 1, 2, 4, 8):
    for rep in range(2):       # the second run is the measured one (queues allocated, code loaded); ptc_stats are running totals of the context
        s0 = pt.stats()
        pt.frame_begin(1920, 1080, spp, seed=3, max_bounces=mb)
        pt.frame_add_samples(spp); pt.sync()
        s1 = pt.stats()
        st = s1      # counters and seconds restart at frame_begin
    cur = dict(nc=st["node_visits_closest"], tc=st["seconds_trace_closest"], na=st["node_visits_any"], ta=st["seconds_trace_any"], seg=st["segments"], sh=st["shadow_rays"], trc=st["tri_tests_closest"], tra=st["tri_tests_any"])
    line = "max_bounces %d: closest %.2f ps per visit (%.2f visits, %.2f tests per ray, %d rays) any %.2f ps per visit (%.2f visits per ray, %d rays)" % (
        mb, 1e12 * cur["tc"] / cur["nc"], cur["nc"] / cur["seg"], cur["trc"] / cur["seg"], cur["seg"], 1e12 * cur["ta"] / max(1, cur["na"]), cur["na"] / max(1, cur["sh"]), cur["sh"])
    if prev:
        dn, dt, dseg = cur["nc"] - prev["nc"], cur["tc"] - prev["tc"], cur["seg"] - prev["seg"]
        dna, dta, dsh = cur["na"] - prev["na"], cur["ta"] - prev["ta"], cur["sh"] - prev["sh"]
        line += "\n   added bounces: closest %.2f ps per visit (%.2f visits per ray), any %.2f ps per visit (%.2f visits per ray)" % (1e12 * dt / dn, dn / dseg, 1e12 * dta / dna, dna / dsh)
    print(line, flush=True)
    prev = cur
