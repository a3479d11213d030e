"""Where a k_trace_closest wave spends its cycles (needs libptc built with EXTRA=-DPT_STAMP)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "physically-based-renderer_amd"))
import pbr_amd
from pbr_amd import scenes
pt = pbr_amd.PathTracer(0).load_scene(scenes.atrium())
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
pt.frame_begin(1920, 1080, spp, seed=3, max_bounces=8)
pt.frame_add_samples(spp); pt.sync()
st = pt.stats(); c = pt.raw_counters()
node, leaf, fin, refill = c[7], c[8], c[9], c[10]
tot = node + leaf + fin + refill
print("wave-cycles: node loop %.3f  leaf phase %.3f  publish %.3f  refill+outer %.3f  (total %.3e)" % (node / tot, leaf / tot, fin / tot, refill / tot, tot))
print("seconds_trace_closest", st["seconds_trace_closest"], "segments", st["segments"], "cycles per ray-lane", tot / st["segments"])
