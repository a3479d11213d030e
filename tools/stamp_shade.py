"""Where a k_shade wave spends its cycles, per phase of a 512-slot window (needs libptc built with -DPT_STAMP_SHADE)."""
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
names = ["front end (class word, sort)", "ray + hit + record loads", "arithmetic", "compaction (ballots, atomics)", "stores + barrier"]
t = [c[7], c[8], c[9], c[10], c[11]]
tot = float(sum(t))
for nm, v in zip(names, t): print("%-32s %.3f" % (nm, v / tot))
print("seconds_shade", st["seconds_shade"], "segments", st["segments"], "memtime ticks per segment-lane", tot / st["segments"])
