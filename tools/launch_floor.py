"""Fixed cost of a launch of each kernel: event time per launch when the queues hold next to nothing (a 32x32x1 frame) and at a few sizes."""
import os
os.environ.setdefault("PTC_TIMING", "2")      # a span per kernel also where a batch's trace kernels run beside each other (small frames)
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "physically-based-renderer_amd"))
import pbr_amd
from pbr_amd import scenes
pt = pbr_amd.PathTracer(0).load_scene(scenes.atrium())
for w, h, spp in ((32, 32, 1), (256, 256, 1), (1024, 1024, 1), (1920, 1080, 4), (1920, 1080, 32)):
    for rep in range(2):
        pt.render(w, h, spp, seed=3, max_bounces=8)
    s = pt.stats()
    nc, na = s["launches_trace_closest"], s["launches_trace_any"]
    print("%5dx%-5d x%-3d paths %9d | per launch: closest %8.1f us  any %8.1f us  shade %8.1f us | whole batch %8.1f us"
          % (w, h, spp, s["paths"], 1e6 * s["seconds_trace_closest"] / nc, 1e6 * s["seconds_trace_any"] / max(na, 1), 1e6 * s["seconds_shade"] / nc, 1e6 * s["seconds_render"]))
