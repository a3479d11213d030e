"""Lane-utilisation diagnostics of k_trace_closest (needs libptc built with EXTRA=-DPT_DIAG)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "physically-based-renderer_amd"))
import pbr_amd
from pbr_amd import scenes
pt = pbr_amd.PathTracer(0).load_scene(scenes.atrium())
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4          # 64 = the bench's batch size: small batches overstate start-up and drain
pt.frame_begin(1920, 1080, spp, seed=3, max_bounces=8)
pt.frame_add_samples(spp); pt.sync()
st = pt.stats(); c = pt.raw_counters()
names = ["segments", "shadow", "hits", "nodes_c", "tris_c", "nodes_a", "tris_a", "node_iters", "tri_iters", "leaf_visits", "rounds", "refilled"]   # leaf_visits / refilled: leaf passes with leftovers / triangles left over   # node_iters / tri_iters / rounds: wave-level iterations of the closest-hit kernel
d = dict(zip(names, c)); print(d)
print("node-phase lane utilisation  %.3f" % (d["nodes_c"] / (64.0 * max(1, d["node_iters"]))))
print("leaf-phase lane utilisation  %.3f  (triangle tests, helpers included, per 64 lanes and pass)" % (d["tris_c"] / (64.0 * max(1, d["tri_iters"]))))
print("leaf passes that leave triangles pending  %.3f of the passes; %.2f triangles left per such pass (of %.2f tested per pass)" % (d["leaf_visits"] / max(1, d["tri_iters"]), d["refilled"] / max(1, d["leaf_visits"]), d["tris_c"] / max(1, d["tri_iters"])))
print("rays per refill round        %.2f" % (d["segments"] / max(1, d["rounds"])))
print("node visits/ray %.1f tri tests/ray %.1f  node wave-iters/ray %.2f tri wave-iters/ray %.2f" % (d["nodes_c"]/d["segments"], d["tris_c"]/d["segments"], d["node_iters"]*64/d["segments"], d["tri_iters"]*64/d["segments"]))
print({k: st[k] for k in ("seconds_trace_closest", "seconds_trace_any", "seconds_shade", "seconds_render")})
