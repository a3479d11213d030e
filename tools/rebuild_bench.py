#!/usr/bin/env python3
"""What a NEW tree costs on the benchmark scene: ptc_update_instance on a third of the instances + ptc_scene_rebuild (csrc/pt_build.hip: the LBVH built on
the device from the vertices in HBM), next to ptc_scene_refit and to ptc_scene_commit with the LBVH builder — on the device (flatten, shading records and tree: the
default on a device context) and on the host (PTC_COMMIT=host), first commit of a fresh context and a second commit that replaces the first.  Under `rocprofv3 --kernel-trace --stats`
the k_bld_* / k_sort_* rows give the per-kernel times.  usage: python3 tools/rebuild_bench.py [atrium|textured] [turns]"""
import copy, json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "physically-based-renderer_amd"))
import numpy as np
import pbr_amd as pbr

name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
turns = int(sys.argv[2]) if len(sys.argv) > 2 else 10
d = copy.deepcopy(pbr.scenes.by_name("textured_atrium" if name == "textured" else "atrium"))
d.bvh_builder = "lbvh"
os.environ["PTC_COMMIT"] = "host"
hp = pbr.PathTracer(0).load_scene(d)
host_first = hp.stats()["seconds_commit"] * 1e3
host_again = hp.load_scene(d).stats()["seconds_commit"] * 1e3
del os.environ["PTC_COMMIT"]
hp.close()
pt = pbr.PathTracer(0).load_scene(d)
assert pt.internals()["commit_on_device"] == 1
dev_first = pt.stats()["seconds_commit"] * 1e3
dev_again = pt.load_scene(d).stats()["seconds_commit"] * 1e3
fresh = [pbr.PathTracer(0).load_scene(d).stats()["seconds_commit"] * 1e3 for _ in range(3)]      # fresh contexts of a process whose code is loaded
out = {"scene": d.name, "triangles": pt.stats()["n_triangles"], "bvh_nodes": pt.stats()["n_bvh_nodes"],
       "commit_host_lbvh_ms": {"first": host_first, "replacing": host_again},
       "commit_device_lbvh_ms": {"first_in_process": dev_first, "replacing": dev_again, "fresh_context": float(np.median(fresh))}}
reb, ref = [], []
for k in range(turns + 1):
    for i, it in enumerate(d.instances):
        if i % 3 or getattr(it, "matrix", None) is not None:
            continue
        a = 0.05 * (k + 1)
        pt.update_instance(i, it.t, (math.cos(a / 2), 0.0, math.sin(a / 2), 0.0), it.s)
    pt.scene_refit()
    ref.append(pt.stats()["seconds_refit"] * 1e3)
    pt.scene_rebuild()
    reb.append(pt.stats()["seconds_rebuild"] * 1e3)
out["refit_ms"] = {"median": float(np.median(ref[1:])), "min": float(np.min(ref[1:]))}
out["rebuild_ms"] = {"median": float(np.median(reb[1:])), "min": float(np.min(reb[1:])), "first": reb[0], "turns": turns}
st = pt.stats()
out["bvh_nodes_after"] = st["n_bvh_nodes"]; out["sa_cost"] = st["bvh_sa_cost"]
img = pt.render(64, 36, 1, seed=1, max_bounces=2)
out["rendered_finite"] = bool(np.isfinite(img).all())
print(json.dumps(out))
