#!/usr/bin/env python3
"""ptc_scene_commit with the LBVH builder ON THE DEVICE (flatten, shading records, tree: csrc/ptc_api.cpp device_commit), n times on fresh contexts of one process, next to the host's
commit of the same description (PTC_COMMIT=host) and the default SAH commit.  Under `rocprofv3 --kernel-trace --stats` the k_refit_* / k_bld_* / k_sort_* rows are the device's share.
usage: python3 tools/commit_bench.py [atrium|textured] [n]"""
import copy, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "physically-based-renderer_amd"))
import numpy as np
import pbr_amd as pbr

name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = copy.deepcopy(pbr.scenes.by_name("textured_atrium" if name == "textured" else "atrium"))
out = {"scene": d.name}
d.bvh_builder = "lbvh"
dev, wall = [], []
for k in range(n + 1):
    t0 = time.perf_counter()
    pt = pbr.PathTracer(0).load_scene(d)          # the description calls (ptc_add_mesh ...: copies into the library) + the commit
    wall.append((time.perf_counter() - t0) * 1e3)
    assert pt.internals()["commit_on_device"] == 1
    dev.append(pt.stats()["seconds_commit"] * 1e3)
    if k == n:
        img = pt.render(64, 36, 1, seed=1, max_bounces=2)
        out["rendered_finite"] = bool(np.isfinite(img).all())
        out["triangles"] = pt.stats()["n_triangles"]; out["bvh_nodes"] = pt.stats()["n_bvh_nodes"]
    pt.close()
out["commit_device_lbvh_ms"] = {"first_in_process": dev[0], "median_fresh_context": float(np.median(dev[1:])), "min": float(np.min(dev[1:])), "max": float(np.max(dev[1:])), "n": n}
out["describe_plus_commit_wall_ms"] = {"median": float(np.median(wall[1:]))}
os.environ["PTC_COMMIT"] = "host"
host = [pbr.PathTracer(0).load_scene(d).stats()["seconds_commit"] * 1e3 for _ in range(3)]
del os.environ["PTC_COMMIT"]
out["commit_host_lbvh_ms"] = {"median": float(np.median(host))}
d.bvh_builder = "sah"
sah = [pbr.PathTracer(0).load_scene(d).stats()["seconds_commit"] * 1e3 for _ in range(3)]
out["commit_host_sah_ms"] = {"median": float(np.median(sah))}
print(json.dumps(out))
