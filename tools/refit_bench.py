#!/usr/bin/env python3
"""What a transform change costs on the benchmark scene: ptc_update_instance on a third of the instances + ptc_scene_refit, on the device
(csrc/pt_refit.hip) and on the host (PTC_REFIT=host), next to ptc_scene_commit.  Under `rocprofv3 --kernel-trace --stats` the k_refit_* rows
give the per-kernel times.  usage: python3 tools/refit_bench.py [atrium|textured] [turns]"""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "physically-based-renderer_amd"))
import numpy as np
import pbr_amd as pbr

name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
turns = int(sys.argv[2]) if len(sys.argv) > 2 else 20
d = pbr.scenes.by_name("textured_atrium" if name == "textured" else "atrium")
pt = pbr.PathTracer(0).load_scene(d)
out = {"scene": d.name, "triangles": pt.stats()["n_triangles"], "bvh_nodes": pt.stats()["n_bvh_nodes"], "instances": len(d.instances),
       "commit_ms": pt.stats()["seconds_commit"] * 1e3}
for mode in ("device", "host"):
    if mode == "host":
        os.environ["PTC_REFIT"] = "host"
    ms = []
    for k in range(turns + 1):
        for i, it in enumerate(d.instances):
            if i % 3 or getattr(it, "matrix", None) is not None:
                continue
            a = 0.01 * (k + 1)
            pt.update_instance(i, it.t, (math.cos(a / 2), 0.0, math.sin(a / 2), 0.0), it.s)
        pt.scene_refit()
        assert pt.internals()["refit_on_device"] == (1 if mode == "device" else 0)
        if k:                                    # the first device refit also builds and uploads the plan
            ms.append(pt.stats()["seconds_refit"] * 1e3)
    out[mode + "_refit_ms"] = {"median": float(np.median(ms)), "min": float(np.min(ms)), "max": float(np.max(ms)), "turns": turns}
os.environ.pop("PTC_REFIT", None)
img = pt.render(64, 36, 1, seed=1, max_bounces=2)   # the refitted scene renders
out["rendered_finite"] = bool(np.isfinite(img).all())
print(json.dumps(out))
