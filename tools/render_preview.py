#!/usr/bin/env python3
"""Renders preview PNGs of the procedural scenes (tonemapped like the reference: ACES + gamma) into gpurun_out/.  usage: python tools/render_preview.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "physically-based-renderer_amd"))
import numpy as np
import pbr_amd
from pbr_amd import scenes, gltf
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for name, desc, w, h, spp in (("atrium", scenes.atrium(), 640, 360, 256), ("textured_atrium", scenes.textured_atrium(), 640, 360, 256),
                              ("cornell", scenes.cornell_box(), 320, 320, 256), ("sphere", scenes.sphere_scene(), 480, 320, 256),
                              ("textured_objects", scenes.textured_objects(), 480, 320, 256)):
    desc.camera.aspect = w / h
    pt = pbr_amd.PathTracer(0).load_scene(desc)
    img = pt.render(w, h, spp, seed=1)
    ldr = pt.tonemap()
    open(os.path.join(ROOT, "gpurun_out", name + ".png"), "wb").write(gltf.png_encode(np.ascontiguousarray(ldr), 6, 8, filters=0, level=6))
    print(name, "mean radiance", float(img[..., :3].mean()), "finite", bool(np.isfinite(img).all()))
