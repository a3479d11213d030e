#!/usr/bin/env python3
"""tests/golden/hdr_*.hdr + .npy: Radiance RGBE files written by pbr_amd.gltf.hdr_encode and their pixels AS THE REFERENCE'S stb_image DECODES THEM
(oracle/_ref/libstb_image_ref.so = src/pbr_engine/image/stb/stb_image.cpp compiled where it lies: stbi_loadf_from_memory(..., 3)).  Run in the build
container (it needs /root/reference for oracle/_ref); the fixtures travel, the reference does not."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "physically-based-renderer_amd")]
from oracle import ora
from pbr_amd import gltf

HERE = os.path.dirname(os.path.abspath(__file__))
assert ora.have_ref_stb(), "oracle/_ref is not built: make -C oracle ref"
rng = np.random.default_rng(2026)
yy, xx = np.mgrid[0:24, 0:40]
sky = np.stack([0.2 + 3.0 * np.exp(-((xx - 30) ** 2 + (yy - 5) ** 2) / 9.0) * 400, 0.3 + 0.02 * yy, 0.6 - 0.01 * yy + 1e-3 * xx], 2)     # a sun of 1200 over a dim gradient
cases = {"sky_rle": (sky, True, "#?RADIANCE"), "noise_flat": (rng.uniform(0, 8, (9, 12, 3)) ** 3, False, "#?RGBE"),
         "narrow": (rng.uniform(0, 1, (5, 7, 3)), True, "#?RADIANCE"), "runs": (np.repeat(np.repeat(rng.uniform(0, 2, (3, 4, 3)), 6, 0), 9, 1), True, "#?RADIANCE")}
cases["runs"][0][2:5, 3:20] = 0.0          # black (exponent byte 0) inside runs
for name, (img, rle, magic) in cases.items():
    data = gltf.hdr_encode(img, rle, magic)
    open(os.path.join(HERE, f"hdr_{name}.hdr"), "wb").write(data)
    ref = ora.ref_stb_decode_float(data)
    np.save(os.path.join(HERE, f"hdr_{name}.npy"), ref)
    print(name, ref.shape, len(data), "bytes; max", float(ref.max()))
