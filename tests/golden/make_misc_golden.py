#!/usr/bin/env python3
"""Fixtures of the BMP / TGA / PGM / PPM decoders: tests/misc_image_files.py writes one file per header kind, pixel layout and row order, and the REFERENCE's decoder —
oracle/_ref/libstb_image_ref.so, its vendored stb_image translation unit compiled where it lies under /root/reference (oracle/Makefile `ref`) — says what
image::loadImage2D would hand to Vulkan for it (stbi_load_from_memory(..., 4)).  Written as tests/golden/misc_images.npz: name → file bytes, name + ':rgba' → texels.
Run in the container that holds the reference checkout; the GPU box only reads the fixture."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ora
import misc_image_files as mif

assert ora.have_ref_stb(), "oracle/_ref is not built: make -C oracle ref (needs /root/reference)"
out = {}
for name, data in mif.corpus():
    out[name] = np.frombuffer(data, np.uint8)
    out[name + ":rgba"] = ora.ref_stb_decode(data)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "misc_images.npz"), **out)
print(len(out) // 2, "files")
