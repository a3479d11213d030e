"""Header mutations of the BMP / TGA / PNM corpus (tests/misc_image_files.py): the decoders of host/misc_decode.hpp against the reference's stb build (oracle/_ref) — what differs is printed
(expected: files whose raw TGA rows pass the end of the file, which stb leaves undefined and this decoder refuses; palette indices beyond the entries stb reads, which are undefined there and black here)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "physically-based-renderer_amd"))
import pbr_amd as pbr
from oracle import ora
import misc_image_files as mif
rng = np.random.default_rng(2)
corp = mif.corpus()
n = 0
for it in range(6000):
    name, data = corp[int(rng.integers(len(corp)))]
    d = bytearray(data)
    pos = []
    for _ in range(int(rng.integers(1, 3))):
        p = int(rng.integers(min(len(d), 64)))
        if name.startswith("bmp") and 18 <= p < 26 and (p % 4) >= 2: continue     # no huge sizes
        if name.startswith("tga") and p in (13, 15): continue
        d[p] = int(rng.integers(256)); pos.append(p)
    d = bytes(d)
    try: g = pbr.gltf.image_decode(d)
    except ValueError as e: g = None; ge = str(e)
    try: r = ora.ref_stb_decode(d)
    except ValueError as e: r = None; re_ = str(e)
    if (g is None) != (r is None):
        n += 1; print(name, pos, "mine:", "ok" if g is not None else ge, "| stb:", "ok" if r is not None else re_)
    elif g is not None and not (g.shape == r.shape and np.array_equal(g, r)):
        n += 1; print(name, pos, "both decode", g.shape, r.shape, "differing texels", int((g != r).any(-1).sum()) if g.shape == r.shape else -1)
print("differences", n)
