"""Mutation fuzzer driven by tools/sanitize.sh (ASan/UBSan build of the JPEG decoder): every mutant of the golden JPEGs must decode or be rejected with an error, never crash."""
import ctypes as C, glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "build_san"); ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
L = C.CDLL(os.path.join(OUT, "libjpeg_fuzz.so")); L.jpeg_try.argtypes = [C.c_char_p, C.c_ulonglong]
seeds = [open(f, "rb").read() for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "jpeg_*.jpg")))]
rng = np.random.default_rng(2)
ok = bad = 0
for it in range(ITERS):
    d = bytearray(seeds[it % len(seeds)])
    for _ in range(rng.integers(1, 5)):
        if len(d) < 8: break
        op = rng.integers(0, 4)
        if op == 0: d[rng.integers(0, len(d))] = rng.integers(0, 256)
        elif op == 1: d = d[: rng.integers(4, len(d))]
        elif op == 2:
            i = rng.integers(2, len(d)); d[i:i] = bytes(rng.integers(0, 256, rng.integers(1, 6), dtype=np.uint8))
        else:                                    # hit the header area, where the structure lives
            d[rng.integers(2, min(len(d), 700))] = rng.choice([0, 1, 2, 3, 4, 8, 16, 17, 0x7F, 0x80, 0xC0, 0xC2, 0xC4, 0xDA, 0xDD, 0xFF])
    r = L.jpeg_try(bytes(d), len(d))
    ok, bad = (ok + 1, bad) if r >= 0 else (ok, bad + 1)
print("decoded", ok, "rejected", bad)
