"""The loader's JPEG and PNG decoders against the REFERENCE's own decoder: oracle/_ref/libstb_image_ref.so is the
reference's vendored stb_image translation unit (src/pbr_engine/image/stb/stb_image.cpp), compiled where it lies by
`make -C oracle ref` — the code behind image::loadImage2D (LoadImage.cpp:56-73).  Every file must decode to the same
RGBA8 texels, bit for bit.  JPEG inputs are written with Pillow (baseline / progressive, 4:4:4 / 4:2:2 / 4:2:0 / 4:1:1 /
4:4:0, optimised tables, restart intervals, grey, CMYK, odd sizes); PNG inputs with pbr_amd.gltf.png_encode.
The committed fixtures tests/golden/jpeg_*.jpg + .npy (made by tests/golden/make_jpeg_golden.py from the same reference
build) pin the decoder where the reference checkout is absent."""
import glob
import io
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def g(pbr):
    return pbr.gltf


def _picture(h, w, seed=0):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(x / 7.0 + seed) * np.cos(y / 5.0), (x * 5 + y * 3) % 256, 255.0 * ((x // 8 + y // 8) % 2)], -1)
    img += rng.normal(0, 12, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def _jpeg(img, mode="RGB", **kw):
    from PIL import Image

    im = Image.fromarray(img if mode != "L" else img[..., 0], mode=None)
    if mode == "CMYK":
        im = Image.fromarray(img).convert("CMYK")
    b = io.BytesIO()
    im.save(b, "JPEG", **kw)
    return b.getvalue()


def test_golden_jpeg_fixtures(g):
    files = sorted(glob.glob(os.path.join(HERE, "golden", "jpeg_*.jpg")))
    assert len(files) >= 6
    for f in files:
        want = np.load(f[:-4] + ".npy")
        got = g.jpeg_decode(open(f, "rb").read())
        assert got.shape == want.shape and (got == want).all(), f


CASES = []
for size in ((1, 1), (7, 5), (8, 8), (16, 16), (17, 33), (64, 48), (100, 75)):
    for sub in (0, 1, 2):
        for prog in (False, True):
            CASES.append((size, dict(quality=80, subsampling=sub, progressive=prog)))
CASES += [((40, 56), dict(quality=q, subsampling=s, progressive=p, optimize=o)) for q in (5, 35, 95, 100) for s in (0, 2) for p in (False, True) for o in (False, True)]
CASES += [((61, 47), dict(quality=75, subsampling=s, progressive=p)) for s in ("4:1:1", "4:4:0") for p in (False, True)]
CASES += [((64, 80), dict(quality=70, subsampling=2, progressive=p, restart_marker_blocks=r)) for p in (False, True) for r in (1, 3, 7)]
CASES += [((33, 50), dict(quality=85, subsampling=1, restart_marker_rows=1))]


@pytest.mark.parametrize("size,kw", CASES)
def test_colour_jpeg_equals_reference_stb(g, ora, size, kw):
    pytest.importorskip("PIL")
    if not ora.have_ref_stb():
        pytest.skip("oracle/_ref not built (no reference checkout)")
    try:
        data = _jpeg(_picture(*size, seed=size[0] * 31 + size[1]), **kw)
    except Exception as e:      # an encoder option this Pillow does not know
        pytest.skip(f"Pillow cannot write this variant: {e}")
    want = ora.ref_stb_decode(data)
    got = g.jpeg_decode(data)
    assert got.shape == want.shape
    assert (got == want).all(), f"{int((got != want).sum())} bytes differ"


@pytest.mark.parametrize("prog", [False, True])
@pytest.mark.parametrize("mode", ["L", "CMYK"])
def test_grey_and_cmyk_jpeg_equal_reference_stb(g, ora, mode, prog):
    pytest.importorskip("PIL")
    if not ora.have_ref_stb():
        pytest.skip("oracle/_ref not built (no reference checkout)")
    for size in ((9, 14), (48, 32)):
        data = _jpeg(_picture(*size, seed=3), mode=mode, quality=80, progressive=prog)
        want, got = ora.ref_stb_decode(data), g.jpeg_decode(data)
        assert got.shape == want.shape and (got == want).all(), (mode, prog, size)


def test_png_decoder_equals_reference_stb(g, ora):
    """The PNG side of the same contract: every colour type / depth / interlacing that tests/test_png.py writes."""
    if not ora.have_ref_stb():
        pytest.skip("oracle/_ref not built (no reference checkout)")
    rng = np.random.default_rng(9)
    n = 0
    for ct, depths in ((0, (1, 2, 4, 8, 16)), (2, (8, 16)), (3, (1, 2, 4, 8)), (4, (8, 16)), (6, (8, 16))):
        ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ct]
        for dp in depths:
            for il in (False, True):
                pal = rng.integers(0, 256, (min(1 << dp, 200), 3)) if ct == 3 else None
                hi = len(pal) if ct == 3 else (1 << dp)
                s = rng.integers(0, hi, (13, 21, ch))
                trns = bytes(rng.integers(0, 256, hi // 2 + 1, dtype=np.uint8)) if ct == 3 else None
                data = g.png_encode(s, ct, dp, interlace=il, palette=pal, trns=trns)
                want, got = ora.ref_stb_decode(data), g.png_decode(data)
                assert got.shape == want.shape and (got == want).all(), (ct, dp, il)
                n += 1
    assert n == 30


def test_jpeg_failures_are_reported(g, pbr):
    pytest.importorskip("PIL")
    good = _jpeg(_picture(24, 24), quality=80)
    assert g.jpeg_decode(good).shape == (24, 24, 4)
    sof = good.index(b"\xff\xc0")
    bad = {
        "no SOI": b"\x00" + good[1:],
        "truncated|missing EOI|bad Huffman": good[: len(good) // 2],
        "8-bit": good[: sof + 4] + b"\x0c" + good[sof + 5 :],
        "not supported": good[:sof] + b"\xff\xc9" + good[sof + 2 :],
        "component count": good[: sof + 9] + b"\x02" + good[sof + 10 :],
    }
    for msg, data in bad.items():
        with pytest.raises(pbr.PtcError, match=msg):
            g.jpeg_decode(data)


# ---- a small baseline JPEG writer (numpy DCT, Annex K Huffman tables) for sampling layouts Pillow does not offer ----
_ZZ = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36,
       29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
_DC_BITS = [0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]
_DC_VALS = list(range(12))
_AC_BITS = [0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7D]
_AC_VALS = [0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xA1, 0x08, 0x23, 0x42, 0xB1, 0xC1,
            0x15, 0x52, 0xD1, 0xF0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0A, 0x16, 0x17, 0x18, 0x19, 0x1A, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2A, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39,
            0x3A, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4A, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5A, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6A, 0x73, 0x74, 0x75,
            0x76, 0x77, 0x78, 0x79, 0x7A, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8A, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9A, 0xA2, 0xA3, 0xA4, 0xA5, 0xA6, 0xA7,
            0xA8, 0xA9, 0xAA, 0xB2, 0xB3, 0xB4, 0xB5, 0xB6, 0xB7, 0xB8, 0xB9, 0xBA, 0xC2, 0xC3, 0xC4, 0xC5, 0xC6, 0xC7, 0xC8, 0xC9, 0xCA, 0xD2, 0xD3, 0xD4, 0xD5, 0xD6, 0xD7, 0xD8,
            0xD9, 0xDA, 0xE1, 0xE2, 0xE3, 0xE4, 0xE5, 0xE6, 0xE7, 0xE8, 0xE9, 0xEA, 0xF1, 0xF2, 0xF3, 0xF4, 0xF5, 0xF6, 0xF7, 0xF8, 0xF9, 0xFA]


def _codes(bits, vals):
    out, code, k = {}, 0, 0
    for ln in range(1, 17):
        for _ in range(bits[ln - 1]):
            out[vals[k]] = (code, ln)
            code += 1
            k += 1
        code <<= 1
    return out


def _write_jpeg(planes, factors, ids=(1, 2, 3), q=6, app=b"", restart=0, interleaved=True):
    """planes: full-resolution (H, W) uint8 arrays, one per component; factors: [(h, v)] sampling factors.  Each component is
    box-subsampled to its own resolution, DCT'd, quantised by the flat table q, and written as a baseline JPEG."""
    import struct

    H, W = planes[0].shape
    hmax, vmax = max(f[0] for f in factors), max(f[1] for f in factors)
    mx, my = -(-W // (8 * hmax)), -(-H // (8 * vmax))
    dc_c, ac_c = _codes(_DC_BITS, _DC_VALS), _codes(_AC_BITS, _AC_VALS)
    k = np.arange(8)
    Cm = np.sqrt(2 / 8) * np.cos((2 * k[None, :] + 1) * k[:, None] * np.pi / 16)
    Cm[0] /= np.sqrt(2)
    coefs = []
    for p, (h, v) in zip(planes, factors):
        hs, vs = hmax // h, vmax // v
        cw, chh = -(-W * h // hmax), -(-H * v // vmax)
        pad = np.pad(p.astype(np.float64), ((0, chh * vs - H), (0, cw * hs - W)), mode="edge")
        sub = pad.reshape(chh, vs, cw, hs).mean((1, 3))
        full = np.pad(sub, ((0, my * v * 8 - chh), (0, mx * h * 8 - cw)), mode="edge") - 128.0
        blocks = full.reshape(my * v, 8, mx * h, 8).transpose(0, 2, 1, 3)
        d = np.rint(np.einsum("ij,abjk,lk->abil", Cm, blocks, Cm) / q).astype(np.int64)
        coefs.append((d, cw, chh))
    bits = []

    def put(code, ln):
        bits.append((code, ln))

    def cat(v):
        a = abs(int(v))
        n = a.bit_length()
        return n, (int(v) if v >= 0 else int(v) + (1 << n) - 1) & ((1 << n) - 1)

    def block(blk, pred):
        z = blk.reshape(64)[_ZZ]
        n, b = cat(z[0] - pred)
        put(*dc_c[n])
        if n:
            put(b, n)
        run = 0
        last = max([i for i in range(1, 64) if z[i] != 0], default=0)
        for i in range(1, last + 1):
            if z[i] == 0:
                run += 1
                continue
            while run > 15:
                put(*ac_c[0xF0])
                run -= 16
            n, b = cat(z[i])
            put(*ac_c[(run << 4) | n])
            put(b, n)
            run = 0
        if last < 63:
            put(*ac_c[0x00])
        return int(z[0])

    def flush():
        acc = nb = 0
        out = bytearray()
        for code, ln in bits:
            acc = (acc << ln) | code
            nb += ln
            while nb >= 8:
                byte = (acc >> (nb - 8)) & 255
                out.append(byte)
                if byte == 255:
                    out.append(0)
                nb -= 8
        if nb:
            byte = ((acc << (8 - nb)) | ((1 << (8 - nb)) - 1)) & 255
            out.append(byte)
            if byte == 255:
                out.append(0)
        bits.clear()
        return bytes(out)

    def seg(m, body):
        return bytes([0xFF, m]) + struct.pack(">H", len(body) + 2) + body

    nc = len(planes)
    out = b"\xff\xd8" + app + seg(0xDB, bytes([0]) + bytes([q] * 64))
    out += seg(0xC0, struct.pack(">BHHB", 8, H, W, nc) + b"".join(bytes([ids[i], (factors[i][0] << 4) | factors[i][1], 0]) for i in range(nc)))
    out += seg(0xC4, bytes([0x00]) + bytes(_DC_BITS) + bytes(_DC_VALS)) + seg(0xC4, bytes([0x10]) + bytes(_AC_BITS) + bytes(_AC_VALS))
    if restart:
        out += seg(0xDD, struct.pack(">H", restart))
    scans = [list(range(nc))] if interleaved or nc == 1 else [[i] for i in range(nc)]
    for comps in scans:
        out += seg(0xDA, bytes([len(comps)]) + b"".join(bytes([ids[i], 0x00]) for i in comps) + bytes([0, 63, 0]))
        pred = [0] * nc
        count = rst = 0
        if len(comps) == 1:
            d, cw, chh = coefs[comps[0]]
            units = [[(comps[0], j, i)] for j in range(-(-chh // 8)) for i in range(-(-cw // 8))]
        else:
            units = [[(c, j * factors[c][1] + y, i * factors[c][0] + x) for c in comps for y in range(factors[c][1]) for x in range(factors[c][0])]
                     for j in range(my) for i in range(mx)]
        for u, unit in enumerate(units):
            for c, by, bx in unit:
                pred[c] = block(coefs[c][0][by, bx], pred[c])
            count += 1
            if restart and count == restart and u + 1 < len(units):
                out += flush() + bytes([0xFF, 0xD0 + rst])
                rst = (rst + 1) & 7
                pred = [0] * nc
                count = 0
        out += flush()
    return out + b"\xff\xd9"


LAYOUTS = [
    ([(1, 2), (1, 1), (1, 1)], "4:4:0: vertical-only chroma subsampling (3:1 vertical filter)"),
    ([(2, 2), (1, 1), (1, 1)], "4:2:0"),
    ([(2, 1), (1, 1), (1, 1)], "4:2:2"),
    ([(4, 2), (1, 1), (1, 1)], "4:1:0: replication path"),
    ([(1, 4), (1, 1), (1, 2)], "vertical factor 4 and mixed factors"),
    ([(2, 2), (2, 1), (1, 2)], "every component at its own resolution"),
    ([(1, 1), (1, 1), (1, 1)], "4:4:4"),
]


@pytest.mark.parametrize("factors,what", LAYOUTS)
def test_sampling_layouts_equal_reference_stb(g, ora, factors, what):
    if not ora.have_ref_stb():
        pytest.skip("oracle/_ref not built (no reference checkout)")
    for (h, w) in ((5, 3), (16, 16), (37, 29), (50, 70)):
        pic = _picture(h, w, seed=h)
        planes = [pic[..., 0], pic[..., 1], pic[..., 2]]
        for kw in (dict(), dict(interleaved=False), dict(restart=2)):
            data = _write_jpeg(planes, factors, **kw)
            want, got = ora.ref_stb_decode(data), g.jpeg_decode(data)
            assert got.shape == want.shape == (h, w, 4) and (got == want).all(), (what, h, w, kw)


def test_colour_space_tags_equal_reference_stb(g, ora):
    """Component ids 'R','G','B' (no conversion), Adobe APP14 transform 0 without JFIF (RGB), Adobe CMYK and YCCK, JFIF."""
    import struct

    if not ora.have_ref_stb():
        pytest.skip("oracle/_ref not built (no reference checkout)")
    pic = _picture(24, 40, seed=5)
    planes = [pic[..., 0], pic[..., 1], pic[..., 2]]
    four = planes + [255 - pic[..., 0]]
    adobe = lambda t: b"\xff\xee" + struct.pack(">H", 14) + b"Adobe\x00" + bytes([100, 0, 0, 0, 0, t])
    jfif = b"\xff\xe0" + struct.pack(">H", 16) + b"JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00"
    f3, f4 = [(1, 1)] * 3, [(1, 1)] * 4
    for data in (_write_jpeg(planes, f3, ids=(82, 71, 66)), _write_jpeg(planes, f3, app=adobe(0)), _write_jpeg(planes, f3, app=jfif + adobe(0)), _write_jpeg(planes, f3, app=jfif),
                 _write_jpeg(four, f4, ids=(1, 2, 3, 4), app=adobe(0)), _write_jpeg(four, f4, ids=(1, 2, 3, 4), app=adobe(2)), _write_jpeg(four, f4, ids=(1, 2, 3, 4)),
                 _write_jpeg([(2, 2), (1, 1), (1, 1), (2, 2)] and four, [(2, 2), (1, 1), (1, 1), (2, 2)], ids=(1, 2, 3, 4), app=adobe(2))):
        want, got = ora.ref_stb_decode(data), g.jpeg_decode(data)
        assert got.shape == want.shape and (got == want).all()
