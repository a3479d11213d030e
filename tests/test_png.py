"""The loader's PNG decoder (host/png_decode.hpp, C-ABI ptc_png_decode_rgba8) against PNG files written by
pbr_amd.gltf.png_encode with Python's zlib: every colour type and bit depth, all five row filters, Adam7,
stored / fixed / dynamic deflate blocks, split IDAT, tRNS, and the failure modes.  The expected RGBA8 follows
what the reference's loader produces (stb_image with 4 requested channels, src/pbr_engine/image/pbr/image/
LoadImage.cpp:56-73): 16-bit samples keep the high byte, sub-byte greys scale to 0..255, palettes expand."""
import struct
import zlib

import numpy as np
import pytest


@pytest.fixture(scope="module")
def g(pbr):
    return pbr.gltf


def _expand(s, color_type, depth, palette=None, trns=None):
    s = np.asarray(s).astype(np.int64)
    h, w, _ = s.shape
    out = np.zeros((h, w, 4), np.uint8)

    def to8(v):
        return {16: v >> 8, 8: v, 4: v * 17, 2: v * 85, 1: v * 255}[depth].astype(np.uint8)

    if color_type == 0:
        out[..., 0] = out[..., 1] = out[..., 2] = to8(s[..., 0])
        out[..., 3] = 255
        if trns is not None:
            out[..., 3][s[..., 0] == struct.unpack(">H", trns)[0]] = 0
    elif color_type == 2:
        out[..., :3] = to8(s)
        out[..., 3] = 255
        if trns is not None:
            key = struct.unpack(">HHH", trns)
            out[..., 3][(s[..., 0] == key[0]) & (s[..., 1] == key[1]) & (s[..., 2] == key[2])] = 0
    elif color_type == 3:
        pal = np.asarray(palette, np.uint8)
        out[..., :3] = pal[s[..., 0]]
        a = np.full(len(pal), 255, np.uint8)
        if trns is not None:
            a[: len(trns)] = np.frombuffer(trns, np.uint8)
        out[..., 3] = a[s[..., 0]]
    elif color_type == 4:
        out[..., 0] = out[..., 1] = out[..., 2] = to8(s[..., 0])
        out[..., 3] = to8(s[..., 1])
    else:
        out[...] = to8(s)
    return out


CASES = [(0, d) for d in (1, 2, 4, 8, 16)] + [(2, 8), (2, 16)] + [(3, d) for d in (1, 2, 4, 8)] + [(4, 8), (4, 16), (6, 8), (6, 16)]


@pytest.mark.parametrize("interlace", [False, True])
@pytest.mark.parametrize("color_type,depth", CASES)
def test_all_colour_types_and_depths(g, color_type, depth, interlace):
    rng = np.random.default_rng(color_type * 100 + depth)
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color_type]
    for (h, w) in ((1, 1), (7, 5), (19, 33)):
        palette = trns = None
        hi = 1 << depth
        if color_type == 3:
            palette = rng.integers(0, 256, (min(hi, 200), 3))
            hi = len(palette)
            trns = bytes(rng.integers(0, 256, hi // 2 + 1, dtype=np.uint8)) if hi > 1 else b"\x07"
        s = rng.integers(0, hi, (h, w, ch))
        if color_type == 0:
            trns = struct.pack(">H", int(s[0, 0, 0]))
        if color_type == 2:
            trns = struct.pack(">HHH", *[int(x) for x in s[0, 0]])
        data = g.png_encode(s, color_type, depth, interlace=interlace, palette=palette, trns=trns)
        got = g.png_decode(data)
        assert got.shape == (h, w, 4)
        assert (got == _expand(s, color_type, depth, palette, trns)).all(), (color_type, depth, interlace, h, w)


@pytest.mark.parametrize("level,strategy", [(0, 0), (1, 0), (9, 0), (6, zlib.Z_FIXED), (6, zlib.Z_RLE), (6, zlib.Z_HUFFMAN_ONLY)])
@pytest.mark.parametrize("filters", ["cycle", 0, 1, 2, 3, 4])
def test_deflate_block_kinds_and_filters(g, level, strategy, filters):
    """A compressible 256×96 image (gradients + repeated rows: long matches at distance > 1024, length 258) so that
    length/distance codes with extra bits are exercised, not only literals."""
    y, x = np.mgrid[0:96, 0:256]
    img = np.stack([(x + y) & 255, (x * 3) & 255, ((x // 16) * 16 + (y // 8)) & 255, np.where((x // 32 + y // 32) % 2, 255, 40)], -1).astype(np.uint8)
    img[40:60] = img[0:20]
    img[60:] = 77
    data = g.png_encode(img, 6, 8, filters=filters, level=level, strategy=strategy, idat_split=997)
    assert (g.png_decode(data) == img).all()


def test_large_image_round_trip(g):
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, (64, 64, 4), dtype=np.uint8)
    img = np.tile(base, (8, 8, 1))                       # 512×512, matches at distance 256 and 64·2048
    img[::7, ::5, 0] ^= 0x55
    assert (g.png_decode(g.png_encode(img, 6, 8, filters=4)) == img).all()


def _chunks(data):
    p, out = 8, []
    while p < len(data):
        n = struct.unpack(">I", data[p : p + 4])[0]
        out.append((data[p + 4 : p + 8], data[p + 8 : p + 8 + n]))
        p += 12 + n
    return out


def _join(chunks):
    out = b"\x89PNG\r\n\x1a\n"
    for t, b in chunks:
        out += struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xFFFFFFFF)
    return out


def test_failures_are_reported(g, pbr):
    img = np.arange(8 * 8 * 4, dtype=np.uint8).reshape(8, 8, 4)
    good = g.png_encode(img, 6, 8)
    assert (g.png_decode(good) == img).all()
    ch = _chunks(good)
    bad = {
        "signature": b"\x89PNX" + good[4:],
        "CRC": good[:20] + bytes([good[20] ^ 1]) + good[21:],
        "chunk exceeds the file": good[: len(good) - 20],
        "truncated chunk": good[: len(good) - 12],
        "missing IHDR or IDAT": _join([c for c in ch if c[0] != b"IDAT"]),
        "Adler": _join([(t, b[:-1] + bytes([b[-1] ^ 1])) if t == b"IDAT" else (t, b) for t, b in ch]),
        "bit depth": _join([(t, b[:8] + b"\x03" + b[9:]) if t == b"IHDR" else (t, b) for t, b in ch]),
        "does not match": _join([(t, b[:4] + struct.pack(">I", 9) + b[8:]) if t == b"IHDR" else (t, b) for t, b in ch]),
        "unknown critical": _join(ch[:1] + [(b"XXXX", b"abc")] + ch[1:]),
    }
    for msg, data in bad.items():
        with pytest.raises(pbr.PtcError, match=msg):
            g.png_decode(data)
    # an ancillary chunk with a broken CRC is skipped, like any unknown ancillary chunk
    anc = _join(ch[:1] + [(b"tEXt", b"Comment\x00hello")] + ch[1:])
    assert (g.png_decode(anc) == img).all()
