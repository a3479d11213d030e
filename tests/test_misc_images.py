"""The BMP / TGA / PGM / PPM / GIF / PSD / PIC / Radiance-as-texture decoders of the glTF loader (host/misc_decode.hpp, ptc_image_decode_rgba8) against the REFERENCE's own decoder: image::loadImage2D hands a glTF
image's bytes to stbi_load_from_memory(..., 4) whatever format they are in (src/pbr_engine/image/pbr/image/LoadImage.cpp:56-73); oracle/_ref is that vendored stb_image
translation unit compiled where it lies.  The committed fixture tests/golden/misc_images.npz was generated from it (tests/golden/make_misc_golden.py) and pins the decoders where the
reference checkout is absent (the GPU box); with the checkout present every file of the corpus, random files and cut-short files are held against it directly."""
import json
import os
import struct

import numpy as np
import pytest

import misc_image_files as mif

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "misc_images.npz")


@pytest.fixture(scope="module")
def g(pbr):
    return pbr.gltf


def test_decoders_equal_the_fixtures_of_the_reference_stb(g):
    z = np.load(GOLD)
    names = [k for k in z.files if not k.endswith(":rgba")]
    assert len(names) >= 125
    for name in names:
        want = z[name + ":rgba"]
        got = g.image_decode(z[name].tobytes())
        assert got.shape == want.shape and np.array_equal(got, want), name


def test_the_corpus_is_what_the_fixture_was_made_from(g):
    """the writers are deterministic: the fixture's files are today's corpus (a changed writer needs tests/golden/make_misc_golden.py run again)"""
    z = np.load(GOLD)
    for name, data in mif.corpus():
        assert name in z.files and z[name].tobytes() == data, name


def test_decoders_equal_reference_stb_on_random_files(g, ora):
    if not ora.have_ref_stb():
        pytest.skip("oracle/_ref not built (no reference checkout)")
    rng = np.random.default_rng(23)
    n = 0
    for k in range(120):
        h, w = int(rng.integers(1, 24)), int(rng.integers(1, 40))
        img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        if k % 2:
            img[:, : w // 2] = img[:, :1]                                      # runs
        files = [mif.bmp(img, rng.choice(["24", "32", "32zero", "555"]), int(rng.choice([40, 56, 108, 124])), top_down=bool(k % 3 == 0)),
                 mif.bmp(img, "24", 12),
                 mif.tga(img, str(rng.choice(["8", "16ga", "15", "16", "24", "32"])), rle=bool(k % 2), top_down=bool(k % 5 == 0), ident=bytes(rng.integers(0, 256, int(rng.integers(0, 9)), dtype=np.uint8))),
                 mif.pnm(rng.integers(0, 256, (h, w, 3)), bool(k % 2), 255, comment=bool(k % 4 == 0)),
                 mif.pnm(rng.integers(0, 65536, (h, w, 3)), bool(k % 3), 65535),
                 g.hdr_encode(rng.uniform(0, 3, (h, w, 3)) ** int(rng.integers(1, 6)) * (1e-3 if k % 4 == 0 else 1.0), rle=bool(k % 2)),      # a Radiance file as an 8-bit texture
                 mif.psd(img, int(rng.integers(0, 7)), int(rng.choice([8, 16])), rle=False), mif.psd(img, int(rng.integers(1, 6)), 8, rle=True),
                 mif.pic(img, (int(rng.integers(0, 3)), int(rng.integers(0, 3))), alpha=bool(k % 3))]
        gp = rng.integers(0, 256, (int(rng.choice([2, 4, 8, 32, 256])), 3), dtype=np.uint8)
        gi = rng.integers(0, len(gp), (h, w))
        if k % 2:
            gi[:, : w // 2] = gi[:, :1]
        sw, sh = w + int(rng.integers(0, 5)), h + int(rng.integers(0, 5))
        files.append(mif.gif(gi, gp, screen=(sw, sh), at=(int(rng.integers(0, sw - w + 1)), int(rng.integers(0, sh - h + 1))), interlace=bool(k % 3 == 0), local=bool(k % 5 == 0),
                             transparent=int(rng.integers(0, len(gp))) if k % 4 == 0 else None, bgindex=int(rng.integers(0, len(gp))) if k % 2 else 0, comment=bool(k % 7 == 0)))
        # bit fields: three disjoint colour masks of 1..8 bits and an optional alpha mask, anywhere in the word
        bits = 32 if k % 2 else 16
        for _ in range(20):
            cnt = [int(rng.integers(1, 9 if bits == 32 else 6)) for _ in range(4)]
            if k % 3 == 0:
                cnt[3] = 0
            if sum(cnt) <= bits:
                break
        order = rng.permutation(4)
        masks, at = [0, 0, 0, 0], int(rng.integers(0, bits - sum(cnt) + 1))
        for c in order:
            masks[c] = ((1 << cnt[c]) - 1) << at
            at += cnt[c]
        if not (masks[0] == masks[1] == masks[2]):
            files.append(mif.bmp(img, f"fields{bits}", int(rng.choice([108, 124])), masks=tuple(masks)))
        nc = int(rng.integers(1, 257))
        pal = rng.integers(0, 256, (nc, 3), dtype=np.uint8)
        pb = int(rng.choice([8, 4, 1]))
        npal = max(min(nc, 1 << pb), 1)
        hd = int(rng.choice([12, 40, 108])) if npal > 4 else int(rng.choice([40, 108]))
        # (a 12-byte header makes the reference drop the palette's last four entries and read undefined memory for them: tests/misc_image_files.py)
        files.append(mif.bmp(rng.integers(0, npal - 4 if hd == 12 else npal, (h, w)), f"pal{pb}", hd, masks=pal[:npal], top_down=bool(k % 2) and hd != 12))
        files.append(mif.tga(rng.integers(0, nc, (h, w)), f"map{int(rng.choice([15, 16, 24, 32]))}", rle=bool(k % 2), top_down=bool(k % 3), palette=np.concatenate([pal, pal[:, :1]], 1),
                             index_bits=16 if nc > 255 or k % 7 == 0 else 8))
        for data in files:
            want, got = ora.ref_stb_decode(data), g.image_decode(data)
            assert got.shape == want.shape and np.array_equal(got, want), (k, data[:32])
            n += 1
    assert n > 1400


def test_files_cut_short(g, ora):
    """A byte past the end of the file reads as 0 in the reference's decoder: a BMP or a run-length TGA cut short decodes, with a black tail — here as there.  Raw TGA rows and
    the PNM body are read whole: the reference refuses a short PNM (and leaves short TGA rows undefined); both are refused here."""
    rng = np.random.default_rng(5)
    img = rng.integers(1, 256, (9, 11, 4), dtype=np.uint8)
    for data in (mif.bmp(img, "24"), mif.bmp(img, "32"), mif.bmp(img, "555"), mif.bmp(rng.integers(0, 16, (9, 11)), "pal4", masks=rng.integers(0, 256, (16, 3), dtype=np.uint8)),
                 mif.tga(img, "24", rle=True), mif.tga(img, "15", rle=False), mif.tga(rng.integers(0, 9, (9, 11)), "map24", rle=False, palette=rng.integers(0, 256, (9, 4), dtype=np.uint8))):
        whole = g.image_decode(data)
        for cut in (len(data) - 1, len(data) - 40, len(data) - 150):
            got = g.image_decode(data[:cut])
            assert got.shape == whole.shape and (cut > len(data) - 150 or not np.array_equal(got, whole))       # (the last bytes may be row padding)
            if ora.have_ref_stb():
                assert np.array_equal(got, ora.ref_stb_decode(data[:cut])), cut
    # a GIF cut short ends its image where the data ends (a block length of 0 terminates it); a raw PSD reads zeros
    for data in (mif.gif(rng.integers(0, 16, (30, 40)), rng.integers(0, 256, (16, 3), dtype=np.uint8), bgindex=3), mif.psd(img, 4, 8), mif.psd(img, 3, 16)):
        for cut in (len(data) - 2, len(data) - 60, len(data) * 2 // 3):
            got = g.image_decode(data[:cut])
            if ora.have_ref_stb():
                assert np.array_equal(got, ora.ref_stb_decode(data[:cut])), cut
    for data in (mif.tga(img, "24"), mif.tga(img, "8"), mif.pnm(img[:, :, :3]), mif.pnm(img[:, :, :3].astype(np.uint32) * 200, True, 65535)):
        with pytest.raises(ValueError, match="truncated"):
            g.image_decode(data[:-3])
    if ora.have_ref_stb():
        with pytest.raises(ValueError):
            ora.ref_stb_decode(mif.pnm(img[:, :, :3])[:-3])


def test_decoders_refuse_what_stb_refuses(g, ora):
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (4, 6, 4), dtype=np.uint8)
    b24 = bytearray(mif.bmp(img, "24"))

    def patched(data, at, fmt, v):
        d = bytearray(data); struct.pack_into(fmt, d, at, v); return bytes(d)
    pal256 = rng.integers(0, 256, (256, 3), dtype=np.uint8)
    bad = [patched(b24, 30, "<I", 1), patched(b24, 30, "<I", 2), patched(b24, 30, "<I", 4), patched(b24, 30, "<I", 3),      # run-length, JPEG, bit fields at 24 bits
           patched(b24, 26, "<H", 2), patched(b24, 14, "<I", 41), patched(b24, 10, "<I", 5000), patched(b24, 10, "<I", 20), patched(b24, 28, "<H", 20),   # planes, header size, offsets, 20 bits
           mif.bmp(img, "fields16", 40, masks=(0x1f, 0x1f, 0x1f, 0)), mif.bmp(img, "fields32", 108, masks=(0x1ff, 0xff << 9, 0xff << 17, 0)),
           mif.bmp(img, "fields32", 108, masks=(0, 0xff00, 0xff0000, 0)), mif.bmp(rng.integers(0, 9, (5, 9)), "pal8", 40, masks=pal256, palette_pad=8),
           patched(mif.bmp(rng.integers(0, 2, (5, 9)), "pal1", 40, masks=pal256[:2]), 28, "<H", 2),
           b"P6\n0 4\n255\n", b"P6\n4 0\n255\n", b"P5\n4 4\n70000\n" + b"\0" * 64, b"P6\n99999999999 4\n255\n", b"P6\n4 4\n255\n" + b"\0" * 10,
           mif.tga(rng.integers(0, 4, (3, 3)), "map24", palette=np.zeros((0, 4), np.uint8))]
    gpal = rng.integers(0, 256, (16, 3), dtype=np.uint8)
    gidx = rng.integers(0, 16, (5, 6))
    gf = mif.gif(gidx, gpal)
    at = gf.index(b"\x2c")
    ps = mif.psd(img, 4, 8)
    bad += [gf[:at] + b"\x3b",                                                       # no image at all
            gf[:at + 10] + bytes([13]) + gf[at + 11:],                               # an LZW code size of 13
            gf[:at] + b"\x99" + gf[at + 1:],                                         # an unknown block
            mif.gif(gidx, gpal, screen=(4, 4)),                                       # the image does not fit the screen
            gf[:10] + bytes([gf[10] & 0x7f]) + gf[13 + 48:],                          # neither a global nor a local colour table
            patched(ps, 4, ">H", 2), patched(ps, 24, ">H", 4), patched(ps, 22, ">H", 12), patched(ps, 12, ">H", 17), patched(ps, len(ps) - img.size - 2, ">H", 2),     # version, CMYK, 12 bits, 17 channels, compression 2
            mif.psd(img, 3, 8, rle=True)[:-40] + b"\x7f" * 40,                       # a literal run past the end of a channel
            mif.pic(img, (2, 2))[:-5], mif.pic(img, (0, 0))[:-1], patched(mif.pic(img, (2, 2)), 105, "B", 4), patched(mif.pic(img, (2, 2)), 106, "B", 3)]     # PIC: cut short (refused there and here), 4 bits, compression 3
    for k, data in enumerate(bad):
        with pytest.raises(ValueError):
            g.image_decode(data)
        if ora.have_ref_stb():
            with pytest.raises(ValueError):
                ora.ref_stb_decode(data)
    for data in (b"", b"BM", b"P6", b"P7\n1 1\n255\n\0\0\0", bytes(64), mif.tga(img, "24")[:2] + b"\x07" + mif.tga(img, "24")[3:]):
        with pytest.raises(ValueError):
            g.image_decode(data)


def test_gltf_images_in_these_formats_reach_the_context(pbr, tmp_path):
    """a glTF whose image is a BMP, a TGA or a PPM file (what the reference's stb decodes as well as PNG / JPEG): the context receives the texels of the PNG version"""
    import base64
    sc = pbr.scene
    rng = np.random.default_rng(3)
    tex = rng.integers(0, 256, (8, 8, 4), dtype=np.uint8)
    tri = np.zeros(3, sc.MESH_VERTEX)
    tri["position"] = [(0, 0, 0), (1, 0, 0), (0, 1, 0)]; tri["normal"] = (0, 0, 1); tri["tangent"] = (1, 0, 0, 1); tri["texCoords"] = [(0, 0), (1, 0), (0, 1)]
    d = sc.SceneDesc([sc.Material(tex_color=0)], [sc.MeshDesc(tri, np.arange(3, dtype=np.uint32), 0)], [sc.InstanceDesc(0)], sc.CameraDesc((0, 0, 5), (0, 0, 0), 1.0, 1.0), textures=[tex])
    glb = str(tmp_path / "a.glb")
    pbr.gltf.write_glb(d, glb)
    raw = open(glb, "rb").read()
    jlen = struct.unpack("<I", raw[12:16])[0]
    doc = json.loads(raw[20: 20 + jlen])
    (tmp_path / "a.bin").write_bytes(raw[20 + jlen + 8:])
    doc["buffers"][0]["uri"] = "a.bin"
    opaque = tex.copy(); opaque[:, :, 3] = 255
    gp = rng.integers(0, 256, (16, 3), dtype=np.uint8); gi = rng.integers(0, 16, (8, 8))
    gimg = np.concatenate([gp[gi], np.full((8, 8, 1), 255, np.uint8)], 2)
    for name, data, want in (("t.bmp", mif.bmp(tex, "32", 124), tex), ("t24.bmp", mif.bmp(tex, "24"), opaque), ("t.tga", mif.tga(tex, "32", rle=True), tex), ("t.ppm", mif.pnm(tex[:, :, :3]), opaque),
                             ("t.gif", mif.gif(gi, gp, interlace=True), gimg), ("t.psd", mif.psd(opaque, 4, 8, rle=True), opaque), ("t.pic", mif.pic(tex, (2, 1)), tex)):
        (tmp_path / name).write_bytes(data)
        for uri in (name, "data:application/octet-stream;base64," + base64.b64encode(data).decode()):
            j = json.loads(json.dumps(doc))
            j["images"][0] = {"uri": uri}
            q = str(tmp_path / "v.gltf")
            open(q, "w").write(json.dumps(j))
            pt = pbr.PathTracer(pbr.DEVICE_NONE)
            pbr.gltf.load_into(pt, q, camera=d.camera)
            mats, texs = pt.description()
            assert len(texs) == 1 and np.array_equal(texs[0], want), name
