"""The Radiance RGBE (.hdr) decoder (host/image_io.hpp decode_hdr, ptc_hdr_decode_rgb32f) against the REFERENCE's own decoder: oracle/_ref is the reference's vendored
stb_image translation unit compiled where it lies (stbi_loadf_from_memory(..., 3): what its image path yields for such a file); the committed fixtures
tests/golden/hdr_* were generated from it (tests/golden/make_hdr_golden.py) and pin the decoder where the reference checkout is absent (the GPU box)."""
import glob
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def g(pbr):
    return pbr.gltf


def test_hdr_decoder_equals_the_fixtures_of_the_reference_stb(g):
    files = sorted(glob.glob(os.path.join(GOLD, "hdr_*.hdr")))
    assert len(files) >= 4
    for f in files:
        want = np.load(f[:-4] + ".npy")
        got = g.hdr_decode(open(f, "rb").read())
        assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32)), f


def test_hdr_decoder_equals_reference_stb_on_random_files(g, ora):
    if not ora.have_ref_stb():
        pytest.skip("oracle/_ref not built (no reference checkout)")
    rng = np.random.default_rng(7)
    for k in range(60):
        h, w = int(rng.integers(1, 40)), int(rng.integers(1, 70))
        img = rng.uniform(0, 4, (h, w, 3)) ** rng.integers(1, 6)
        if k % 3 == 0:
            img = np.repeat(img[:, ::4], 4, 1)[:, :w]                      # long runs
        if k % 5 == 0:
            img[rng.random((h, w)) < 0.2] = 0.0                            # black pixels: exponent byte 0
        data = g.hdr_encode(img, rle=bool(k % 2), magic="#?RGBE" if k % 7 == 0 else "#?RADIANCE")
        want, got = ora.ref_stb_decode_float(data), g.hdr_decode(data)
        assert got.shape == want.shape == (h, w, 3) and np.array_equal(got.view(np.uint32), want.view(np.uint32)), k
        # the quantisation of the shared-exponent format: a pixel comes back within 1/128 of its largest channel
        assert np.all(np.abs(got - img) <= img.max(2, keepdims=True) / 128.0 + 1e-30)


def test_hdr_decoder_refuses_what_stb_refuses(g, ora):
    good = g.hdr_encode(np.ones((4, 16, 3)))
    bad = [b"", b"#?RADIANCE\n", good[:40], good[:-5], good.replace(b"32-bit_rle_rgbe", b"32-bit_rle_xyze"), good.replace(b"-Y 4 +X 16", b"+Y 4 +X 16"),
           good.replace(b"#?RADIANCE", b"#?RADIANCF"), good.replace(b"-Y 4 +X 16", b"-Y 4 +X 17")]
    for k, data in enumerate(bad):
        with pytest.raises(ValueError):
            g.hdr_decode(data)
        if ora.have_ref_stb() and data:
            with pytest.raises(ValueError):
                ora.ref_stb_decode_float(data)


@pytest.mark.gpu
def test_cli_reads_a_radiance_environment(pbr, tmp_path):
    """ptc_render --env file.hdr: the reader is wired into the host CLI (extension .hdr / .pic; PFM otherwise): a corrupt file is reported as such, a good one lights
    the scene — the same image, bit for bit, as the same floats handed over as a PFM."""
    exe = os.path.join(ROOT, "physically-based-renderer_amd", "lib", "ptc_render")
    img = np.ones((8, 16, 3)) * [1.0, 2.0, 3.0]
    img[:3] *= 4.0
    data = pbr.gltf.hdr_encode(img)
    good = tmp_path / "sky.hdr"
    good.write_bytes(data)
    dec = pbr.gltf.hdr_decode(data)
    pfm = tmp_path / "sky.pfm"
    with open(pfm, "wb") as f:
        f.write(b"PF\n16 8\n-1.0\n")
        f.write(dec[::-1].astype("<f4").tobytes())
    bad = tmp_path / "bad.hdr"
    bad.write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 8 +X 16\n\x02\x02")
    glb = str(tmp_path / "s.glb")
    pbr.gltf.write_glb(pbr.scenes.by_name("textured_objects"), glb)
    common = ["--gltf", glb, "--width", "24", "--height", "24", "--spp", "2"]
    r = subprocess.run([exe, "--scene", "cornell", "--env", str(good)], capture_output=True, text=True, timeout=120)     # the built-in scenes take no environment: said, not ignored
    assert r.returncode != 0 and "--gltf" in r.stderr, r.stdout + r.stderr
    r = subprocess.run([exe, *common, "--env", str(bad), "--out", str(tmp_path / "o.pfm")], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "HDR" in (r.stdout + r.stderr), r.stdout + r.stderr
    outs = []
    for env, name in ((good, "a.pfm"), (pfm, "b.pfm")):
        r = subprocess.run([exe, *common, "--env", str(env), "--out", str(tmp_path / name)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(open(tmp_path / name, "rb").read())
    assert outs[0] == outs[1] and len(outs[0]) > 24 * 24 * 12
