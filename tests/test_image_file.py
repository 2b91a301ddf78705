"""Row f1, textures: csrc/image_file.cpp (spcbpt_image_load and the scene loaders' texture path) against the reference's
vendored stb_image v2.27 -- stbi_load(path, &w, &h, &c, STBI_rgb_alpha) as scene_shift.cpp:35-40 calls it.  BIT-EXACT:
  * 45 authored PNG / JPEG fixtures (tests/golden/images, written by tests/golden/make_images.py) against stb's outputs in
    tests/golden/ref_images.npz: every PNG colour type / bit depth / tRNS / Adam7 / filter / deflate block type; JPEG baseline
    4:4:4, 4:2:0, 4:2:2, 4:4:0, 4:1:1, greyscale, restart intervals, 16-bit quantisation tables, progressive with successive
    approximation (DC and AC refinement scans), odd sizes down to 2x3 and 1 pixel wide;
  * every texture the reference ships (43 files of data/house/textures: baseline and progressive JPEG, PNG) by SHA-256 of
    stb's output, when /root/reference is present (the build container);
  * stb itself, live, when oracle/_ref is built."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), "golden")
IMAGES = os.path.join(G, "images")
SHIPPED = "/root/reference/src/data/house/textures"


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(G, "ref_images.npz"))


def test_fixtures_decode_like_stb_image(hip_lib, pkg, golden):
    names = sorted(k[len("fixture/"):] for k in golden.files if k.startswith("fixture/"))
    assert names == sorted(os.listdir(IMAGES)) and len(names) == 45
    for n in names:
        got = pkg.api.image_load(os.path.join(IMAGES, n))
        want = golden["fixture/" + n]
        assert got.shape == want.shape, n
        assert np.array_equal(got, want), (n, int(np.abs(got.astype(int) - want.astype(int)).max()), float((got != want).mean()))


def test_shipped_textures_decode_like_stb_image(hip_lib, pkg, golden):
    if not os.path.isdir(SHIPPED):
        pytest.skip("the reference checkout is not present on this machine")
    names = sorted(k[len("shipped/"):] for k in golden.files if k.startswith("shipped/"))
    assert len(names) == 43
    for n in names:
        px = pkg.api.image_load(os.path.join(SHIPPED, n))
        digest = hashlib.sha256(np.array(px.shape, np.int64).tobytes() + px.tobytes()).digest()
        assert digest == golden["shipped/" + n].tobytes(), n


def test_against_stb_image_live(hip_lib, pkg, ob):
    r = ob.ref_lib()
    if r is None:
        pytest.skip("oracle/_ref not built here")
    for n in ("jp420.jpg", "j420_rst.jpg", "pal4.png", "rgba16.png"):
        p = os.path.join(IMAGES, n)
        w, h = C.c_int(), C.c_int()
        assert r.ref_image_load(p.encode(), C.byref(w), C.byref(h), None, 0) == 0
        ref = np.zeros((h.value, w.value, 4), np.uint8)
        r.ref_image_load(p.encode(), C.byref(w), C.byref(h), C.c_void_p(ref.ctypes.data), ref.nbytes)
        assert np.array_equal(pkg.api.image_load(p), ref)


def test_errors_and_truncation(hip_lib, pkg, tmp_path):
    with pytest.raises(pkg.SpcbptError, match="-7"):
        pkg.api.image_load(str(tmp_path / "missing.png"))
    (tmp_path / "x.bin").write_bytes(b"GIF89a" + bytes(64))
    with pytest.raises(pkg.SpcbptError, match="-7"):
        pkg.api.image_load(str(tmp_path / "x.bin"))
    for n in ("rgb8.png", "j444.jpg", "jp444.jpg"):
        data = open(os.path.join(IMAGES, n), "rb").read()
        for cut in (5, len(data) // 3, len(data) - 9):
            p = tmp_path / ("cut_" + n)
            p.write_bytes(data[:cut])
            try:                                   # a damaged file is an error or a partial image, never a crash
                pkg.api.image_load(str(p))
            except pkg.SpcbptError:
                pass
    w, h = C.c_int(), C.c_int()
    small = np.zeros(8, np.uint8)
    assert hip_lib.spcbpt_image_load(os.path.join(IMAGES, "rgb8.png").encode(), C.byref(w), C.byref(h), small.ctypes.data, small.nbytes) == -6
    assert hip_lib.spcbpt_image_load(None, C.byref(w), C.byref(h), None, 0) == -1


def test_scene_loader_takes_jpeg_and_png_textures(hip_lib, pkg, tmp_path):
    """`.scene` route: albedoTex naming a JPEG / PNG now loads (the reference's shipped scene names 30 JPEGs)."""
    import shutil
    root = tmp_path / "data"
    (root / "s").mkdir(parents=True)
    for n in ("j420.jpg", "rgb8.png"):
        shutil.copy(os.path.join(IMAGES, n), root / "s" / n)
    (root / "s" / "quad.obj").write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nf 1/1 2/2 3/3 4/4\n")
    scene = ""
    for k, tex in enumerate(("j420.jpg", "rgb8.png", "nothing.jpg")):
        scene += f"material m{k}\n{{\n   color 0.5 0.5 0.5\n   albedoTex s/{tex}\n}}\n\nmesh\n{{\n    file s/quad.obj\n    material m{k}\n}}\n\n"
    scene += "light l0\n{\n   type Quad\n   position 0 2 0\n   u 1 2 0\n   v 0 2 1\n   emission 5 5 5\n}\n"
    (root / "s" / "t.scene").write_text(scene)
    sc, warn = pkg.load_scene_file(str(root / "s" / "t.scene"), str(root))
    assert len(sc.textures) == 2 and "nothing.jpg" in warn and "j420.jpg" not in warn
    g = np.load(os.path.join(G, "ref_images.npz"))
    for tex, name in zip(sc.textures, ("j420.jpg", "rgb8.png")):
        want = g["fixture/" + name]
        assert np.array_equal(np.asarray(tex).reshape(want.shape), want)
