"""Row f4, environment map, host side (no GPU): the Radiance .hdr reader (csrc/env_file.cpp <- HDRLoader, scene_shift.cpp:334-500)
against an independent decoder of the format, the `.scene` route (env_file of the cameraSetting block) and the scene box the
reference derives sky.center / sky.r from (first third of every OBJ shape's vertices + the light quads)."""
import os

import numpy as np
import pytest


def _decode_hdr(path):
    """An independent reader of the same files: header by hand, new-style RLE per channel, value = (m + 0.5) 2^(e - 136) / exposure."""
    raw = open(path, "rb").read()
    head, _, rest = raw.partition(b"\n\n")
    assert head.startswith(b"#?RADIANCE")
    exposure = 1.0
    for line in head.split(b"\n"):
        if line.startswith(b"EXPOSURE="):
            exposure = float(line[9:])
    res, _, data = rest.partition(b"\n")
    t = res.split()
    assert t[0] == b"-Y" and t[2] == b"+X"
    h, w = int(t[1]), int(t[3])
    out = np.zeros((h, w, 4), np.uint8)
    p = 0
    for y in range(h):
        if 8 <= w <= 0x7fff and data[p] == 2 and data[p + 1] == 2 and not (data[p + 2] & 0x80):
            assert (data[p + 2] << 8 | data[p + 3]) == w
            p += 4
            for ch in range(4):
                x = 0
                while x < w:
                    c = data[p]; p += 1
                    if c > 128:
                        out[y, x:x + c - 128, ch] = data[p]; p += 1; x += c - 128
                    else:
                        out[y, x:x + c, ch] = np.frombuffer(data[p:p + c], np.uint8); p += c; x += c
        else:
            out[y] = np.frombuffer(data[p:p + 4 * w], np.uint8).reshape(w, 4); p += 4 * w
    e = out[..., 3].astype(np.int32)
    s = np.where(e == 0, 0.0, np.ldexp(1.0, e - 136)).astype(np.float32) * np.float32(1.0 / exposure)
    rgb = (out[..., :3].astype(np.float32) + np.float32(0.5)) * s[..., None]
    rgb[e == 0] = 0
    return rgb


@pytest.mark.parametrize("rle,w,h,exposure", [(True, 64, 32, None), (False, 64, 32, None), (True, 7, 5, None), (True, 200, 3, 2.5), (True, 300, 2, None)])
def test_hdr_reader_matches_an_independent_decoder(hip_lib, pkg, tmp_path, rle, w, h, exposure):
    rng = np.random.default_rng(w * 131 + h)
    img = rng.gamma(0.7, 2.0, (h, w, 3)).astype(np.float32)
    img[rng.random((h, w)) < 0.1] = 0.0                     # black texels: exponent byte 0
    img[:, : w // 3] = img[:, :1]                            # long runs for the RLE branch
    if w == 300:
        img[:, :] = 0.37                                     # one run longer than 127 per channel
    f = str(tmp_path / "t.hdr")
    pkg.scenes.write_hdr(f, img, rle=rle, exposure=exposure)
    got = pkg.api.hdr_load(f)
    want = _decode_hdr(f)
    assert got.shape == (h, w, 4) and np.array_equal(got[..., :3], want) and (got[..., 3] == 0).all()
    lim = np.maximum(img.max(-1, keepdims=True), 1e-30) * (1.0 / (exposure or 1.0))
    assert (np.abs(got[..., :3] * 1.0 - img / (exposure or 1.0)) <= lim / 128 + 1e-30).all()     # RGBE: 8 bits of the largest channel


def test_hdr_reader_refuses_what_hdrloader_refuses(hip_lib, pkg, tmp_path):
    good = str(tmp_path / "g.hdr")
    pkg.scenes.write_hdr(good, np.ones((4, 8, 3), np.float32))
    raw = open(good, "rb").read()
    for name, data in (("magic", raw.replace(b"#?RADIANCE", b"#?RGBE")), ("xyze", raw.replace(b"32-bit_rle_rgbe", b"32-bit_rle_xyze")),
                       ("order", raw.replace(b"-Y 4 +X 8", b"+Y 4 +X 8")), ("short", raw[:-7])):
        f = str(tmp_path / (name + ".hdr"))
        open(f, "wb").write(data)
        with pytest.raises(pkg.SpcbptError):
            pkg.api.hdr_load(f)
    with pytest.raises(pkg.SpcbptError):
        pkg.api.hdr_load(str(tmp_path / "missing.hdr"))


def test_scene_file_route_carries_the_environment_and_the_reference_scene_box(hip_lib, pkg, tmp_path):
    sc = pkg.scenes.courtyard()
    path = pkg.scenes.write_scene(sc, str(tmp_path), "yard")
    assert "env_file yard/sky.hdr" in open(path).read()
    s2, warn = pkg.load_scene_file(path, str(tmp_path))
    assert warn == "" and s2.environment is not None
    want = _decode_hdr(str(tmp_path / "yard" / "sky.hdr"))
    assert np.array_equal(s2.environment["rgba"][..., :3], want)
    # sky.center / sky.r: the box over the FIRST THIRD of every shape's vertices (get_aabb(std::vector<float>), scene_shift.cpp:21-32)
    # and the four corners of every light quad
    lo, hi = np.full(3, 1e30), np.full(3, -1e30)
    v0 = 0
    for k in range(len(s2.materials)):                 # write_scene: one OBJ (one shape) per material, vertices in file order
        tris = s2.indices[s2.tri_material == k]
        nv = len(np.unique(tris))
        pts = s2.vertices[v0:v0 + nv][: (nv + 2) // 3]
        lo, hi = np.minimum(lo, pts.min(0)), np.maximum(hi, pts.max(0))
        v0 += nv
    for l in s2.lights:
        p, u, v = (np.asarray(l[x], np.float32) for x in ("position", "u", "v"))
        for c in (p, p + u, p + v, p + u + v):
            lo, hi = np.minimum(lo, c), np.maximum(hi, c)
    assert np.allclose(s2.environment["center"], 0.5 * (lo + hi), atol=1e-6)
    assert abs(s2.environment["radius"] - np.linalg.norm(lo - hi)) < 1e-5
    # a scene that names a missing file: a warning, no environment
    txt = open(path).read().replace("yard/sky.hdr", "yard/none.hdr")
    open(path, "w").write(txt)
    s3, warn = pkg.load_scene_file(path, str(tmp_path))
    assert s3.environment is None and "not loaded" in warn
