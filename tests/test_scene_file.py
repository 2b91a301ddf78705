"""`.scene` + OBJ ingestion: write a scene in the reference's syntax, read it back with the C++ loader,
and parse the reference's shipped house scene where it lies (skipped when /root/reference is absent)."""
import os

import numpy as np
import pytest

HOUSE = "/root/reference/src/data/house/house_uvrefine2.scene"


def _tri_set(scene):
    t = scene.vertices[scene.indices.reshape(-1)].reshape(-1, 9)
    return t[np.lexsort(t.T[::-1])]


@pytest.mark.parametrize("name,kw", [("cornell_box", {}), ("bedroom", {"target_tris": 3000, "tex_size": 16})])
def test_scene_round_trip(hip_lib, pkg, tmp_path, name, kw):
    scene = getattr(pkg.scenes, name)(**kw)
    scene.materials[1]["brdf"] = 1     # `brdf <int>` (sceneLoader.cpp:107) travels with the material
    path = pkg.scenes.write_scene(scene, str(tmp_path), name)
    loaded, warn = pkg.load_scene_file(path, str(tmp_path))
    assert warn == "", warn
    assert loaded.indices.shape == scene.indices.shape
    assert len(loaded.materials) == len(scene.materials) and len(loaded.lights) == len(scene.lights)
    np.testing.assert_allclose(_tri_set(loaded), _tri_set(scene), rtol=0, atol=1e-6)
    for a, b in zip(loaded.materials, scene.materials):
        np.testing.assert_allclose(a["color"], b["color"], rtol=1e-6)
        assert abs(a["roughness"] - b["roughness"]) < 1e-6 and abs(a["metallic"] - b["metallic"]) < 1e-6
        assert (a["albedo_tex"] > 0) == (b.get("albedo_tex", 0) > 0)
        assert a["brdf"] == b.get("brdf", 0)
    for a, b in zip(loaded.lights, scene.lights):
        for k in ("position", "u", "v", "emission"):
            np.testing.assert_allclose(a[k], b[k], rtol=1e-5, atol=1e-6)
        assert a["div_level"] == b["div_level"]
    np.testing.assert_allclose(loaded.camera["eye"], scene.camera["eye"], rtol=1e-6)
    assert abs(loaded.camera["fov"] - scene.camera["fov"]) < 1e-5
    for ta, tb in zip(loaded.textures, scene.textures):
        assert np.array_equal(ta[..., :3], tb[..., :3]) and (ta[..., 3] == 255).all()
    # per-triangle material follows the k-th mesh / k-th material rule
    cen = lambda s: s.vertices[s.indices].mean(axis=1)
    key = lambda s: np.round(np.concatenate([cen(s), s.tri_material[:, None]], 1), 4)
    assert set(map(tuple, key(loaded))) == set(map(tuple, key(scene)))


def test_loaded_scene_renders_like_the_generated_one(hip_lib, ob, pkg, tmp_path):
    scene = pkg.scenes.cornell_box()
    path = pkg.scenes.write_scene(scene, str(tmp_path), "cornell")
    loaded, _ = pkg.load_scene_file(path, str(tmp_path))
    imgs = []
    for s in (scene, loaded):
        o = ob.Oracle(s)
        c = s.camera
        o.set_camera_lookat(c["eye"], c["lookat"], c["up"], c["fov"], 1.0)
        o.resize(32, 32)
        for f in range(4):
            o.launch("pt", f)
        imgs.append(o.read_accum()[..., :3])
    assert np.abs(imgs[0] - imgs[1]).max() < 1e-4


def test_grammar_quirks(hip_lib, pkg, tmp_path):
    """Comment lines, back-slash paths, a block without closing brace at EOF, a non-Quad light, an unknown texture format."""
    d = tmp_path / "q"
    d.mkdir()
    (d / "m.obj").write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nf 1/1 2/1 3/1 4/1\nf -4//1 -3//1 -2//1\n")
    (tmp_path / "q.scene").write_text(
        "#material Commented\n#{\n# color 9 9 9\n#}\nmaterial A\n{\n   color 0.1 0.2 0.3\n   roughness 0.25\n   albedoTex q/none.jpg\n   brdf 0x2\n}\n"
        "mesh\n{\n    file q\\m.obj\n    material A\n}\n"
        "light\n{\n  position 0 0 0\n  type Sphere\n  radius 1\n}\n"
        "light\n{\n  position 0 2 0\n  v1 1 2 0\n  v2 0 2 1\n  emission 5 5 5\n divLevel 3\n  type Quad\n")
    s, warn = pkg.load_scene_file(str(tmp_path / "q.scene"), str(tmp_path))
    # quad fan (2 triangles, 4 v/vt vertices) + a triangle with negative indices and no vt (3 more distinct v/vt pairs)
    assert s.indices.shape[0] == 3 and s.vertices.shape[0] == 7
    assert len(s.materials) == 1 and abs(s.materials[0]["roughness"] - 0.25) < 1e-7 and s.materials[0]["albedo_tex"] == 0
    assert s.materials[0]["brdf"] == 2                 # `%i` reads 0x2; any nonzero value is `true` in MaterialData::Pbr (scene_shift.cpp:75)
    assert len(s.lights) == 1 and s.lights[0]["div_level"] == 3 and tuple(s.lights[0]["u"]) == (1.0, 0.0, 0.0)
    assert "Sphere" in warn and "none.jpg" in warn


@pytest.mark.skipif(not os.path.exists(HOUSE), reason="reference data not present")
def test_reference_house_scene_parses(hip_lib, pkg):
    s, warn = pkg.load_scene_file(HOUSE, "/root/reference/src/data")
    assert len(s.lights) == 2 and all(l["div_level"] == 10 for l in s.lights)
    np.testing.assert_allclose(s.lights[0]["position"], (5.5, -1, 7))
    np.testing.assert_allclose(s.lights[0]["u"], (-30, 0, 0))
    np.testing.assert_allclose(s.lights[0]["emission"], (70, 55, 45))
    np.testing.assert_allclose(s.camera["eye"], (-0.813158, 5.627658, -7.363544), rtol=1e-6)
    assert abs(s.camera["fov"] - 60) < 1e-6
    assert len(s.materials) >= 25 and s.indices.shape[0] > 50000     # 30 meshes, ~67 k faces present
    # material `Glass` (house_uvrefine2.scene:129-136: roughness 0, `brdf 1`) is the one flagged material of the shipped file, and
    # the only mesh block that names it is commented out there (lines 306-310).  Materials reach the renderer per mesh block
    # (scene_shift.cpp:235), so as shipped no flagged material is live; test_house_glass_block_reaches_the_renderer un-comments it.
    assert not any(m["brdf"] for m in s.materials)
    assert "could not be read" in warn                                 # three referenced OBJs were stripped from the checkout
    assert ".jpg" in warn or ".png" in warn                           # stb_image formats are not decoded here


@pytest.mark.skipif(not os.path.exists(HOUSE), reason="reference data not present")
def test_house_glass_block_reaches_the_renderer(hip_lib, pkg, tmp_path):
    """The shipped scene's `material Glass` block verbatim (roughness 0, `brdf 1`) on a mesh that names it -- the commented-out
    block of house_uvrefine2.scene:306-310 restored: the flag arrives in spcbpt_material::brdf (sceneLoader.cpp:107 ->
    scene_shift.cpp:75 `mtl.pbr.brdf = p.brdf`)."""
    lines = open(HOUSE, "rb").read().decode("latin-1").splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("material Glass"))
    end = next(i for i in range(start, len(lines)) if "}" in lines[i])
    (tmp_path / "m.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    (tmp_path / "g.scene").write_text("\n".join(lines[start:end + 1]) + "\nmesh\n{\n    file m.obj\n    material Glass\n}\n")
    s, warn = pkg.load_scene_file(str(tmp_path / "g.scene"), str(tmp_path))
    assert warn == "", warn
    (m,) = s.materials
    assert m["brdf"] == 1 and m["roughness"] == 0.0 and m["metallic"] == 0.0
    np.testing.assert_allclose(m["color"], (0.8, 0.8, 0.8), rtol=1e-6)


# ---- pins against the reference's own vendored loaders (oracle/_ref; vectors in tests/golden/ref_loaders.npz) ------------
_GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_loaders.npz")
_OBJ_CASES = ["tri_quad_fan", "uv_shared_and_split", "negative_and_normals", "groups_and_objects", "mixed_uv_then_none",
              "comments_blank_tabs", "usemtl_splits_shapes", "number_formats", "zero_index_and_short_faces"]


def _scene_with(tmp_path, mesh_rel=None, tex_rel=None):
    lines = ["material m0", "{", " color 0.5 0.5 0.5", " roughness 0.5", " metallic 0"]
    if tex_rel: lines.append(f" albedoTex {tex_rel}")
    lines.append("}")
    if mesh_rel: lines += ["mesh", "{", f" file {mesh_rel}", " material m0", "}"]
    p = tmp_path / "s.scene"
    p.write_text("\n".join(lines) + "\n")
    return str(p)


@pytest.mark.parametrize("case", _OBJ_CASES)
def test_obj_reader_matches_reference_tinyobj(hip_lib, pkg, tmp_path, case):
    """The OBJ reader of csrc/scene_file.cpp against tinyobj::LoadObj as scene_shift.cpp:187-250 consumes it: same vertex
    de-duplication (one vertex per distinct v/vt/vn triple, restarted per shape), same fan triangulation, same index order,
    missing UVs zero — bit-exact.  Shapes are concatenated with their index base added (one triangle soup)."""
    g = np.load(_GOLD)
    (tmp_path / "m.obj").write_bytes(g[case + "__text"].tobytes())
    loaded, warn = pkg.load_scene_file(_scene_with(tmp_path, mesh_rel="m.obj"), str(tmp_path))
    assert warn == "", warn
    base = np.concatenate([[0], np.cumsum(g[case + "__shape_v"])[:-1]])
    idx = g[case + "__idx"].astype(np.int64).copy()
    o = 0
    for b, n in zip(base, g[case + "__shape_i"]):
        idx[o:o + n] += b
        o += n
    assert np.array_equal(loaded.vertices, g[case + "__pos"])
    assert np.array_equal(loaded.texcoords, g[case + "__uv"])
    assert np.array_equal(loaded.indices.reshape(-1).astype(np.int64), idx)


@pytest.mark.parametrize("k", [0, 1, 2])
def test_ppm_texture_matches_reference_stb_image(hip_lib, pkg, tmp_path, k):
    """Binary PPM decode against stbi_load(..., STBI_rgb_alpha) as Material_shift calls it (scene_shift.cpp:38-40)."""
    g = np.load(_GOLD)
    (tmp_path / "t.ppm").write_bytes(g[f"ppm{k}__file"].tobytes())
    (tmp_path / "m.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    loaded, warn = pkg.load_scene_file(_scene_with(tmp_path, mesh_rel="m.obj", tex_rel="t.ppm"), str(tmp_path))
    assert warn == "", warn
    assert len(loaded.textures) == 1 and np.array_equal(loaded.textures[0], g[f"ppm{k}__rgba"])
