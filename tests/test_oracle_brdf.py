"""MaterialData::Pbr::brdf (`brdf <int>` of a .scene material block), oracle only (CPU).

The survey called the `#ifdef BRDF` branches of Tracer::Eval / Pdf dead (they are: the macro is never defined) -- the flag itself is
read by five un-guarded ternaries that divide the BSDF value by |n . out|:
    hit_program.cu:286 / 384   NextVertex.flux of the eye / light walk
    raygen.cu:271 / 278        fa / fb of connectVertex_SPCBPT
    rmis.h:105                 getFluxMultiplier of the recursive-MIS recursions
and by nothing else: "pt" (hit_program.cu:439-552), the full-path-MIS raygen (raygen.cu:445-606), the training records
(cuProg.h:1193-1282: the only mention is commented out) and direction_connect_ZGCBPT (raygen.cu:234-252) evaluate the plain BSDF.
What the restated flag does is checked here against what the lines say, and against first principles: with the five divisions in
place the recursive-MIS weights of a flagged path are still the balance heuristic over its strategies (and stop being one as
soon as any of the five is left out -- tried by hand with rmis.h:105: the depth-3 partition breaks by up to 40 %)."""
import numpy as np
import pytest

from tests.parity_util import cornell_with_flagged_box, minimal_tuple


def _oracle(ob, scene, w=32, h=32, lt=(3000, 64, 1)):
    o = ob.Oracle(scene)
    cam = scene.camera
    o.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], w / h)
    o.resize(w, h)
    o.set_light_trace(*lt)
    return o


@pytest.mark.parametrize("which", ["all_flagged", "glass_box", "room_mixed"])
def test_rmis_partition_holds_with_flagged_materials(pkg, ob, which):
    """tests/test_oracle_rmis_partition.py with `brdf 1` materials: every strategy's weight equals its rate / sum of rates."""
    if which == "all_flagged":
        scene = pkg.scenes.cornell_box()
        for m in scene.materials: m["brdf"] = 1
    elif which == "glass_box":
        scene = cornell_with_flagged_box(pkg, roughness=0.3, flag_walls=False)
    else:
        scene = pkg.scenes.bedroom(target_tris=3000, tex_size=16)
        for k in (0, 1, 3, 5, 7): scene.materials[k]["brdf"] = 1    # textured floor, walls, textured wood, 0.05-roughness metal, plastic
    o = _oracle(ob, scene, 64, 64)
    o.set_subspace(*minimal_tuple(o, 2))
    for depth in (1, 2, 3, 4):
        w, truth = o.quad_partition(depth, 600)
        assert len(w) >= 200, (which, depth, len(w))
        ok = np.abs(w[:, 0] - 1) < 1e-3
        assert ok.mean() > 0.99, (which, depth, ok.mean())
        d = np.abs(w[ok, 1:] - truth[ok])
        assert d.max() < 2e-3 and d.mean() < 2e-5, (which, depth, d.max(), d.mean())


def test_flag_divides_next_vertex_flux_by_the_cosine_and_nothing_else(pkg, ob):
    """hit_program.cu:286: the same eye step with the flag off and on -- NextVertex.flux is multiplied by 1 / |N . dir| (operator/(float3,
    float) multiplies by the reciprocal: bit-exact), every other output and the random-number state are untouched."""
    outs = []
    for flag in (0, 1):
        scene = pkg.scenes.cornell_box()
        for m in scene.materials: m["brdf"] = flag
        o = _oracle(ob, scene)
        o.set_subspace(*minimal_tuple(o, 1))
        cam = scene.camera
        rng = np.random.default_rng(3)
        n = 4000
        U, V, W = pkg.camera_frame(np.array(cam["eye"], np.float32), np.array(cam["lookat"], np.float32), np.array(cam["up"], np.float32),
                                   np.float32(cam["fov"]), np.float32(1.0))
        d = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
        dirs = d[:, :1] * U[None, :] + d[:, 1:] * V[None, :] + W[None, :]
        dirs = np.ascontiguousarray(dirs / np.linalg.norm(dirs, axis=1, keepdims=True), dtype=np.float32)
        rec = np.zeros(n, ob.EYE_STEP_IN_DTYPE)
        rec["last"]["position"] = cam["eye"]; rec["last"]["normal"] = dirs; rec["last"]["flux"] = 1.0
        rec["last"]["last_position"] = cam["eye"]; rec["last"]["pdf"] = 1.0; rec["last"]["single_pdf"] = 1.0
        rec["next_single_pdf"] = 1.0
        rec["seed"] = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
        rec["dir"] = dirs
        outs.append(o.eye_step(rec))
    off, on = outs
    surf = off["kind"] == 1
    assert surf.sum() > 1500
    for k in off.dtype.names:
        if k != "next_flux":
            assert off[k].tobytes() == on[k].tobytes(), k
    cos = np.abs((off["mid"]["normal"][surf] * off["dir"][surf]).sum(1, dtype=np.float32))   # not the oracle's rounding of the dot product: compare loosely
    live = (off["next_flux"][surf] != 0).any(1)
    ratio = on["next_flux"][surf][live] / off["next_flux"][surf][live]
    np.testing.assert_allclose(ratio, (1.0 / cos[live])[:, None] * np.ones(3), rtol=2e-6)
    assert (on["next_flux"][surf][~live] == 0).all() or np.isnan(on["next_flux"][surf][~live]).any()   # 0 / |cos| stays 0 (0 / 0 -> NaN: rejected later, as upstream)


def test_pt_ignores_the_flag_and_spcbpt_does_not(pkg, ob):
    """A quirk that follows from the lines above (documented in DESIGN.md): "pt" has no such division, so PT and SPCBPT converge
    to different images on a flagged material -- f / |n . out| is a brighter BSDF than f."""
    plain = pkg.scenes.cornell_box()
    flagged = pkg.scenes.cornell_box()
    for m in flagged.materials: m["brdf"] = 1
    imgs = {}
    for name, scene in (("plain", plain), ("flagged", flagged)):
        o = _oracle(ob, scene, 32, 32, lt=(2000, 64, 1))
        o.set_subspace(*minimal_tuple(o, 2))
        for f in range(8):
            o.launch("pt", f)
        imgs[name, "pt"] = o.read_accum()[..., :3].copy()
        o.clear_accum()
        for f in range(8):
            o.render_frame("SPCBPT_eye", f)
        imgs[name, "spcbpt"] = o.read_accum()[..., :3].copy()
    assert imgs["plain", "pt"].tobytes() == imgs["flagged", "pt"].tobytes()
    m = {k: float(v.mean()) for k, v in imgs.items()}
    assert abs(m["plain", "spcbpt"] - m["plain", "pt"]) < 0.05 * m["plain", "pt"]              # unflagged: the two estimators agree
    assert m["flagged", "spcbpt"] > 1.3 * m["flagged", "pt"], m                                 # flagged: SPCBPT integrates f / |cos|
