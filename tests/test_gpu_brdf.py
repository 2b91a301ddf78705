"""MaterialData::Pbr::brdf on the device (rows a8, a9, a15, a16): the five |n . out| divisions of the bidirectional programs
(hit_program.cu:286, 384; raygen.cu:271, 278; rmis.h:105 -- see tests/test_oracle_brdf.py for what they are and the first-principles
check of the restatement) in the HIP kernels against the oracle: function by function through spcbpt_debug_unit, the light-vertex
cache vertex by vertex, and images on the same seeds.  Bars are those of the un-flagged tests (tests/test_gpu_units.py,
tests/test_gpu_parity.py); the one exception is the `Glass` of the reference's shipped scene (roughness 0 -> alpha clamped to
0.001), whose lobe is too narrow to compare Eval at two independently sampled directions tightly."""
import numpy as np
import pytest

from tests.parity_util import cornell_with_flagged_box, image_parity, minimal_tuple, tails_explained
from tests.test_gpu_parity import _pair, check_lvc
from tests.test_gpu_units import build_world, run_chain

pytestmark = pytest.mark.gpu

GLASS = 11


@pytest.fixture(scope="module")
def world_flagged(gpu, pkg, ob):
    """The bedroom of test_gpu_units with `brdf 1` on the textured floor and wood, the walls, the 0.05-roughness metal and a plastic,
    and the dark glossy material replaced by the shipped scene's `Glass` block (colour 0.8, roughness 0, metallic 0, brdf 1)."""
    scene = pkg.scenes.bedroom(target_tris=40000, tex_size=64)
    for k in (0, 1, 3, 5, 7):
        scene.materials[k]["brdf"] = 1
    scene.materials[GLASS] = dict(color=(0.8, 0.8, 0.8), roughness=0.0, metallic=0.0, brdf=1)
    assert (scene.tri_material == GLASS).sum() > 100
    return build_world(pkg, ob, scene)


def test_eye_step_connection_and_emitter_hit_chain_with_flagged_materials(world_flagged, pkg, ob):
    """a8 / a10 / a15 / a16 with the divisions live: NextVertex.flux (hit_program.cu:286), fa / fb (raygen.cu:271, 278) and the flux
    multipliers of RMIS_pointer_3 / D_A (rmis.h:105).  The light-vertex cache of this world was traced by the oracle WITH the
    flag, so the light side's flux and RMIS_pointer carry hit_program.cu:384 as well."""
    res = run_chain(world_flagged, pkg, ob, sharp_materials=(GLASS,))
    scene = world_flagged["scene"]
    flagged = np.array([bool(m.get("brdf", 0)) for m in scene.materials] + [False] * 8)
    ev, lv = res["eye"], res["light"]
    live = np.abs(res["rgb"]).max(1) > 0
    ea, lb = flagged[ev["material_id"]], flagged[np.maximum(lv["material_id"], 0)] & (lv["depth"] > 0)
    # the compared connections really exercise every combination: flagged eye vertex, flagged light vertex, both, neither, and Glass
    for name, sel in (("eye", ea & ~lb), ("light", ~ea & lb), ("both", ea & lb), ("neither", ~ea & ~lb), ("glass eye vertex", ev["material_id"] == GLASS)):
        assert (sel & live).sum() > 50, (name, int((sel & live).sum()))


def test_flag_changes_the_device_result_exactly_where_the_lines_say(world_flagged, pkg, ob):
    """The same eye-step records on a renderer of the UN-flagged scene: everything but NextVertex.flux is bit-identical, and the
    flux differs by the factor 1 / |N . dir| on flagged materials only."""
    from tests.test_gpu_units import OP, _camera_records
    scene_plain = pkg.scenes.bedroom(target_tris=40000, tex_size=64)
    scene_plain.materials[GLASS] = dict(color=(0.8, 0.8, 0.8), roughness=0.0, metallic=0.0)
    rp = pkg.Renderer(scene_plain, 0)
    rf = world_flagged["r"]
    rp.set_subspace(*world_flagged["tup"])
    rec = _camera_records(pkg, ob, world_flagged, 8192, np.random.default_rng(21))
    words = rec.view(np.uint32).reshape(len(rec), -1)
    a = rp.unit(OP["EYE_STEP"], words, 40).view(ob.EYE_STEP_OUT_DTYPE).reshape(-1)
    b = rf.unit(OP["EYE_STEP"], words, 40).view(ob.EYE_STEP_OUT_DTYPE).reshape(-1)
    for k in a.dtype.names:
        if k != "next_flux":
            assert a[k].tobytes() == b[k].tobytes(), k
    surf = a["kind"] == 1
    flagged = np.array([bool(m.get("brdf", 0)) for m in world_flagged["scene"].materials])[a["mid"]["material_id"][surf]]
    assert flagged.sum() > 1000 and (~flagged).sum() > 300
    fa, fb = a["next_flux"][surf], b["next_flux"][surf]
    assert fa[~flagged].tobytes() == fb[~flagged].tobytes()
    cos = np.abs((a["mid"]["normal"][surf] * a["dir"][surf]).sum(1))
    live = flagged & (fa != 0).any(1) & np.isfinite(fb).all(1)
    np.testing.assert_allclose(fb[live] / np.where(fa[live] == 0, 1, fa[live]), np.where(fa[live] == 0, 1, 1.0 / cos[live][:, None]), rtol=3e-6)


@pytest.mark.parametrize("roughness", [0.3, 0.0])
def test_light_vertex_cache_with_a_flagged_box(gpu, pkg, ob, roughness):
    """a9: hit_program.cu:384 in k_light_trace -- flux, pdf and the RMIS_pointer recursion of every stored vertex."""
    scene = cornell_with_flagged_box(pkg, roughness=roughness, flag_walls=True)
    a, b = check_lvc(pkg, ob, scene, (3000, 64, 2))
    box = a["material_id"] == len(scene.materials) - 1
    deep = a["depth"] >= 2
    assert box.sum() > 100 and deep.sum() > 1000


@pytest.mark.parametrize("roughness,bar", [(0.3, 0.997), (0.0, 0.99)])
def test_spcbpt_image_with_a_flagged_box(gpu, pkg, ob, roughness, bar):
    """Image parity on the same seeds, Cornell box with a flagged short box (roughness 0: the shipped `Glass`) and flagged white walls."""
    scene = cornell_with_flagged_box(pkg, roughness=roughness, flag_walls=True)
    r, o = _pair(pkg, ob, scene, 96, 64)
    tup = minimal_tuple(o, 2)
    r.set_subspace(*tup); o.set_subspace(*tup)
    o.set_cmf_double(True)
    for f in range(4):
        r.render_frame("SPCBPT_eye", f); o.render_frame("SPCBPT_eye", f)
    a, b = r.read_accum()[..., :3], o.read_accum()[..., :3]
    s = image_parity(a, b)
    assert s["frac_close"] >= bar and s["mean_rel"] < 5e-3 and tails_explained(s), s
    # ... and it is not the un-flagged image: the flag is live on the device
    rp, _ = _pair(pkg, ob, pkg.scenes.cornell_box(), 96, 64)
    rp.set_subspace(*tup)
    for f in range(4):
        rp.render_frame("SPCBPT_eye", f)
    assert a.mean() > 1.15 * rp.read_accum()[..., :3].mean()


def test_pt_ignores_the_flag_on_the_device(gpu, pkg, ob):
    """hit_program.cu:439-552 has no division: the "pt" image of the flagged scene is the un-flagged one, bit for bit."""
    imgs = []
    for flag in (0, 1):
        scene = pkg.scenes.cornell_box()
        for m in scene.materials: m["brdf"] = flag
        r, _ = _pair(pkg, ob, scene, 64, 64)
        for f in range(2):
            r.launch("pt", f)
        imgs.append(r.read_accum().copy())
    assert imgs[0].tobytes() == imgs[1].tobytes()
