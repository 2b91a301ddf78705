"""Per-function parity of the device library against the oracle, through spcbpt_debug_unit (csrc/unit.hip): the functions the
megakernel inlines, evaluated one record per lane on identical inputs.  This is the bar for rows a7 (BSDF), a8 / a10 (eye
vertex, emitter hit), a12 (classification), a13 (both resampling stages), a15 / a16 (connection value and recursive-MIS
weight) -- the image tests only show that their composition agrees for ~99 % of the pixels.

Tolerances: integers (labels, bins, slots, seeds, flags) exact.  Floating point: FP32 on both sides, and since round 3 the device
code is compiled like the oracle, WITHOUT implicit multiply-add fusion (-ffp-contract=off; the explicit fmaf / packed FMAs of the
traversal loop and the device's own sincos / pow / log / rcp remain) -- the arithmetic is the oracle's operation for operation and
most records agree bit for bit:
  * BSDF Eval 1e-6, Pdf 3e-6, sampled direction 5e-6 for >= 99.9 % of the records, 1e-4 hard on EVERY record (measured: Eval
    bit-exact for 99 %, max 2.6e-7; Pdf max 2.0e-6; direction max 2.3e-6);
  * connection value and recursive-MIS weight 2e-6 for >= 99.9 %, 1e-4 hard (measured: bit-exact for 90 %, max 5.0e-7);
  * the eye step (which goes through the traversal: v_rcp_f32 and contracted cross products there, DESIGN d13) 2e-5 for >= 99.8 %;
  * values downstream of a texture fetch (`color` = pow(texel, 2.2), NextVertex.flux / singlePdf, RMIS_pointer_3 after four
    bounces): device powf is 1e-4 off libm in the worst case -> 2e-4 / 99.8 %;
  * a record within rounding of a branch (Russian roulette r ~ rr, the hemisphere test of Eval, a triangle edge, an octree
    split) takes the other branch on one side: at most 2e-3 of the records.
(Round 2, with hipcc's default fusion: 1e-5 ... 5e-5 for 99.8 %, hard limits 1e-3.)
Measured error quantiles are printed by every check (SPCBPT_UNIT_REPORT=1 prints all of them without stopping)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
OP = dict(BSDF=0, TREE=1, STAGE1=2, BSEARCH=3, STAGE2=4, UNIFORM=5, CONNECT=6, EYE_STEP=7)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def close(a, b, rel=1e-5, scale=None):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    s = np.abs(b) if scale is None else scale
    if a.ndim > 1 and scale is None:
        s = np.abs(b).max(axis=-1, keepdims=True)
    return np.abs(a - b) <= rel * s + 1e-30


def frac(mask):
    return float(np.asarray(mask).mean())


def relerr(a, b, scale=None):
    """|a - b| / scale per record (vector fields: max component over the record's own magnitude); NaN on both sides = equal"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    a, b = np.where(both_nan, 0.0, a), np.where(both_nan, 0.0, b)
    if scale is None:
        scale = np.abs(b).max(axis=-1, keepdims=True) if a.ndim > 1 else np.abs(b)
    e = np.abs(a - b) / (scale + 1e-30)
    e = np.where(np.isnan(e), np.inf, e)
    return e.max(axis=-1) if e.ndim > 1 else e


SOFT = bool(os.environ.get("SPCBPT_UNIT_REPORT"))   # developer: print every error distribution instead of stopping at the first bar missed


def check(name, a, b, rel, min_frac, scale=None, hard=None):
    """>= min_frac of the records within `rel`; every record within `hard` (if given).  The message carries the quantiles."""
    e = relerr(a, b, scale)
    q = np.quantile(e, [0.5, 0.9, 0.99, 0.999, 1.0]) if len(e) else np.zeros(5)
    msg = f"{name}: within {rel:g}: {frac(e <= rel):.5f} (need {min_frac}); error quantiles 50/90/99/99.9/100 % = " + " ".join(f"{x:.3g}" for x in q)
    print(msg)
    if SOFT:
        return
    assert frac(e <= rel) >= min_frac, msg
    if hard is not None:
        assert q[-1] <= hard, msg


def build_world(pkg, ob, scene):
    """A renderer / oracle pair on `scene` with a TRAINED tuple (multi-leaf trees, non-trivial Gamma / Q), the same tuple and the
    same light-vertex cache on both sides."""
    W, H = 128, 72
    r = pkg.Renderer(scene, 0)
    o = ob.Oracle(scene)
    cam = scene.camera
    for x in (r, o):
        x.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], W / H)
        x.resize(W, H)
        x.set_light_trace(8000, 64, 1)
    r.set_pretrace(20000, 10)
    r.preprocess(target_paths=100000, target_q_paths=100000, train=True)
    tup = r.get_subspace()
    o.set_subspace(*tup)
    o.set_cmf_double(True)
    o.launch("light trace", 11)
    lvc = o.lvc_read()
    o.build_sampler()
    r.lvc_import(lvc); r.build_sampler()
    return dict(scene=scene, r=r, o=o, tup=tup, lvc=lvc, W=W, H=H)


@pytest.fixture(scope="module")
def world(gpu, pkg, ob):
    """Bedroom-class scene (40 k triangles, textured materials)."""
    return build_world(pkg, ob, pkg.scenes.bedroom(target_tris=40000, tex_size=64))


# ---------------------------------------------------------------------------------------------------------------- a7
def _bsdf_records(n, rng):
    rec = np.zeros((n, 24), np.float32)
    rec[:, 0:3] = rng.uniform(0.02, 1.0, (n, 3))                    # base colour
    rec[:, 3] = rng.choice([0.0, 1.0, 0.5, 0.3], n)                 # metallic
    rec[:, 4] = rng.choice([0.05, 0.1, 0.3, 0.5, 0.8, 1.0, 0.0005], n)   # roughness (incl. below the 0.001 clamp)
    rec[:, 5], rec[:, 6], rec[:, 7] = 0.5, 0.0, 0.0                 # specular, specularTint, subsurface: MaterialData() defaults (q17)
    rec[:, 8], rec[:, 9], rec[:, 10], rec[:, 11] = 0.0, 0.5, 0.0, 1.0
    k = n // 4                                                       # a quarter with every Disney lobe switched on
    rec[:k, 5:12] = np.stack([rng.uniform(0, 1, k), rng.uniform(0, 1, k), rng.uniform(0, 1, k), rng.uniform(0, 1, k),
                              rng.uniform(0, 1, k), rng.uniform(0, 1, k), rng.uniform(0, 1, k)], 1)

    def unit(v):
        return v / np.linalg.norm(v, axis=1, keepdims=True)
    N = unit(rng.normal(size=(n, 3)))
    V = unit(N + 0.9 * unit(rng.normal(size=(n, 3))))               # mostly above the surface
    L = unit(N * rng.uniform(-0.2, 1.0, (n, 1)) + unit(rng.normal(size=(n, 3))))   # some below: Eval must return exactly 0
    rec[:, 12:15], rec[:, 15:18], rec[:, 18:21] = N, V, L
    words = rec.view(np.uint32).copy()
    words[:, 21] = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    return words


def test_bsdf_sample_eval_pdf(world, ob):
    r = world["r"]
    rng = np.random.default_rng(3)
    n = 20000
    words = _bsdf_records(n, rng)
    out = r.unit(OP["BSDF"], words, 12)
    rec = words.view(np.float32)
    fo, po, so, fso, pso = (np.zeros((n, 3), np.float32), np.zeros(n, np.float32), np.zeros((n, 3), np.float32),
                            np.zeros((n, 3), np.float32), np.zeros(n, np.float32))
    seeds_after = np.zeros(n, np.uint32)
    for i in range(n):                      # the oracle's entry points take one material per call
        m = dict(color=tuple(rec[i, 0:3]), metallic=float(rec[i, 3]), roughness=float(rec[i, 4]), specular=float(rec[i, 5]),
                 specular_tint=float(rec[i, 6]), subsurface=float(rec[i, 7]), sheen=float(rec[i, 8]), sheen_tint=float(rec[i, 9]),
                 clearcoat=float(rec[i, 10]), clearcoat_gloss=float(rec[i, 11]))
        ls, sa = ob.bsdf_sample(m, rec[i:i + 1, 12:18], words[i:i + 1, 21])
        so[i], seeds_after[i] = ls[0], sa[0]
        f, p = ob.bsdf_eval_pdf(m, rec[i:i + 1, 12:21])
        fo[i], po[i] = f[0], p[0]
        f, p = ob.bsdf_eval_pdf(m, np.concatenate([rec[i, 12:18], ls[0]])[None, :])
        fso[i], pso[i] = f[0], p[0]
    g = out.view(np.float32)
    assert np.array_equal(out[:, 3], seeds_after)                                   # three rnd() draws, integer LCG
    check("Sample direction", g[:, 0:3], so, 5e-6, 0.999, scale=1.0, hard=1e-4)      # unit vector; measured (round 3, device code without implicit FMA fusion): 99.9 % within 3.6e-7, max 2.3e-6
    rough = rec[:, 4] >= 0.05
    check("Sample direction, roughness >= 0.05", g[rough, 0:3], so[rough], 5e-6, 0.999, scale=1.0, hard=1e-4)
    check("Eval", g[:, 4:7], fo, 1e-6, 0.999, hard=1e-4)                             # measured: bit-exact for 99 %, max 2.6e-7 (round 2, with fusion: 99.88 % within 1e-5, max 3.2e-4)
    check("Pdf", g[:, 7], po, 3e-6, 0.999, hard=1e-4)                                 # measured: bit-exact for 90 %, 99.9 % within 6.3e-7, max 2.0e-6 (logf / GTR1 at alpha = 0.001)
    check("Eval, roughness >= 0.05", g[rough, 4:7], fo[rough], 1e-6, 0.999, hard=2e-5)    # measured max 1.2e-7
    check("Pdf, roughness >= 0.05", g[rough, 7], po[rough], 3e-6, 0.999, hard=2e-5)
    below = (rec[:, 12:15] * rec[:, 18:21]).sum(1) < -1e-6
    assert below.sum() > 100 and (g[below, 4:7] == 0).all() and (fo[below] == 0).all()     # Eval == 0 below the surface, exactly
    # Eval / Pdf at the device's own sampled direction vs the oracle's at ITS sampled direction (what a path actually multiplies
    # in): the GGX peak amplifies the 1e-7 difference of the two directions, without bound as alpha -> 0.001, so this pair is
    # held at roughness >= 0.05 and 1e-3 (the fixed-direction rows above carry the tight bar)
    check("Eval at own sample, roughness >= 0.05", g[rough, 8:11], fso[rough], 1e-3, 0.995)
    check("Pdf at own sample, roughness >= 0.05", g[rough, 11], pso[rough], 1e-3, 0.995)


def test_bsdf_known_answer_of_the_reference_on_the_device(world):
    """the one record extracted from the reference's own Tracer::Sample / Eval / Pdf, evaluated by the kernel code."""
    k = np.load(os.path.join(G, "survey_kat.npz"))
    rec = np.zeros((1, 24), np.float32)
    rec[0, 0:3] = k["bsdf_mat"][:3]; rec[0, 3] = k["bsdf_mat"][3]; rec[0, 4] = k["bsdf_mat"][4]
    rec[0, 5:12] = [0.5, 0.0, 0.0, 0.0, 0.5, 0.0, 1.0]
    V = k["bsdf_V"] / np.linalg.norm(k["bsdf_V"])
    rec[0, 12:15], rec[0, 15:18], rec[0, 18:21] = k["bsdf_N"], V, k["bsdf_L"]
    words = rec.view(np.uint32).copy()
    from oracle import binding
    words[0, 21] = binding.lib().orc_tea4(C.c_uint(int(k["bsdf_seed_args"][0])), C.c_uint(int(k["bsdf_seed_args"][1])))
    g = world["r"].unit(OP["BSDF"], words, 12).view(np.float32)[0]
    np.testing.assert_allclose(g[0:3], k["bsdf_L"], rtol=3e-6, atol=3e-7)            # Sample
    np.testing.assert_allclose(g[4:7], k["bsdf_f"], rtol=1e-5)                       # Eval at the reference's L
    np.testing.assert_allclose(g[7], k["bsdf_pdf"], rtol=1e-5)                       # Pdf at the reference's L
    np.testing.assert_allclose(g[8:11], k["bsdf_f"], rtol=2e-4)                      # ... and at the device's own sample
    np.testing.assert_allclose(g[11], k["bsdf_pdf"], rtol=2e-4)


# ---------------------------------------------------------------------------------------------------------------- a12
def test_tree_labels_exact(world, ob):
    r, scene, (et, lt, q, cmf) = world["r"], world["scene"], world["tup"]
    rng = np.random.default_rng(4)
    n = 50000
    lo, hi = scene.vertices.min(0), scene.vertices.max(0)
    pos = rng.uniform(lo, hi, (n, 3))
    nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    axis = rng.integers(0, 3, n); sign = rng.choice([-1.0, 1.0], n)
    k = n // 2                                          # half the normals axis-aligned: exact zeros sit ON the normal splits (mid = 0)
    nrm[:k] = 0.0; nrm[np.arange(k), axis[:k]] = sign[:k]
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    pnd = f32(np.concatenate([pos, nrm, d], 1))
    for which, tree in ((0, et), (1, lt)):
        rec = np.zeros((n, 10), np.uint32)
        rec[:, 0] = which
        rec[:, 1:] = pnd.view(np.uint32)
        got = r.unit(OP["TREE"], rec, 1)[:, 0].astype(np.int32)
        want = ob.tree_index(tree, pnd)
        assert np.array_equal(got, want), int((got != want).sum())
        assert len(set(want.tolist())) > 50            # a real multi-leaf tree


# ---------------------------------------------------------------------------------------------------------------- a13
def test_first_stage_counting_equals_the_reference_bisection(world, ob):
    r, (et, lt, q, cmf) = world["r"], world["tup"]
    rng = np.random.default_rng(5)
    n = 40000
    rec = np.zeros((n, 2), np.uint32)
    rec[:, 0] = rng.integers(0, 1000, n)
    rec[:, 1] = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    out = r.unit(OP["STAGE1"], rec, 6)
    pmf = out.view(np.float32)
    assert np.array_equal(out[:, 0], out[:, 3]) and np.array_equal(out[:, 1], out[:, 4]) and np.array_equal(out[:, 2], out[:, 5])
    for i in range(0, n, 7):                            # the oracle's binary_sample over the same row: bin, pmf bits, seed
        idx, p, s = ob.binary_sample(cmf[rec[i, 0]], int(rec[i, 1]))
        assert (idx, np.float32(p), s) == (int(out[i, 0]), pmf[i, 1], int(out[i, 2])), i
    assert len(set(out[:, 0].tolist())) > 100


def test_binary_sample_edge_cases(world, ob):
    """q9: the bespoke bisection on sizes 1, 2, 3, on CMFs with zero-mass bins (never picked) and on long tables."""
    r = world["r"]
    rng = np.random.default_rng(6)
    tables = [np.array([1.0]), np.array([0.37]), np.array([0.25, 1.0]), np.array([0.0, 1.0]), np.array([1.0, 1.0]), np.array([0.2, 0.2, 1.0]),
              np.array([0.0, 0.0, 0.5, 0.5, 0.5, 1.0]), np.cumsum(rng.uniform(0, 1, 1000)) / 500.0, np.cumsum(rng.uniform(0, 1, 4097))]
    tables[-1] = tables[-1] / tables[-1][-1]
    z = np.cumsum(np.where(rng.uniform(0, 1, 777) < 0.6, 0.0, rng.uniform(0, 1, 777)))      # 60 % zero-mass bins
    tables.append(z / z[-1])
    aux = f32(np.concatenate(tables))
    offs = np.cumsum([0] + [len(t) for t in tables[:-1]])
    recs, want = [], []
    for t, off in zip(tables, offs):
        for seed in rng.integers(0, 2**32, 300, dtype=np.uint64):
            recs.append((off, len(t), int(seed)))
            want.append(ob.binary_sample(aux[off:off + len(t)], int(seed)))
    rec = np.array(recs, dtype=np.uint32)
    out = r.unit(OP["BSEARCH"], rec, 3, aux=aux)
    pmf = out.view(np.float32)
    for i, (idx, p, s) in enumerate(want):
        assert (idx, np.float32(p), s) == (int(out[i, 0]), pmf[i, 1], int(out[i, 2])), (i, recs[i])
    zero_bins = [(i, int(out[i, 0])) for i, rc in enumerate(recs) if rc[1] == 777]
    zt = aux[offs[-1]:offs[-1] + 777]
    assert all(zt[b] > (zt[b - 1] if b else 0.0) for _, b in zero_bins)                     # a zero-mass bin is never returned


def test_second_stage_and_uniform_sample(world, ob):
    r, o = world["r"], world["o"]
    sub, cmfs, jump, vc, pc = o.sampler_read()
    subg, cmfg, jumpg, vcg, pcg = r.sampler_read()
    assert (vc, pc) == (vcg, pcg) and np.array_equal(jump, jumpg) and np.array_equal(sub["size"], subg["size"])
    rng = np.random.default_rng(7)
    n = 20000
    rec = np.zeros((n, 2), np.uint32)
    filled = np.nonzero(sub["size"] > 0)[0]
    rec[:, 0] = np.where(rng.uniform(0, 1, n) < 0.9, rng.choice(filled, n), rng.integers(0, 1000, n))
    rec[:, 1] = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    out = r.unit(OP["STAGE2"], rec, 5)
    pmf = out.view(np.float32)
    same = 0
    for i in range(n):
        l, seed = int(rec[i, 0]), int(rec[i, 1])
        size, b = int(sub["size"][l]), int(sub["jump_bias"][l])
        assert int(out[i, 0]) == size
        if size == 0:
            assert out[i, 1] == 0xFFFFFFFF and out[i, 4] == seed          # skipped before any random number is drawn
            same += 1
            continue
        k, p, s = ob.binary_sample(cmfg[b:b + size], seed)               # on the DEVICE's table: bit-exact bisection
        assert (k, np.float32(p), s, int(jumpg[b + k])) == (int(out[i, 1]), pmf[i, 3], int(out[i, 4]), int(out[i, 2])), i
        ko, _, _ = ob.binary_sample(cmfs[b:b + size], seed)              # on the oracle's table (CMFs agree to 2e-7): same bin but for ties
        same += ko == k
    assert same >= 0.999 * n
    seeds = rng.integers(0, 2**32, 4000, dtype=np.uint64).astype(np.uint32)
    out = r.unit(OP["UNIFORM"], seeds.reshape(-1, 1), 3)
    from oracle import binding
    for i, s0 in enumerate(seeds):                                        # uniformSample (cuProg.h:283-289)
        s = C.c_uint(int(s0))
        u = binding.lib().orc_rnd(C.byref(s))
        idx = min(int(np.float32(u) * np.float32(vc)), vc - 1)
        assert (int(out[i, 0]), out.view(np.float32)[i, 1], int(out[i, 2])) == (int(jump[idx]), np.float32(1.0 / vc), s.value)


# ------------------------------------------------------------------------------------------------------- a8 / a10 / a16
def _camera_records(pkg, ob, world, n, rng):
    scene, W, H = world["scene"], world["W"], world["H"]
    cam = scene.camera
    U, V, Wv = pkg.camera_frame(np.array(cam["eye"], np.float32), np.array(cam["lookat"], np.float32), np.array(cam["up"], np.float32),
                                np.float32(cam["fov"]), np.float32(W / H))
    d = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
    dirs = d[:, :1] * U[None, :] + d[:, 1:] * V[None, :] + Wv[None, :]
    dirs = f32(dirs / np.linalg.norm(dirs, axis=1, keepdims=True))
    rec = np.zeros(n, ob.EYE_STEP_IN_DTYPE)
    rec["last"]["position"] = cam["eye"]; rec["last"]["normal"] = dirs; rec["last"]["flux"] = 1.0
    rec["last"]["last_position"] = cam["eye"]; rec["last"]["pdf"] = 1.0; rec["last"]["single_pdf"] = 1.0
    rec["next_single_pdf"] = 1.0
    rec["seed"] = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    rec["dir"] = dirs
    return rec


def _compare_steps(ob, g, o, level, sharp_materials=()):
    """sharp_materials: material ids whose GGX lobe is so narrow (roughness below the 0.001 clamp) that Eval / Pdf at each side's OWN
    freshly sampled direction cannot be compared tightly; those records get the loose bar below."""
    g = g.view(ob.EYE_STEP_OUT_DTYPE).reshape(-1)
    n = len(o)
    same_kind = g["kind"] == o["kind"]
    assert frac(same_kind) >= 0.999, (level, frac(same_kind))
    surf = same_kind & (o["kind"] == 1) & close(g["t_hit"], o["t_hit"], 1e-5, np.maximum(1.0, o["t_hit"]))
    assert frac(surf[o["kind"] == 1]) >= 0.998
    a, b = g[surf], o[surf]
    for k in ("material_id", "depth", "last_zone_id"):
        assert np.array_equal(a["mid"][k], b["mid"][k]), (level, k)
    assert frac(a["mid"]["subspace_id"] == b["mid"]["subspace_id"]) >= 0.999          # a hit within rounding of an octree split
    assert np.array_equal(a["seed"], b["seed"])                                       # 3 (Sample) + 1 (RR) draws
    assert frac(a["done"] == b["done"]) >= 0.999
    scale = dict(position=1.0, normal=1.0, last_position=1.0)
    tex = dict(color=2e-4, rmis3=2e-4)        # downstream of pow(texel, 2.2): see the module docstring
    for k in ("position", "normal", "flux", "color", "last_position", "rmis3", "pdf", "single_pdf", "last_normal_projection"):
        check(f"level {level} mid.{k}", a["mid"][k], b["mid"][k], tex.get(k, 2e-5), 0.998, scale=scale.get(k), hard=2e-3)
    check(f"level {level} next direction", a["dir"], b["dir"], 2e-5, 0.998, scale=1.0, hard=1e-3)
    same_rr = a["done"] == b["done"]           # r within rounding of rr: one side multiplies NextVertex.singlePdf by rr, the other ends the path
    # Eval / Pdf at the freshly sampled direction (each side at its own): 5e-4 for >= 99.8 % (measured 99.9 % within 2.8e-4); no hard
    # limit -- on the 0.05-roughness metal the GGX peak turns the 1e-7 difference of the two directions into percents (test_bsdf_*)
    sharp = np.isin(b["mid"]["material_id"], list(sharp_materials))
    check(f"level {level} NextVertex.flux", a["next_flux"][~sharp], b["next_flux"][~sharp], 5e-4, 0.998)
    check(f"level {level} NextVertex.singlePdf", a["next_single_pdf"][same_rr & ~sharp], b["next_single_pdf"][same_rr & ~sharp], 5e-4, 0.998)
    if sharp.any():   # alpha = 0.001: the lobe's value moves by percents over the 1e-7 between the two sampled directions
        check(f"level {level} NextVertex.flux, alpha-clamped lobe", a["next_flux"][sharp], b["next_flux"][sharp], 5e-2, 0.98)
        check(f"level {level} NextVertex.singlePdf, alpha-clamped lobe", a["next_single_pdf"][same_rr & sharp], b["next_single_pdf"][same_rr & sharp], 5e-2, 0.98)
    emit = same_kind & (o["kind"] == 2)
    if emit.any():
        check(f"level {level} emitter radiance", g["emit"][emit], o["emit"][emit], 1e-4, 0.998, hard=1e-3)   # measured max 6.6e-5
    back = same_kind & (o["kind"] == 3)
    assert (g["emit"][back] == 0).all()
    return int(surf.sum()), int(emit.sum())


def test_eye_step_connection_and_emitter_hit_chain(world, pkg, ob):
    """Three levels of the eye walk, every level started from the ORACLE's previous output on both sides: vertex build with the RMIS
    recursion (a8), classification (a12), emitter hits with rmis::light_hit at depth >= 2 (a10) -- then connectVertex_SPCBPT and
    the RMIS connection weights (a15, a16) of those vertices against real light vertices, b.depth == 0 and > 0."""
    run_chain(world, pkg, ob)


def run_chain(world, pkg, ob, sharp_materials=(), seed=8):
    r, o, lvc = world["r"], world["o"], world["lvc"]
    rng = np.random.default_rng(seed)
    rec = _camera_records(pkg, ob, world, 16384, rng)
    eye_vertices, emit_total = [], 0
    for level in range(1, 5):
        want = o.eye_step(rec)
        got = r.unit(OP["EYE_STEP"], rec.view(np.uint32).reshape(len(rec), -1), 40)
        n_surf, n_emit = _compare_steps(ob, got, want, level, sharp_materials)
        emit_total += n_emit if level > 1 else 0
        go = (want["kind"] == 1) & (want["done"] == 0)
        eye_vertices.append(want["mid"][want["kind"] == 1].copy())
        nxt = np.zeros(int(go.sum()), ob.EYE_STEP_IN_DTYPE)
        nxt["last"] = want["mid"][go]; nxt["next_flux"] = want["next_flux"][go]; nxt["next_single_pdf"] = want["next_single_pdf"][go]
        nxt["seed"] = want["seed"][go]; nxt["dir"] = want["dir"][go]
        rec = nxt
        assert len(rec) > 500, level
    assert emit_total > 20            # emitter hits whose weight went through rmis::light_hit (depth >= 2)
    ev = np.concatenate(eye_vertices)
    n = len(ev)
    assert (ev["depth"] >= 3).sum() > 300
    origins = np.nonzero(lvc["depth"] == 0)[0]
    pick = np.where(rng.uniform(0, 1, n) < 0.3, rng.choice(origins, n), rng.integers(0, len(lvc), n))
    lv = lvc[pick]
    rgb_o, w_o = o.connect(ev, lv)
    words = np.zeros((n, 52), np.uint32)
    words[:, :25] = ev.view(np.uint32).reshape(n, 25)
    words[:, 25:49] = lv.view(np.uint32).reshape(n, 24)
    out = r.unit(OP["CONNECT"], words, 4).view(np.float32)
    live = (np.abs(rgb_o).max(1) > 0)
    assert live.sum() > 0.2 * n
    for sel, name in (((lv["depth"] == 0), "connection_lightSource"), ((lv["depth"] > 0), "general_connection")):
        assert sel.sum() > 1000, name
        check(name + " RMIS weight", out[sel, 3], w_o[sel], 2e-6, 0.999, hard=1e-4)     # measured: bit-exact for 90 %, max 5.0e-7 (round 2, with FMA fusion: 99.93 % within 5e-5)
        check(name + " value", out[sel, :3], rgb_o[sel], 2e-6, 0.999, hard=1e-4)
    assert ((out[:, :3] == 0).all(1) == (rgb_o == 0).all(1)).mean() >= 0.999          # the exact zeros (back-facing pairs, rejected values)
    return dict(eye=ev, light=lv, rgb=rgb_o, w=w_o)
