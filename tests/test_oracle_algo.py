"""Behavioural checks of the CPU restatement (host logic, no GPU): sampler tables, bisection quirks,
octree descent, unbiasedness of SPCBPT+RMIS against PT+NEE, thread-count invariance."""
import numpy as np
import pytest

from tests.parity_util import minimal_tuple, rmse


def _oracle(ob, pkg, scene, w, h, lt=(400, 64, 1), threads=0):
    o = ob.Oracle(scene, nthreads=threads)
    cam = scene.camera
    o.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], w / h)
    o.resize(w, h)
    o.set_light_trace(*lt)
    return o


def test_binary_sample_is_the_reference_lower_bound(ob):
    """cuProg.h:245-264: returns the first bin with u < cmf[bin]; zero-weight bins are never picked;
    size == 1 returns bin 0 with pmf = cmf[0]."""
    rng = np.random.default_rng(0)
    for size in (1, 2, 3, 7, 64, 1000):
        w = rng.random(size)
        w[rng.random(size) < 0.3] = 0.0
        if w.sum() == 0:
            w[0] = 1.0
        cmf = (np.cumsum(w) / w.sum()).astype(np.float32)
        cmf[-1] = 1.0
        for seed in rng.integers(0, 2**32, 50, dtype=np.uint64):
            idx, pmf, s_after = ob.binary_sample(cmf, int(seed))
            s = (1664525 * int(seed) + 1013904223) & 0xFFFFFFFF
            u = np.float32((s & 0xFFFFFF) / 16777216.0)
            assert s_after == s
            expect = int(np.searchsorted(cmf, u, side="right")) if size > 1 else 0
            expect = min(expect, size - 1)
            assert idx == expect
            ref_pmf = cmf[idx] if idx == 0 else np.float32(cmf[idx] - cmf[idx - 1])
            assert pmf == ref_pmf
            if size > 1:
                assert pmf > 0  # zero-mass bins never selected


def test_tree_index_descends_by_type(ob, pkg):
    t = np.zeros(9, dtype=pkg.TREE_NODE_DTYPE)
    t[0]["mid"] = (0, 0, 0); t[0]["type"] = 0; t[0]["child"] = np.arange(1, 9)
    for k in range(1, 9):
        t[k]["leaf"] = 1; t[k]["label"] = 10 + k
    # make octant 7 (x>0,y>0,z>0) a normal split feeding two leaves
    t = np.concatenate([t, np.zeros(2, dtype=pkg.TREE_NODE_DTYPE)])
    t[8]["leaf"] = 0; t[8]["type"] = 1; t[8]["mid"] = (0, 0, 0); t[8]["child"] = [9, 10, 9, 10, 9, 10, 9, 10]
    t[9]["leaf"] = 1; t[9]["label"] = 500
    t[10]["leaf"] = 1; t[10]["label"] = 501
    pnd = np.array([[-1, -1, -1, 0, 0, 1, 0, 0, 1], [1, -1, -1, 0, 0, 1, 0, 0, 1], [1, 1, 1, -1, 0, 0, 0, 0, 1],
                    [1, 1, 1, 1, 0, 0, 0, 0, 1], [0, 0, 0, 0, 0, 1, 0, 0, 1]], np.float32)
    lab = ob.tree_index(t, pnd)
    assert list(lab) == [11, 12, 500, 501, 11]  # `>` is strict: points on the split plane go to the low child


def test_lvc_process_tables(ob, pkg):
    scene = pkg.scenes.simple_room()
    o = _oracle(ob, pkg, scene, 16, 16, lt=(300, 16, 2))
    o.set_subspace(*minimal_tuple(o, 1))
    o.launch("light trace", 3)
    lvc = o.lvc_read()
    o.build_sampler()
    sub, cmfs, jump, vc, pc = o.sampler_read()
    assert vc == len(lvc) and pc == int((lvc["depth"] == 0).sum())
    assert sub["size"].sum() == vc
    assert np.array_equal(sub["jump_bias"], np.concatenate([[0], np.cumsum(sub["size"])[:-1]]))
    assert sorted(jump.tolist()) == list(range(vc))
    for s in np.nonzero(sub["size"])[0]:
        b, n = sub["jump_bias"][s], sub["size"][s]
        seg = cmfs[b:b + n]
        assert (lvc["subspace_id"][jump[b:b + n]] == s).all()
        assert (np.diff(jump[b:b + n]) > 0).all()          # slot order inside a subspace
        assert (np.diff(seg) >= 0).all() and seg[-1] == 1.0
    # emitter patches occupy the top ids (cuProg.h:586-589): depth-0 vertices only
    assert (lvc["subspace_id"][lvc["depth"] == 0] >= 1000 - 16).all()
    assert (lvc["subspace_id"][lvc["depth"] > 0] == 0).all()


def test_core_budget_and_path_ids(ob, pkg):
    """LightTraceParams geometry (raygen.cu:620-685): <= core_padding vertices per core, path ids continuous per core."""
    scene = pkg.scenes.simple_room()
    o = _oracle(ob, pkg, scene, 8, 8, lt=(50, 5, 4))
    o.set_subspace(*minimal_tuple(o, 1))
    o.launch("light trace", 9)
    lvc = o.lvc_read()
    core = lvc["path_id"] // 4
    counts = np.bincount(core, minlength=50)
    assert counts.max() <= 5 and counts.min() >= 1
    assert (np.diff(lvc["path_id"].astype(np.int64)) >= 0).all()


def test_thread_count_invariance(ob, pkg):
    scene = pkg.scenes.cornell_box()
    imgs = []
    for th in (1, 4):
        o = _oracle(ob, pkg, scene, 24, 24, threads=th)
        o.set_subspace(*minimal_tuple(o, 1))
        for f in range(2):
            o.render_frame("SPCBPT_eye", f)
        imgs.append(o.read_accum())
    np.testing.assert_array_equal(imgs[0], imgs[1])


def test_row_bands_partition_the_image(ob, pkg):
    scene = pkg.scenes.cornell_box()
    full = _oracle(ob, pkg, scene, 24, 40)
    full.launch("pt", 0)
    parts = _oracle(ob, pkg, scene, 24, 40)
    for r in range(3):
        parts.launch("pt", 0, rows=(8 * r, 40, 3))
    np.testing.assert_array_equal(full.read_accum(), parts.read_accum())


@pytest.mark.parametrize("scene_name", ["cornell_box", "simple_room"])
def test_spcbpt_agrees_with_pt(ob, pkg, scene_name):
    """Unbiasedness cross-check: SPCBPT + RMIS and PT + NEE converge to the same image."""
    scene = getattr(pkg.scenes, scene_name)()
    o = _oracle(ob, pkg, scene, 32, 32, lt=(1000, 64, 1))
    n = 96
    for f in range(n):
        o.launch("pt", f)
    pt = o.read_accum()[..., :3].copy()
    o.clear_accum()
    o.set_subspace(*minimal_tuple(o, 4))
    for f in range(n):
        o.render_frame("SPCBPT_eye", f)
    sp = o.read_accum()[..., :3]
    assert abs(sp.mean() - pt.mean()) / pt.mean() < 0.02
    assert rmse(sp, pt) < 0.25 * pt.mean() + 0.02
    # coarse structure: 4x4 block means agree within 8 %
    blk = lambda a: a.reshape(4, 8, 4, 8, 3).mean(axis=(1, 3))
    assert np.abs(blk(sp) - blk(pt)).max() / blk(pt).mean() < 0.2


def test_three_estimators_of_the_restatement_agree(ob, pkg):
    """Row f4 / config 5's comparator on the CPU side: the recursive-MIS sampler ("SPCBPT_eye"), the same sampler with classic
    full-path MIS ("SPCBPT_no_rmis": contriCompute / pdfCompute / MISWeight_SPCBPT, none of rmis.h) and "plain BDPT" (uniformSample
    over the cache) estimate the same image on a multi-leaf tuple -- three derivations that would have to share a misreading."""
    from tests.parity_util import grid_tree_tuple
    scene = pkg.scenes.cornell_box()
    o = _oracle(ob, pkg, scene, 40, 40, lt=(3000, 64, 1))
    o.set_subspace(*grid_tree_tuple(pkg, o, scene))
    means = {}
    for name, alg, uniform, n in (("rmis", "SPCBPT_eye", False, 48), ("full_path", "SPCBPT_no_rmis", False, 48), ("uniform", "SPCBPT_eye", True, 96)):
        o.set_uniform_lvc(uniform)
        o.clear_accum()
        for f in range(n):
            o.render_frame(alg, f, launch_frame=300 + f)
        a = o.read_accum()[..., :3]
        assert np.isfinite(a).all()
        means[name] = float(a.astype(np.float64).mean())
    o.set_uniform_lvc(False)
    assert abs(means["full_path"] - means["rmis"]) / means["rmis"] < 5e-3, means        # same samples, two weight derivations
    assert abs(means["uniform"] - means["rmis"]) / means["rmis"] < 3e-2, means          # another sampler: Monte-Carlo noise only
