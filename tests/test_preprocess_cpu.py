"""Preprocessing (pretrace -> trees -> Q -> Gamma_0 -> Adam -> CMF Gamma) of the CPU restatement: invariants of every
stage and the end-to-end unbiasedness of SPCBPT rendered with a trained tuple (no GPU)."""
import numpy as np
import pytest

from tests.parity_util import rmse


def _oracle(ob, pkg, scene, w, h, lt=(2000, 64, 1)):
    o = ob.Oracle(scene)
    cam = scene.camera
    o.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], w / h)
    o.resize(w, h)
    o.set_light_trace(*lt)
    return o


@pytest.fixture(scope="module")
def trained(ob, pkg):
    scene = pkg.scenes.cornell_box()
    o = _oracle(ob, pkg, scene, 64, 64)
    o.preprocess(target_paths=6000, target_q_paths=6000, train=True, num_core=4000, batch=2000)
    return scene, o


def test_pretrace_records_invariants(ob, pkg):
    scene = pkg.scenes.cornell_box()
    o = _oracle(ob, pkg, scene, 64, 64)
    n_valid = o.pretrace(1, 3000)
    paths, nodes = o.train_records()
    assert n_valid == len(paths) and 0.2 * 3000 < n_valid <= 3000
    assert (paths["valid"] == 1).all() and (nodes["valid"] == 1).all()
    assert paths["begin_ind"][0] == 0 and (paths["begin_ind"][1:] == paths["end_ind"][:-1]).all() and paths["end_ind"][-1] == len(nodes)
    k = paths["end_ind"] - paths["begin_ind"]
    assert k.min() >= 1 and k.max() <= 9           # eye surface vertices per path, padding 10 incl. the camera
    assert (nodes["path_id"] == np.repeat(np.arange(len(paths)), k)).all()
    # label_a carries the eye depth 1..k until the trees exist (set_eye_depth)
    assert (nodes["label_a"] == np.concatenate([np.arange(1, kk + 1) for kk in k])).all()
    # the last node of a path joins the last eye vertex with the emitter vertex; the others have surface B vertices
    last = paths["end_ind"] - 1
    assert (nodes["light_source"][last] == 1).all()
    inner = np.ones(len(nodes), bool); inner[last] = False
    assert (nodes["light_source"][inner] == 0).all()
    assert (nodes["label_b"][last] >= 900).all()   # emitter patch ids of the single divLevel-10 light
    assert np.isfinite(nodes["peak_pdf"]).all() and (nodes["peak_pdf"] >= 0).all()
    assert np.isfinite(paths["contri"]).all() and (paths["sample_pdf"] >= paths["fix_pdf"]).all()
    assert (paths["pixel_id"] >= 0).all() and (paths["pixel_id"] < 64).all()
    # a second launch appends with re-based indices (valid_sample_gather)
    o.pretrace(2, 3000)
    p2, n2 = o.train_records()
    assert (p2[:len(paths)]["begin_ind"] == paths["begin_ind"]).all() and p2["end_ind"][-1] == len(n2)
    assert (n2["path_id"][len(nodes):] >= len(paths)).all()


def test_tree_builder(ob, pkg):
    rng = np.random.default_rng(3)
    n = 6000
    pos = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    nor = rng.normal(size=(n, 3)).astype(np.float32); nor /= np.linalg.norm(nor, axis=1, keepdims=True)
    dire = rng.normal(size=(n, 3)).astype(np.float32); dire /= np.linalg.norm(dire, axis=1, keepdims=True)
    w = rng.uniform(0.5, 1.5, (n, 1)).astype(np.float32)
    samples = np.concatenate([pos, dire, nor, w], 1)
    K = 40
    t = ob.build_tree(samples, K, label_bias=7)
    leaves = t[t["leaf"] == 1]
    assert len(t) % 8 == 1 and t[0]["leaf"] == 0 and t[0]["type"] == 0
    assert leaves["label"].min() >= 7 and leaves["label"].max() < 7 + K + 1
    assert len(set(leaves["label"].tolist())) >= 0.8 * K
    inner = t[t["leaf"] == 0]
    assert (inner["child"] > 0).all() and (inner["child"] < len(t)).all()
    assert set(inner["type"].tolist()) <= {0, 1}          # DIR_JUDGE = 0: never a direction split
    np.testing.assert_allclose(t[0]["mid"], 0.5 * (pos.max(0) + pos.min(0)), rtol=1e-6)
    # normal-type nodes start at the origin of the unit cube of directions
    first_normal = t[(t["leaf"] == 0) & (t["type"] == 1)]
    assert np.abs(first_normal["mid"]).max() <= 1.0
    lab = ob.tree_index(t, np.concatenate([pos, nor, dire], 1))
    # labels the tree assigns reproduce the nearest-centroid partition for >= 90 % of the weight (threshold 0.99, depth 15)
    t2 = ob.build_tree(samples, K, label_bias=7)
    assert t2.tobytes() == t.tobytes()                    # deterministic
    counts = np.bincount(lab, minlength=7 + K + 1)
    assert counts[7:].sum() == n and (counts[7:] > 0).sum() >= 0.8 * K
    # degenerate input
    one = ob.build_tree(samples[:1], K, 3)
    assert len(one) == 1 and one[0]["leaf"] == 1 and one[0]["label"] == 3


def test_trained_tuple_is_valid(trained, pkg):
    scene, o = trained
    n = pkg.NUM_SUBSPACE
    q, cmf, gam = o.get_q(), o.get_cmf_gamma(), o.get_gamma()
    et, lt = o.get_tree(False), o.get_tree(True)
    assert (q > 0).all() and np.isfinite(q[q < 1e30]).all()
    assert (q[900:] < 1e30).all()                          # every emitter patch of the divLevel-10 light receives paths
    assert (np.diff(cmf, axis=1) > 0).all() and (cmf[:, -1] == 1.0).all() and (cmf[:, 0] > 0).all()
    np.testing.assert_allclose(gam.sum(axis=1), 1.0, rtol=2e-4)
    assert (gam >= 0).all()
    assert et[et["leaf"] == 1]["label"].max() < n and lt[lt["leaf"] == 1]["label"].max() < n - pkg.NUM_SUBSPACE_LIGHTSOURCE
    # the conservative mix guarantees every light subspace keeps >= 0.2/1000 of every row
    pmf = np.diff(np.concatenate([np.zeros((n, 1), np.float32), cmf], axis=1), axis=1)
    assert pmf.min() >= 0.9 * 0.2 / n
    # rows that saw training paths put most of their mass on light subspaces that carry light (Q finite)
    paths, nodes = o.train_records()
    rows = np.unique(nodes["label_a"])
    assert (nodes["label_a"] < n).all() and (nodes["label_b"] < n).all()
    lit = q < 1e30
    assert pmf[rows][:, lit].sum(axis=1).mean() > 0.9


def test_spcbpt_with_trained_tuple_is_unbiased(trained, ob, pkg):
    scene, o = trained
    n = 48
    for f in range(n):
        o.launch("pt", f)
    pt = o.read_accum()[..., :3].copy()
    o.clear_accum()
    for f in range(n):
        o.render_frame("SPCBPT_eye", f, launch_frame=500 + f)
    sp = o.read_accum()[..., :3]
    assert abs(sp.mean() - pt.mean()) / pt.mean() < 0.02
    blk = lambda a: a.reshape(4, 16, 4, 16, 3).mean(axis=(1, 3))
    assert np.abs(blk(sp) - blk(pt)).max() / blk(pt).mean() < 0.2


def test_gamma_training_reduces_the_loss(ob, pkg):
    """The objective of train_optimal_E: sum f^2 / (pdf0 + sum peak*E) must not increase over the Adam steps."""
    scene = pkg.scenes.simple_room()
    o = _oracle(ob, pkg, scene, 64, 64)
    o.train_records_clear()
    it = 0
    while o.train_records_count() < 8000:
        it += 1
        o.pretrace(it, 4000)
    o.preprocess_stage(1); o.preprocess_stage(2, 4000); o.preprocess_stage(3, 8000)
    g0 = o.get_gamma().copy()
    paths, nodes = o.train_records()
    q = o.get_q()

    def loss(gamma):
        E = gamma * 0.8 + 0.2 / 1000
        pk = np.where(q[nodes["label_b"]] > 0, nodes["peak_pdf"] / q[nodes["label_b"]], 0).astype(np.float64)
        pk[~np.isfinite(pk)] = 0
        per_node = pk * E[nodes["label_a"], nodes["label_b"]]
        pdf = paths["fix_pdf"].astype(np.float64) + np.bincount(nodes["path_id"], weights=per_node, minlength=len(paths))
        f2 = np.minimum(paths["contri"].sum(1).astype(np.float64) ** 2 / paths["sample_pdf"], 1e6)
        f2[~np.isfinite(f2)] = 1e6
        return float((f2[:8000] / pdf[:8000]).sum())

    l0 = loss(g0)
    o.preprocess_stage(4, 2000)
    l1 = loss(o.get_gamma())
    assert np.isfinite(l0) and np.isfinite(l1) and l1 <= l0 * 1.001, (l0, l1)
