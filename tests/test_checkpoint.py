"""Row f4: checkpoint files of the subspace tuple in the reference's text formats (tree_eye.txt, tree_light.txt, Q.txt, E.txt;
readers: decisionTree/classTree_host.h:15-59, cuda_thrust/device_thrust.cu:3347-3404).  The context-free reader / writer and
Gamma2CMFGamma run on the CPU; the context-level save/load is a GPU test."""
import os

import numpy as np
import pytest

NS = 1000


def _random_tree(pkg, rng, n_internal, label_bias=0):
    """A well-formed octree in the flat layout of classTree: internal nodes first come first served, leaves labelled."""
    nodes = np.zeros(1 + 8 * n_internal, dtype=pkg.api.TREE_NODE_DTYPE)
    internal = [0]
    nxt, label = 1, label_bias
    queue = [0]
    made = 0
    while queue:
        i = queue.pop(0)
        if made < n_internal:
            made += 1
            nodes[i]["leaf"] = 0
            nodes[i]["type"] = int(rng.integers(0, 3))
            nodes[i]["mid"] = rng.standard_normal(3).astype(np.float32) * np.float32(3.7)
            nodes[i]["label"] = int(rng.integers(0, 50))   # internal nodes carry a label field too; it must survive
            for c in range(8):
                nodes[i]["child"][c] = nxt
                queue.append(nxt)
                nxt += 1
        else:
            nodes[i]["leaf"] = 1
            nodes[i]["label"] = label
            label += 1
    return nodes[:nxt]


def test_write_read_round_trip_is_bit_exact(hip_lib, pkg, tmp_path):
    rng = np.random.default_rng(5)
    et, lt = _random_tree(pkg, rng, 9), _random_tree(pkg, rng, 4)
    q = rng.random(NS).astype(np.float32) * np.float32(1e-3)
    q[17] = np.float32(1.17549435e-38)     # smallest normal, a denormal and a large value must survive the text form
    q[18] = np.float32(1e-42)
    g = (rng.random((NS, NS)) ** 8).astype(np.float32)
    g[3, 5] = np.float32(3.4e38)
    pkg.api.checkpoint_write(str(tmp_path), et, lt, q, g)
    assert sorted(os.listdir(tmp_path)) == ["E.txt", "Q.txt", "tree_eye.txt", "tree_light.txt"]
    et2, lt2, q2, g2 = pkg.api.checkpoint_read(str(tmp_path))
    assert et2.tobytes() == et.tobytes() and lt2.tobytes() == lt.tobytes()
    assert q2.tobytes() == q.tobytes() and g2.tobytes() == g.tobytes()


def test_reader_parses_the_reference_grammar(hip_lib, pkg, tmp_path):
    # tree_load extracts with operator>>: any whitespace separates fields, a leaf record is `1 label`, an internal one
    # `0 label type mx my mz c0..c7`; a leaf's other fields keep tree_node's defaults (0)
    (tmp_path / "tree_eye.txt").write_text("0 7 2 0.5 -0.25 1e-3\n1 2 3 4\n5 6 7 8\n" + "".join(f"1 {k}\t" for k in range(10, 18)) + "\n")
    (tmp_path / "tree_light.txt").write_text("1 0")
    (tmp_path / "Q.txt").write_text("\n".join(str(0.001 * (k % 7)) for k in range(NS)))
    (tmp_path / "E.txt").write_text(" ".join("0.5" if k % NS == 0 else "0" for k in range(NS * NS)))
    et, lt, q, g = pkg.api.checkpoint_read(str(tmp_path))
    assert len(et) == 9 and len(lt) == 1
    assert et[0]["leaf"] == 0 and et[0]["label"] == 7 and et[0]["type"] == 2
    assert np.array_equal(et[0]["mid"], np.array([0.5, -0.25, 1e-3], np.float32)) and list(et[0]["child"]) == [1, 2, 3, 4, 5, 6, 7, 8]
    assert all(et[1 + k]["leaf"] == 1 and et[1 + k]["label"] == 10 + k and not et[1 + k]["child"].any() for k in range(8))
    assert lt[0]["leaf"] == 1 and lt[0]["label"] == 0
    assert np.array_equal(q, np.array([np.float32(float(str(0.001 * (k % 7)))) for k in range(NS)]))
    assert (g[:, 0] == 0.5).all() and not g[:, 1:].any()


def test_emitter_columns_keep_the_current_gamma_like_load_gamma_file(hip_lib, pkg, tmp_path):
    rng = np.random.default_rng(6)
    leaf = pkg.single_leaf_tree()
    g_file = rng.random((NS, NS)).astype(np.float32)
    pkg.api.checkpoint_write(str(tmp_path), leaf, leaf, np.ones(NS, np.float32), g_file)
    cur = rng.random((NS, NS)).astype(np.float32)
    _, _, _, g = pkg.api.checkpoint_read(str(tmp_path), current_gamma=cur)
    assert np.array_equal(g[:, :800], g_file[:, :800]) and np.array_equal(g[:, 800:], cur[:, 800:])
    _, _, _, g = pkg.api.checkpoint_read(str(tmp_path))
    assert np.array_equal(g, g_file)


def test_gamma_to_cmf_is_gamma2cmfgamma(hip_lib, pkg):
    rng = np.random.default_rng(7)
    g = (rng.random((NS, NS)) ** 6).astype(np.float32)
    g /= g.sum(axis=1, keepdims=True).astype(np.float32)
    got = pkg.api.gamma_to_cmf(g)
    t = np.float32(0.2)
    mixed = ((g * (np.float32(1) - t)).astype(np.float64) + (1.0 / NS) * np.float64(t)).astype(np.float32)
    want = mixed.copy()
    for j in range(1, NS):                      # fp32 running sum, as the reference's host loop
        want[:, j] = want[:, j] + want[:, j - 1]
    want[:, -1] = 1
    assert got.tobytes() == want.tobytes()
    assert (np.diff(got, axis=1) >= 0).all()


def test_errors(hip_lib, pkg, tmp_path):
    with pytest.raises(pkg.SpcbptError, match="-7"):
        pkg.api.checkpoint_read(str(tmp_path / "nowhere"))
    leaf = pkg.single_leaf_tree()
    pkg.api.checkpoint_write(str(tmp_path), leaf, leaf, np.ones(NS, np.float32), np.zeros((NS, NS), np.float32))
    (tmp_path / "Q.txt").write_text("1 2 3")     # truncated
    with pytest.raises(pkg.SpcbptError, match="-7"):
        pkg.api.checkpoint_read(str(tmp_path))
    big = np.zeros(40, dtype=pkg.api.TREE_NODE_DTYPE); big["leaf"] = 1
    pkg.api.checkpoint_write(str(tmp_path), big, leaf, np.ones(NS, np.float32), np.zeros((NS, NS), np.float32))
    with pytest.raises(pkg.SpcbptError, match="-6"):
        pkg.api.checkpoint_read(str(tmp_path), cap=8)
    assert hip_lib.spcbpt_checkpoint_save(None, b".") == -1 and hip_lib.spcbpt_checkpoint_load(None, b".") == -1


@pytest.mark.gpu
def test_context_save_load_reinstalls_the_trained_tuple(gpu, pkg, tmp_path):
    scene = pkg.scenes.cornell_box()
    cam = scene.camera

    def make():
        r = pkg.Renderer(scene, 0)
        r.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], 1.0)
        r.resize(96, 96)
        r.set_light_trace(4000, 64, 1)
        return r

    a = make()
    with pytest.raises(pkg.SpcbptError):       # nothing preprocessed yet: no Gamma to write
        a.set_subspace()
        a.checkpoint_save(str(tmp_path))
    a.set_pretrace(8000, 10)
    a.preprocess(target_paths=30000, target_q_paths=30000, train=True)
    a.checkpoint_save(str(tmp_path))
    want = a.get_subspace()
    b = make()
    b.checkpoint_load(str(tmp_path))
    got = b.get_subspace()
    for x, y in zip(got, want):
        assert np.ascontiguousarray(x).tobytes() == np.ascontiguousarray(y).tobytes()
    assert b.get_gamma().tobytes() == a.get_gamma().tobytes()
    for r in (a, b):
        r.clear_accum()
        for f in range(2):
            r.render_frame("SPCBPT_eye", f)
        r.sync()
    assert np.array_equal(a.read_accum(), b.read_accum())
