"""Rows a15 / a16 on the DEVICE against first principles (not against the oracle's rmis code).

tests/test_oracle_rmis_partition.py builds, for explicit camera paths, every strategy that can produce the path and derives each
strategy's MIS weight from nothing but the vertices' pdfs (rate = eye pdf x connectRate_SOL x light pdf; weight = rate / sum).  Here
the oracle is used for exactly that -- to GENERATE the vertex pairs (eye vertex e_d, light vertex y_k) and the first-principles
weights -- and the product's `connect_vertices` (the per-function harness, through the C ABI) evaluates its recursive-MIS weight for
every pair: it must equal rate / sum.  That holds the device's general_connection / connection_lightSource /
connection_direction_lightSource to the balance heuristic itself, with a trained tuple (so the relabelling and the cached labels
matter) and with an environment map."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
W, H = 64, 64


def _weights(r, ev, lv):
    from tests.test_gpu_units import OP
    n = len(ev)
    words = np.zeros((n, 52), np.uint32)
    words[:, :25] = np.ascontiguousarray(ev).view(np.uint32).reshape(n, 25)
    words[:, 25:49] = np.ascontiguousarray(lv).view(np.uint32).reshape(n, 24)
    return r.unit(OP["CONNECT"], words, 4).view(np.float32)[:, 3]


def _check(r, o, partition, depths, what):
    for depth in depths:
        w, truth, ev, lv = partition(depth, 600, vertices=True)
        assert len(w) >= 150, (what, depth, len(w))
        ok = np.abs(w[:, 0] - 1) < 1e-3                         # (paths on which the oracle's own weights are a partition: the check of the check)
        assert ok.mean() > 0.99
        w, truth, ev, lv = w[ok], truth[ok], ev[ok], lv[ok]
        total = truth[:, 0].astype(np.float64)                   # the emitter hit / the sky miss: first principles (no device function returns that weight by itself)
        for k in range(min(depth, 4)):
            got = _weights(r, ev[:, k], lv[:, k])
            d = np.abs(got - truth[:, 1 + k])
            print(what, depth, k, np.percentile(d, [50, 99.5, 100]))
            assert np.percentile(d, 99.5) < 2e-4 and d.max() < 5e-3, (what, depth, k, np.percentile(d, [50, 99.5, 100]))
            total += got
        assert np.percentile(np.abs(total - 1), 99.5) < 5e-4, (what, depth)


def test_device_connection_weights_are_the_balance_heuristic(gpu, pkg, ob):
    scene = pkg.scenes.cornell_box()
    cam = scene.camera
    r = pkg.Renderer(scene, 0)
    o = ob.Oracle(scene)
    for x in (r, o):
        x.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], W / H)
        x.resize(W, H)
        x.set_light_trace(3000, 64, 1)
    r.set_pretrace(20000, 10)
    r.preprocess(target_paths=100000, target_q_paths=100000, train=True)      # the product's own trained tuple: multi-leaf trees, Gamma != Q
    o.set_subspace(*r.get_subspace())
    r.launch("light trace", 1); r.build_sampler()
    _check(r, o, o.quad_partition, (1, 2, 3, 4), "cornell, trained tuple")


def test_device_weights_of_the_sky_strategies(gpu, pkg, ob):
    scene = pkg.scenes.courtyard()
    env = scene.environment
    r = pkg.Renderer(scene, 0)
    o = ob.Oracle(scene)
    for x in (r, o):
        x.set_camera_lookat((0.0, 2.6, 2.6), (0.0, 0.2, 0.0), (0, 1, 0), 40.0, W / H)
        x.resize(W, H)
        x.set_environment(env["rgba"], env["center"], env["radius"])
        x.set_light_trace(3000, 64, 1)
    r.set_pretrace(20000, 10)
    r.preprocess(target_paths=100000, target_q_paths=100000, train=True)
    o.set_subspace(*r.get_subspace())
    r.launch("light trace", 1); r.build_sampler()
    _check(r, o, o.env_partition, (1, 2, 3), "courtyard with a sky, trained tuple")
