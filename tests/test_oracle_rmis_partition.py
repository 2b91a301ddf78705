"""Rows a10 / a15 / a16, oracle only (CPU): the restated recursive-MIS weights against FIRST PRINCIPLES.

The reference cannot run here and holds no fixtures, so the oracle's core stays unpinned against reference-compiled code (DESIGN.md
2).  What can be checked without the reference is whether the restated `rmis.h` is what it claims to be: a balance heuristic over
the strategies of a path.  For explicit camera paths c, x1 .. xD that end on an emitter, every strategy that can produce the path is
built by the generation code itself -- the eye sub-path as traced (RMIS_pointer_3 recursion of tracing_update_eye), the emitter
hit (light_hit), the light sub-path re-traced from the emitter point with its scattering directions forced onto the same vertices
(tracing_init_light / tracing_update_light) -- and the weight each renderer would apply (1 / RMIS_pointer of the hit,
connection_lightSource, general_connection) is compared with the weight computed from nothing but the vertices' pdfs:
rate(strategy) = eye pdf x connectRate_SOL(eye subspace, light subspace, flux / pdf) x light pdf, weight = rate / sum of rates.
A misread term anywhere in the recursion (a cosine, a pdf measure, a flux multiplier, the relabelling) breaks the equality; it holds
to float rounding, for every strategy, at every depth tried.  tests/test_oracle_env.py does the same for the sky's strategies."""
import numpy as np
import pytest

from tests.parity_util import minimal_tuple


@pytest.mark.parametrize("which", ["cornell", "room"])
def test_rmis_weights_equal_first_principles_and_sum_to_one(pkg, ob, which):
    scene = pkg.scenes.cornell_box() if which == "cornell" else pkg.scenes.bedroom(target_tris=3000, tex_size=16)
    W, H = 64, 64
    o = ob.Oracle(scene)
    cam = scene.camera
    o.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], W / H)
    o.resize(W, H)
    o.set_light_trace(3000, 64, 1)
    o.set_subspace(*minimal_tuple(o, 2))
    for depth in (1, 2, 3, 4):
        w, truth = o.quad_partition(depth, 600)
        assert len(w) >= 200, (which, depth, len(w))
        total = w[:, 0]
        ok = np.abs(total - 1) < 1e-3
        assert ok.mean() > 0.99, (which, depth, ok.mean(), np.percentile(total, [1, 50, 99]))
        d = np.abs(w[ok, 1:] - truth[ok])
        assert d.max() < 2e-3 and d.mean() < 2e-5, (which, depth, d.max(), d.mean())
        assert (w[:, 1] > 0).all()                                   # the emitter hit always has a share
        if depth >= 2: assert w[:, 3].mean() > 0.05                  # ... and so have the connections to light vertices of depth >= 1


def test_partition_holds_with_trained_classifier_trees(pkg, ob):
    """The same with a TRAINED tuple (multi-leaf trees that split on positions and normals, a non-uniform Gamma): now the relabelling of
    rmis.h:58-79 / 131-151 matters -- the label a strategy's weight predicts for a vertex must be the label the vertex gets when the
    other strategy really creates it.  (Paths that scattered INTO a surface on the way are left out: upstream lets them live with a
    flux of exactly zero, DESIGN d11; they carry nothing and no light sub-path can retrace them.)"""
    scene = pkg.scenes.cornell_box()
    W, H = 64, 64
    o = ob.Oracle(scene)
    cam = scene.camera
    o.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], W / H)
    o.resize(W, H)
    o.set_light_trace(3000, 64, 1)
    o.preprocess(40000, 40000, True)
    for depth in (2, 3, 4):
        w, truth = o.quad_partition(depth, 600)
        assert len(w) >= 200
        d = np.abs(w[:, 1:] - truth).max(1)
        assert (d < 1e-3).mean() > 0.995, (depth, (d < 1e-3).mean())
        assert np.median(np.abs(w[:, 0] - 1)) < 1e-5
