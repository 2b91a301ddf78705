"""Helpers shared by the parity tests, smoke() and bench.py (test infrastructure)."""
from __future__ import annotations

import os

import numpy as np

FLT_MAX = np.float32(3.4028234663852886e38)


def q_from_lvc(lvc: np.ndarray, n_sub: int = 1000):
    """Q[s] = E[sum_{v in s} flux/pdf per light path] (preprocess_getQ, device_thrust.cu:347-409) for one pass."""
    w = lvc["flux"].sum(axis=1, dtype=np.float32) / lvc["pdf"]
    w = np.where(np.isfinite(w), w, 0).astype(np.float32)
    q = np.bincount(lvc["subspace_id"].astype(np.int64), weights=w.astype(np.float64), minlength=n_sub)
    paths = int((lvc["depth"] == 0).sum())
    return (q / max(paths, 1)), paths


def minimal_tuple(oracle, frames: int = 4, first_frame: int = 10_000):
    """The cheapest VALID subspace tuple: single-leaf trees, Q from a few
    light passes, every Gamma row proportional to Q.  Returns (eye_tree, light_tree, q, cmf_gamma)."""
    pkg = oracle.pkg
    n = pkg.NUM_SUBSPACE
    acc = np.zeros(n, dtype=np.float64)
    tot = 0
    for f in range(frames):
        oracle.launch("light trace", first_frame + f)
        q, paths = q_from_lvc(oracle.lvc_read(), n)
        acc += q * paths
        tot += paths
    return tuple_from_q(pkg, acc / max(tot, 1))


def tuple_from_q(pkg, q64):
    n = pkg.NUM_SUBSPACE
    q = q64.astype(np.float32)
    row = q64 / q64.sum()
    cmf = np.cumsum(row).astype(np.float32)
    cmf[-1] = 1.0
    # monotone, and zero-mass bins keep pmf exactly 0
    cmf = np.maximum.accumulate(cmf)
    q = np.where(q == 0, FLT_MAX, q).astype(np.float32)  # Q_zero_handle device_thrust.cu:335-346
    gamma = np.tile(cmf[None, :], (n, 1)).astype(np.float32)
    return pkg.single_leaf_tree(0), pkg.single_leaf_tree(0), q, gamma


def image_parity(a: np.ndarray, b: np.ndarray, rel: float = 2e-3, abs_: float = 1e-4):
    """Per-pixel comparison of two linear accum images rendered with the same seeds.

    Besides the share of pixels within tolerance (`frac_close`) it describes the pixels OUTSIDE it, so that a test cannot pass
    with "3 % of the pixels differ by anything".  Two images of the same estimator at the same seeds differ beyond rounding only
    where a path took another turn -- a hit within rounding of a triangle edge, a Russian-roulette or resampling decision within
    rounding of its threshold -- and such a pixel then holds another SAMPLE of the same distribution: finite, of ordinary
    magnitude, and unbiased.  So for the outliers: `finite` (no NaN / Inf on either side), `out_abs_share` = sum |a - b| over them
    / sum |b| over the image (ordinary magnitude: about the share of outliers itself, not orders above it) and `out_signed_share`
    = sum (a - b) over them / sum b (no systematic shift hiding in the tail)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    finite = bool(np.isfinite(a).all() and np.isfinite(b).all())
    d = np.abs(a - b)
    close = (d <= abs_ + rel * np.abs(b)).all(axis=-1)
    l2 = np.sqrt((d ** 2).sum(axis=-1))
    out = ~close
    tot = max(float(np.abs(b).sum()), 1e-30)
    if os.environ.get("SPCBPT_PARITY_REPORT"):   # developer: what the image tests actually measure (run pytest with -s)
        import inspect
        fr = inspect.stack()[1]
        print(f"image_parity {os.path.basename(fr.filename)}:{fr.lineno} {fr.function}: frac_close {close.mean():.5f} mean_rel {abs(a.mean() - b.mean()) / max(b.mean(), 1e-12):.2e} "
              f"p99_l2 {np.percentile(l2, 99):.2e} pixels {close.size}")
    return dict(frac_close=float(close.mean()), mean_a=float(a.mean()), mean_b=float(b.mean()),
                mean_rel=float(abs(a.mean() - b.mean()) / max(b.mean(), 1e-12)),
                rmse=float(np.sqrt((d ** 2).mean())), max_l2=float(l2.max()), p99_l2=float(np.percentile(l2, 99)),
                finite=finite, n_out=int(out.sum()), out_abs_share=float(d[out].sum() / tot),
                out_signed_share=float((a - b)[out].sum() / tot))


def tails_explained(s: dict, magnitude: float = 4.0, shift: float = 2e-3) -> bool:
    """The outliers of image_parity are other samples of the same distribution, not garbage: everything finite; their summed
    absolute difference at most `magnitude` x their share of the pixels (two independent samples of a pixel differ by about the
    pixel's own size; heavy-tailed scenes get a larger factor from their test); their summed SIGNED difference below `shift` of the
    image's energy."""
    share = 1.0 - s["frac_close"]
    return bool(s["finite"] and s["out_abs_share"] <= magnitude * share + 1e-6 and abs(s["out_signed_share"]) <= shift)


def rmse(a, b):
    return float(np.sqrt(((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2).mean()))


def grid_tree_tuple(pkg, oracle, scene, frames: int = 2, first_frame: int = 20_000):
    """A valid multi-leaf tuple for tests: both classifiers are one position octree level (8 leaves) followed by a
    normal split under octant 0; Q measured with those trees; Gamma rows proportional to Q."""
    lo, hi = scene.vertices.min(0), scene.vertices.max(0)
    mid = 0.5 * (lo + hi)

    def tree(base):
        t = np.zeros(11, dtype=pkg.TREE_NODE_DTYPE)
        t[0]["mid"] = mid; t[0]["type"] = 0; t[0]["child"] = np.arange(1, 9)
        for k in range(1, 9):
            t[k]["leaf"] = 1; t[k]["label"] = base + k
        t[1]["leaf"] = 0; t[1]["type"] = 1; t[1]["mid"] = (0, 0, 0); t[1]["child"] = [9, 10, 9, 10, 10, 9, 10, 9]
        t[9]["leaf"] = 1; t[9]["label"] = base + 20
        t[10]["leaf"] = 1; t[10]["label"] = base + 21
        return t

    et, lt = tree(100), tree(300)
    n = pkg.NUM_SUBSPACE
    uni = np.tile((np.arange(1, n + 1, dtype=np.float32) / n)[None, :], (n, 1))
    oracle.set_subspace(et, lt, np.ones(n, np.float32), uni)
    acc = np.zeros(n, np.float64)
    tot = 0
    for f in range(frames):
        oracle.launch("light trace", first_frame + f)
        q, paths = q_from_lvc(oracle.lvc_read(), n)
        acc += q * paths
        tot += paths
    _, _, q, gamma = tuple_from_q(pkg, acc / max(tot, 1))
    return et, lt, q, gamma


def cornell_with_flagged_box(pkg, roughness: float = 0.0, color=(0.8, 0.8, 0.8), flag_walls: bool = False):
    """Cornell box whose SHORT box carries a material with `brdf 1` (MaterialData::Pbr::brdf) -- with roughness 0 and colour 0.8 it
    is the `Glass` block the reference ships (house_uvrefine2.scene:129-136).  flag_walls flags the white wall material too."""
    scene = pkg.scenes.cornell_box()
    tri_y = scene.vertices[scene.indices][:, :, 1]
    box = (tri_y.max(1) <= 0.6 + 1e-6) & (tri_y.max(1) > 0.0)       # sides and top of the short box; the floor lies at y = 0
    assert box.sum() == 10
    scene.materials.append(dict(color=tuple(color), roughness=roughness, metallic=0.0, brdf=1))
    scene.tri_material = scene.tri_material.copy()
    scene.tri_material[box] = len(scene.materials) - 1
    if flag_walls:
        scene.materials[0]["brdf"] = 1
    return scene
