"""The image-parity bar itself (tests/parity_util.py): an image pair may only pass when the pixels OUTSIDE the per-pixel tolerance
are other samples of the same distribution -- finite, of ordinary size, unbiased -- not an unbounded few per cent."""
import numpy as np

from tests.parity_util import image_parity, tails_explained


def _pair(seed=0, n=64):
    rng = np.random.default_rng(seed)
    b = rng.gamma(2.0, 0.3, (n, n, 3))
    return b.copy(), b


def test_identical_images_and_resampled_pixels_pass():
    a, b = _pair()
    s = image_parity(a, b)
    assert s["frac_close"] == 1.0 and s["n_out"] == 0 and tails_explained(s)
    rng = np.random.default_rng(1)
    idx = rng.random(a.shape[:2]) < 0.02            # 2 % of the pixels hold another sample of the same distribution
    a[idx] = rng.gamma(2.0, 0.3, (int(idx.sum()), 3))
    s = image_parity(a, b)
    assert 0.97 < s["frac_close"] < 0.99 and tails_explained(s), s


def test_garbage_in_the_tail_fails_although_frac_close_is_high():
    for kind in ("nan", "huge", "biased"):
        a, b = _pair(2)
        rng = np.random.default_rng(3)
        idx = rng.random(a.shape[:2]) < 0.02
        if kind == "nan": a[idx] = np.nan
        elif kind == "huge": a[idx] = 1e4             # orders of magnitude above the image
        else: a[idx] = b[idx] * 1.6                   # every outlier brighter: a systematic shift hiding in 2 % of the pixels
        s = image_parity(a, b)
        assert s["frac_close"] > 0.97 and not tails_explained(s), (kind, s)
