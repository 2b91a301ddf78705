"""Tripwire for the traversal schedule: round 4 rewrote the pooled pass three times (quad tail, fan-out tail, one gather per iteration a
step ahead) under the rule that no film may change by a bit -- verified across builds with tools/film_dump.py / film_cmp.py.  This test
keeps the rule: the accumulation buffers of three small scenes (SPCBPT with a tuple trained on the spot, and pt) must hash to what
the committed code rendered (tests/golden/film_hashes.json).  The films are deterministic across boxes; they do depend on the
compiler (the image's ROCm 7.2.0): after a toolchain change, or a change that is meant to alter them, regenerate the file as its
note says -- the oracle parity tests are what judges correctness then."""
import importlib.util
import json
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_films_hash_to_the_committed_ones(gpu, pkg):
    spec = importlib.util.spec_from_file_location("film_dump", os.path.join(ROOT, "tools", "film_dump.py"))
    fd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fd)
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "film_hashes.json")))["films"]
    got = fd.hashes(fd.films(pkg))
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k] == want[k], (k, "film changed: see the docstring of this test")
