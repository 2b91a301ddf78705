"""C-ABI checks that need no GPU: the library loads, exports every symbol include/spcbpt.h declares,
and fails loudly (no CPU fallback) when no HIP device exists."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "spcbpt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(spcbpt_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(hip_lib, pkg):
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(hip_lib, n), f"{n} declared in include/spcbpt.h but not exported"
    assert sorted(pkg.api.EXPORTED_SYMBOLS) == names


def test_struct_sizes_match_header(hip_lib, pkg, ob):
    """The ctypes / numpy mirrors against sizeof as the library was COMPILED from include/spcbpt.h (spcbpt_abi_struct_sizes)."""
    sizes = (C.c_int32 * 32)()
    n = hip_lib.spcbpt_abi_struct_sizes(sizes, 32)
    assert n == 13
    mat, tex, quad, desc, node, ltp, lv, sub, cnt, uev, ppath, pnode, vstate = list(sizes)[:n]
    assert C.sizeof(pkg.api.Material) == mat == 56         # 13 floats / ints + brdf
    assert C.sizeof(pkg.api.Texture) == tex and C.sizeof(pkg.api.QuadLight) == quad == 52 and C.sizeof(pkg.api.SceneDesc) == desc
    assert C.sizeof(pkg.api.TreeNode) == pkg.TREE_NODE_DTYPE.itemsize == node == 56      # sizeof(classTree::tree_node) in the reference is 56 too
    assert C.sizeof(pkg.api.LightTraceParams) == ltp
    assert pkg.LIGHT_VERTEX_DTYPE.itemsize == lv == 96
    assert pkg.SUBSPACE_DTYPE.itemsize == sub == 20
    assert C.sizeof(pkg.api.Counters) == cnt == 14 * 8
    assert ob.EYE_VERTEX_DTYPE.itemsize == uev
    assert pkg.api.PRETRACE_PATH_DTYPE.itemsize == ppath == 48 and pkg.api.PRETRACE_NODE_DTYPE.itemsize == pnode == 96
    assert C.sizeof(pkg.api.ViewerState) == vstate


def test_create_fails_loudly_without_a_gpu(hip_lib, pkg):
    from tests.conftest import gpu_available
    if gpu_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.SpcbptError) as e:
        pkg.Renderer(pkg.scenes.cornell_box(), 0)
    assert "no HIP device" in str(e.value) or "-2" in str(e.value)


def test_create_rejects_bad_scenes_before_touching_the_gpu(hip_lib, pkg):
    sc = pkg.scenes.cornell_box()
    sd, keep = sc.desc()
    bad = np.array(sc.indices, dtype=np.uint32).copy()
    bad[0, 0] = 10**6
    sd.indices = bad.ctypes.data
    h = C.c_void_p()
    rc = hip_lib.spcbpt_create(C.byref(sd), 0, C.byref(h))
    assert rc in (-1, -2)  # invalid argument, or no device when that check comes first
    assert hip_lib.spcbpt_create(None, 0, C.byref(h)) == -1


def test_null_context_is_an_error_not_a_crash(hip_lib):
    assert hip_lib.spcbpt_sync(None) == -1
    assert hip_lib.spcbpt_launch(None, b"pt", 0, 0, 8, 1) == -1
    assert hip_lib.spcbpt_build_sampler(None) == -1
