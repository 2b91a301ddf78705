import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _make(path, target=None):
    cmd = ["make", "-C", path] + ([target] if target else [])
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


@pytest.fixture(scope="session")
def ob(pkg):
    """oracle binding (test infrastructure)"""
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        _make(os.path.join(ROOT, "oracle"))
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def hip_lib(pkg):
    """the product shared library (cross-compiled by hipcc when missing; needs no GPU to load)"""
    csrc = os.path.join(ROOT, "spcbpt-optix7_amd", "csrc")
    if not os.path.exists(pkg.api.LIB_PATH) or not os.path.exists(pkg.dist.MGPU_LIB_PATH):
        _make(csrc)
    try:
        lib = pkg.load_library()       # refuses a library whose embedded source hash differs from the tree's (api.source_hash)
    except pkg.SpcbptError as e:
        if "built from other sources" not in str(e):
            raise
        _make(csrc)                    # stale binary: rebuild (hipcc cross-compiles here and compiles on the GPU box), then load
        lib = pkg.load_library()
    want = os.environ.get("SPCBPT_EXPECT_ARITHMETIC")   # the child run of tests/test_gpu_fast_build.py: prove which build is under test
    if want:
        assert lib.spcbpt_build_arithmetic().decode() == want, (lib.spcbpt_build_arithmetic(), want)
    return lib


def gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu(pkg, hip_lib):
    if not gpu_available():
        pytest.fail("GPU test selected but no HIP device is visible — the product has no CPU fallback")
    return 0
