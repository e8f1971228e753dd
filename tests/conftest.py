import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU oracle case")


def _ensure_built():
    """Build the product libraries / CLI stages and the oracle when they are missing (fresh checkout:
    *.so are git-ignored).  hipcc cross-compiles for gfx950 without a GPU; on the GPU box the built
    files arrive with the snapshot and nothing happens here."""
    import subprocess
    pk = os.path.join(ROOT, "isee3-decoder_amd")
    need = [os.path.join(pk, "lib", "libviterbi224_hip.so"), os.path.join(pk, "lib", "libisee3dsp_hip.so"),
            os.path.join(pk, "bin", "vdecode"), os.path.join(pk, "bin", "isee3chain"),
            os.path.join(pk, "lib", "libisee3chain.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.run(["make", "-s", "-C", pk, "all"], check=True)
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)


def pytest_sessionstart(session):
    _ensure_built()


def load_pkg():
    """Import the hyphen-named package directory isee3-decoder_amd/ as module isee3_decoder_amd."""
    name = "isee3_decoder_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, "isee3-decoder_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def orc():
    import orc as _orc
    _orc.lib()
    return _orc
