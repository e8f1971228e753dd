"""CPU: the C-ABI libraries load and export every symbol include/*.h declares (no compute)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT, load_pkg


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b([a-z_0-9]+(?:_viterbi224(?:_blk)?|hip_[a-z0-9_]+))\s*\(", txt)))


def test_viterbi_headers_vs_library():
    pkg = load_pkg()
    so = pkg.lib_path("libviterbi224_hip.so")
    assert os.path.exists(so), "build() must produce %s" % so
    L = C.CDLL(so)
    names = _declared("viterbi224.h") + _declared("viterbi224_hip.h")
    assert len(names) >= 9 + 15
    for n in names:
        assert hasattr(L, n), "missing export %s" % n
    assert sorted(set(names)) == sorted(set(pkg.V224_SYMBOLS))


def test_reference_api_signature_names():
    """The nine entry points of reference viterbi224.h:8-16, verbatim names."""
    want = {"init_viterbi224", "create_viterbi224", "chainback_viterbi224", "delete_viterbi224",
            "update_viterbi224_blk", "max_metric_viterbi224", "min_metric_viterbi224",
            "decodebit_viterbi224", "decodeword_viterbi224"}
    assert want == set(_declared("viterbi224.h"))


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    pkg = load_pkg()
    with pytest.raises(RuntimeError):
        pkg.Viterbi224(16)
    L = pkg.v224_lib()
    assert L.init_viterbi224(None, 0) == -1            # NULL handle convention, port.c:38-39
    assert L.update_viterbi224_blk(None, None, 0) == -1
    assert L.decodebit_viterbi224(None, 1, 0) == -1


def test_dsp_header_vs_library():
    pkg = load_pkg()
    so = pkg.lib_path("libisee3dsp_hip.so")
    assert os.path.exists(so), "build() must produce %s" % so
    L = C.CDLL(so)
    txt = open(os.path.join(ROOT, "include", "isee3_dsp_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = sorted(set(re.findall(r"\b((?:symd|pmd|isync|isee3dsp)_[a-z0-9_]+)\s*\(", txt)))
    assert len(names) >= 14
    for n in names:
        assert hasattr(L, n), "missing export %s" % n
    assert names == sorted(pkg.DSP_SYMBOLS)


def test_cli_binaries_built():
    pkg = load_pkg()
    for exe in ("vdecode", "symdemod", "pmdemod", "decode"):
        assert os.access(pkg.cli_path(exe), os.X_OK)


def test_chain_header_vs_library():
    pkg = load_pkg()
    so = pkg.lib_path("libisee3chain.so")
    assert os.path.exists(so)
    L = C.CDLL(so)
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "isee3_chain.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(isee3_chain_[a-z0-9_]+)\s*\(", txt)))
    assert names == sorted(pkg.CHAIN_SYMBOLS)
    for n in names:
        assert hasattr(L, n)


def test_icesync_header_vs_library():
    pkg = load_pkg()
    L = C.CDLL(pkg.lib_path("libisee3chain.so"))
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "isee3_icesync.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(icesync_[a-z0-9_]+)\s*\(", txt)))
    assert names == sorted(pkg.ICESYNC_SYMBOLS) and len(names) == 6
    for n in names:
        assert hasattr(L, n)
