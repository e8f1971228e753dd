"""CPU: the product's vdecode host logic (pairing, phase auto-flip, start-up suppression; C code in
isee3-decoder_amd/cli/vdecode_core.c) reproduces the reference vdecode's stdout byte for byte when
driven by the oracle engine; and the oracle's own vdecode restatement does too."""
import os
import subprocess

import numpy as np
import pytest

import orc
from conftest import ROOT

G = os.path.join(orc.GOLDEN, "vdecode_cli.npz")
BUILD = os.path.join(ROOT, "tests", "_build")


def _names():
    return [str(n) for n in np.load(G)["names"]]


def _args(z, name):
    return [a for a in z[name + "/args"] if a]


@pytest.fixture(scope="module")
def harness():
    orc.lib()
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "vdecode_oracle_test")
    src = [os.path.join(ROOT, "tests", "csrc", "vdecode_oracle_engine.c"),
           os.path.join(ROOT, "isee3-decoder_amd", "cli", "vdecode_core.c")]
    subprocess.run(["gcc", "-O2", "-o", exe] + src + ["-L" + orc.ORACLE_DIR, "-loracle",
                   "-Wl,-rpath," + orc.ORACLE_DIR, "-fopenmp", "-lm"], check=True)
    return exe


@pytest.mark.parametrize("whole", ["0", "1", "2", "limit"], ids=["blockwise", "whole_input", "progressive", "blockwise_read_limit_777"])
@pytest.mark.parametrize("name", _names())
def test_host_logic_with_oracle_engine(harness, name, whole):
    """block by block as the input arrives (the reference's way), or pass 1 over ALL input first and one engine call
    (whole-input mode, what `vdecode < file` uses), or the engine fed with the paired symbols while they are read and
    asked for everything at the end (progressive, what the chain uses): the same stdout"""
    z = np.load(G)
    p = subprocess.run([harness, "-q"] + _args(z, name), input=z[name + "/syms"].tobytes(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True, timeout=600,
                       env=dict(os.environ, VDECODE_WHOLE=whole if whole in ("1", "2") else "0",
                                **({"VDECODE_TEST_LIMIT": "777"} if whole == "limit" else {})))
    assert p.stdout == z[name + "/stdout"].tobytes()
    assert (b"PROGRESSIVE feeds=" in p.stderr) == (whole == "2")
    if name == "flip":
        assert b"flips=1" in p.stderr


@pytest.mark.parametrize("name", ["startphase_p", "delay_too_small"])
def test_oracle_vdecode_restatement(name):
    z = np.load(G)
    args = _args(z, name)
    delay = int(args[args.index("-d") + 1]) if "-d" in args else 200
    out, st = orc.vdecode(z[name + "/syms"], delay, int("-p" in args), "-F" in args)
    assert out == z[name + "/stdout"].tobytes()


E = os.path.join(orc.GOLDEN, "vdecode_stderr.npz")


def status_lines(stderr_bytes):
    """stderr of a vdecode run without the program name (argv[0] differs) and without this harness's own RESULT line"""
    out = []
    for ln in stderr_bytes.decode().splitlines():
        if ln.startswith("RESULT ") or ln.startswith("PROGRESSIVE ") or not ln.strip():
            continue
        out.append(ln.split(": ", 1)[1] if ": " in ln else ln)
    return out


@pytest.mark.parametrize("whole", ["0", "1", "2"], ids=["blockwise", "whole_input", "progressive"])
@pytest.mark.parametrize("name", [str(n) for n in np.load(E)["names"]])
def test_reencode_symbol_error_statistic_matches_reference_stderr(harness, name, whole):
    """vdecode.c:159-184: the decoded bits are re-encoded and compared with the hard-sliced received symbols; the tally
    goes to stderr every -i bits.  Fixture = stderr of the reference's own vdecode (port decoder) with -i 256, in the C
    locale: every status line, the `flipping phase` notice and the delay warning, in the same order."""
    z = np.load(E)
    args = [str(a) for a in z[name + "/args"]]
    p = subprocess.run([harness] + args, input=np.load(G)[name + "/syms"].tobytes(), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, check=True, timeout=600,
                       env=dict(os.environ, VDECODE_WHOLE=whole, LANG="C", LC_ALL="C"))
    want = status_lines(z[name + "/stderr"].tobytes())
    assert any("symerrs" in ln for ln in want)
    assert status_lines(p.stderr) == want


@pytest.mark.parametrize("whole", ["0", "1", "2"], ids=["blockwise", "whole_input", "progressive"])
def test_engine_fault_and_short_output_are_errors_not_truncation(harness, whole):
    """pass 2's guards (vdecode_core.c): an engine that hands back something that is not a bit after start-up is an
    engine error (-1); an output that cannot take the bits -- the chain's memory stream with too small a buffer -- is
    reported as -2, distinct from it, and nothing is silently dropped.  In all three ways the stage takes its input."""
    z = np.load(G)
    name = "F_noflip" if "F_noflip" in _names() else _names()[0]
    syms, want = z[name + "/syms"].tobytes(), z[name + "/stdout"].tobytes()
    env = dict(os.environ, VDECODE_WHOLE=whole)
    p = subprocess.run([harness, "-q"] + _args(z, name), input=syms, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600, env=dict(env, VDECODE_TEST_BADBIT="700"))
    assert p.returncode == 2 and b"returned no bit for trellis step 699" in p.stderr and b"RESULT rc=-1" in p.stderr
    assert want.startswith(p.stdout) and len(p.stdout) < len(want)
    cap = len(want) - 100
    p = subprocess.run([harness, "-q"] + _args(z, name), input=syms, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600, env=dict(env, VDECODE_TEST_OUTCAP=str(cap)))
    assert p.returncode == 3 and b"short write on the output" in p.stderr and b"RESULT rc=-2" in p.stderr
    # (what did fit is a prefix of the true output; glibc's memory stream ends a full buffer with a NUL byte)
    assert len(p.stdout) <= cap and want.startswith(p.stdout.rstrip(b"\0"))
    p = subprocess.run([harness, "-q"] + _args(z, name), input=syms, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600, env=dict(env, VDECODE_TEST_OUTCAP=str(len(want) + 1)))
    assert p.returncode == 0 and p.stdout == want
