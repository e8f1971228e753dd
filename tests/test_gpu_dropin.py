"""GPU: the drop-in claim itself.  The reference's UNMODIFIED vdecode.c and vtest224.c, linked against
libviterbi224_hip.so in place of viterbi224_port.o (oracle/Makefile -> oracle/_ref/*_hiplink; prebuilt
in the build container, /root/reference is not needed at run time), behave like the port build."""
import os
import re
import subprocess

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu
VG = os.path.join(orc.GOLDEN, "vdecode_cli.npz")


def _exe(name):
    p = os.path.join(orc.REF_DIR, name)
    if not os.path.exists(p):
        pytest.skip("%s not prebuilt (oracle/_ref travels from the build container)" % name)
    return p


@pytest.mark.parametrize("name", ["forced_F", "delay64"])
def test_reference_vdecode_source_on_hip_library(name):
    z = np.load(VG)
    args = ["-q"] + [a for a in z[name + "/args"] if a]
    p = subprocess.run([_exe("vdecode_hiplink")] + args, input=z[name + "/syms"].tobytes(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-1000:]
    assert p.stdout == z[name + "/stdout"].tobytes()


def test_reference_vtest224_source_on_hip_library():
    """vtest224 -e 5: random frames through its own AWGN channel; at 5 dB the decoder must be clean."""
    p = subprocess.run([_exe("vtest224_hiplink"), "-l", "512", "-n", "3", "-e", "5"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-1000:]
    m = re.search(rb"BER (\d+)/(\d+).*FER (\d+)/(\d+)", p.stdout)
    assert m, p.stdout
    assert int(m.group(1)) == 0 and int(m.group(2)) == 3 * 512 and int(m.group(3)) == 0


def test_reference_hybridtest_source_on_hip_library():
    """hybridtest.c (Fano first, create / init / update / chainback / delete per failed frame, hybridtest.c:186-193):
    with one Fano move per bit every frame falls through to the Viterbi decoder, which at 4.5 dB must return the data."""
    p = subprocess.run([_exe("hybridtest_hiplink"), "-m", "1", "-n", "4", "-e", "4.5"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-1000:]
    m = re.search(rb"Viterbi attempts (\d+) good frames: (\d+) frame errors (\d+)", p.stdout)
    assert m, p.stdout
    assert int(m.group(1)) >= 3 and int(m.group(2)) == int(m.group(1)) and int(m.group(3)) == 0


def test_reference_decode_source_on_hip_library():
    """decode.c -V (frame sync, init(sync state) / update(1024) / chainback per frame,
    hex frame dump) unmodified on the HIP library == the same program on viterbi224_port.c."""
    z = np.load(os.path.join(orc.GOLDEN, "decode_cli.npz"))
    p = subprocess.run([_exe("decode_hiplink"), "-V"], input=z["syms"].tobytes(), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-1000:]
    def body(b):        # drop the banner lines, which start with argv[0]
        return [l for l in b.split(b"\n") if b"_ref:" not in l and b"_hiplink:" not in l]
    got, want = body(p.stdout), body(z["stdout"].tobytes())
    assert got == want
    assert any(l.endswith(b"12 fc 81 9f be") for l in got) and sum(l.startswith(b"Frame ") for l in got) >= 3


# ---- the product's own framed stage (isee3-decoder_amd/bin/decode) --------------------------
def _decode_cases():
    import test_decode_host
    return test_decode_host.cases()


@pytest.mark.parametrize("case", _decode_cases(), ids=lambda c: c[0])
def test_decode_stage_byte_exact(case):
    """bin/decode -V (frame sync + batches of frames through v224hip_decode_frames on two decoders) prints exactly
    what the reference's decode.c -V printed on the port decoder, lost lock and symbol slip included."""
    import test_decode_host
    from conftest import load_pkg
    name, args, syms, want = case
    exe = load_pkg().cli_path("decode")
    p = test_decode_host.run_as(exe, test_decode_host.argv0_of(want), args + ["-v"], syms.tobytes(), timeout=300)
    assert p.returncode == 0, p.stderr
    assert p.stdout == want
    assert b"batches" in p.stderr
