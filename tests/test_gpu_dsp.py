"""GPU parity of the pmdemod / symdemod kernels (libisee3dsp_hip.so through its C-ABI) and of the
three C pipe stages (bin/pmdemod, bin/symdemod, bin/vdecode).

symdemod / vdecode: byte-identical to the REFERENCE's stdout (fixtures minted from symdemod.c /
vdecode.c + viterbi224_port.c).  pmdemod: within 1e-9 of the oracle restatement's double path
(tolerance: |gpu - ref| <= 1e-9 * max|ref| per block; the reference stage itself cannot be built
here -- FFTW3 missing -- so this parity is UNPINNED, see oracle/pmdemod_oracle.c)."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest

import orc
from conftest import load_pkg
from test_dsp_host import sym_input, _pm_args

pytestmark = pytest.mark.gpu
SG = os.path.join(orc.GOLDEN, "symdemod_cli.npz")
PG = os.path.join(orc.GOLDEN, "pmdemod_oracle.npz")
VG = os.path.join(orc.GOLDEN, "vdecode_cli.npz")


@pytest.fixture(scope="module")
def pkg():
    return load_pkg()


def _run(exe, args, data, timeout=900):
    p = subprocess.run([exe] + list(args), input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return p.stdout


@pytest.mark.parametrize("name", [str(n) for n in np.load(SG)["names"]])
def test_symdemod_cli_byte_exact(pkg, name):
    z = np.load(SG)
    out = _run(pkg.cli_path("symdemod"), [str(a) for a in z[name + "/args"]], sym_input(z, name).tobytes())
    assert out == z[name + "/stdout"].tobytes()


@pytest.mark.parametrize("name", [str(n) for n in np.load(VG)["names"]])
def test_vdecode_cli_byte_exact(pkg, name):
    z = np.load(VG)
    args = ["-q"] + [a for a in z[name + "/args"] if a]
    out = _run(pkg.cli_path("vdecode"), args, z[name + "/syms"].tobytes())
    assert out == z[name + "/stdout"].tobytes()


def test_symd_primitives_vs_oracle(pkg):
    bb, _ = orc.gen_baseband(31, 250000.0, 2.2, amp=700.0, noise_sigma=5000.0)
    ss = 250000 / 1024.545058
    nsym, first = 1024, 122
    eng = pkg.SymDemodEngine(len(bb))
    eng.load(bb)
    half = 0.5 * ss
    sw, sc = [0], half
    for _ in range(2 * nsym):
        sw.append(int(np.rint(sc))); sc += half          # np.rint = round-half-even = nearbyint
    first_off = int(-ss / 2)
    noff = len([o for o in range(first_off, 10000) if o < ss / 2])
    en = eng.timesearch(first + first_off, sw, 1, nsym, noff)
    import ctypes as C
    ph = C.c_int(0)
    best = orc.lib().orc_timesearch(C.byref(ph), bb.ctypes.data_as(orc.i16p), first, ss, ss, 1, nsym)
    bi = int(np.argmax(en))                              # first maximum
    assert first_off + bi == ph.value and en[bi] / nsym == best
    # final demod at the chosen phase
    fs = first + ph.value
    edges, sc = [fs], fs + half
    for _ in range(2 * nsym):
        edges.append(int(np.rint(sc))); sc += half
    gain = 100. / math.sqrt(best)
    out, esum = eng.demod(edges, 1, nsym, gain)
    ref = np.zeros(nsym, np.uint8)
    e_ref = orc.lib().orc_trial_demod(bb.ctypes.data_as(orc.i16p), fs, ss, 1, nsym, gain, ref.ctypes.data_as(orc.u8p))
    assert np.array_equal(out, ref) and esum / nsym == e_ref
    eng.close()


def _tau(x):
    return 0.25 * math.log(3 * x * x + 6 * x + 1) - math.sqrt(6.) / 24 * math.log((x + 1 - math.sqrt(2 / 3.)) / (x + 1 + math.sqrt(2 / 3.)))


def _gpu_pmdemod(pkg, iq, samprate, binsize, flip=False):
    """Python mirror of cli/pmdemod_core.c's block loop (full-band search), returning pre-quantised doubles"""
    N = 1 << int(np.rint(np.log2(samprate / binsize)))
    eng = pkg.PmDemodEngine(N)
    nb = (len(iq) // 2) // N
    outs, pres, reps = [], [], []
    for b in range(nb):
        eng.load(iq[2 * b * N:2 * (b + 1) * N], flip)
        pk = eng.fft_peak(0, N)
        ap = (pk.next_re * pk.peak_re + pk.next_im * pk.peak_im) / pk.maxenergy
        dp = -ap / (1 - ap)
        am = (pk.prev_re * pk.peak_re + pk.prev_im * pk.peak_im) / pk.maxenergy
        dm = am / (1 - am)
        d = (dp + dm) / 2 + _tau(dp * dp) - _tau(dm * dm)
        cf = (samprate / N) * (pk.peak + d)
        if cf > samprate / 2:
            cf -= samprate
        mx, o16, pre = eng.mix_quantise(2 * math.pi * cf / samprate)
        cn0 = 10 * math.log10(samprate * mx.amplitude ** 2 / (2 * mx.diffsumsq))
        outs.append(o16); pres.append(pre); reps.append((pk.peak, cf, cn0))
    eng.close()
    return np.concatenate(outs), np.concatenate(pres), reps, N


def _check_pm(out, pre, reps, N, ref_out, ref_pre, ref_rep, tol=1e-9):
    assert [r[0] for r in reps] == [r["peak"] for r in ref_rep]
    for (pk, cf, cn0), r in zip(reps, ref_rep):
        assert abs(cf - r["carrier_freq"]) <= 1e-9 * max(1.0, abs(r["carrier_freq"]))
        assert abs(cn0 - r["cn0"]) <= 1e-7
    worst = 0.0
    for b in range(len(reps)):
        a, r = pre[b * N:(b + 1) * N], ref_pre[b * N:(b + 1) * N]
        scale = np.max(np.abs(r))
        worst = max(worst, float(np.max(np.abs(a - r)) / scale))
    assert worst <= tol, "pre-quantisation doubles differ by %.3g relative" % worst
    # int16: identical except where the double sits within tol*scale of an integer boundary
    diff = np.flatnonzero(out != ref_out)
    for i in diff:
        assert abs(int(out[i]) - int(ref_out[i])) == 1
        assert abs(ref_pre[i] - np.rint(ref_pre[i])) <= 1e-6
    return worst


@pytest.mark.parametrize("name", ["b4_r16384", "b4_r16384_neg_flip"])
def test_pmd_kernels_vs_oracle_fixture(pkg, name):
    z = np.load(PG)
    cfg = z[name + "/cfg"]
    out, pre, reps, N = _gpu_pmdemod(pkg, z[name + "/iq"], float(cfg[0]), float(cfg[1]), bool(cfg[6]))
    ref_rep = [dict(peak=int(p), carrier_freq=float(c), cn0=float(n)) for p, c, n in
               zip(z[name + "/peak"], z[name + "/carrier_freq"], z[name + "/cn0"])]
    _check_pm(out, pre, reps, N, z[name + "/out"], z[name + "/pre"], ref_rep)


@pytest.mark.parametrize("lg,fc", [(18, 12345.678), (22, -1234567.891)])
def test_pmd_large_block_closed_form_carrier(pkg, lg, fc):
    """N = 2^18 (config 3) and 2^22: the closed-form spin-down must track the reference's
    sequential recurrence to 1e-9 even after millions of steps."""
    N = 1 << lg
    fs = float(N)                                   # 1 Hz bins
    iq, _ = orc.gen_iq(50 + lg, fs, 1.0, fc_hz=fc, amp=3000.0, cn0_dbhz=45.0 + 10 * math.log10(fs / 250000.0))
    ref_out, ref_pre, ref_rep, n2 = orc.pmdemod(iq, samprate=fs, binsize=1.0)
    assert n2 == N
    out, pre, reps, _ = _gpu_pmdemod(pkg, iq, fs, 1.0)
    w = _check_pm(out, pre, reps, N, ref_out, ref_pre, ref_rep)
    print("N=2^%d worst relative deviation %.3g" % (lg, w))


def test_pmd_fft_vs_numpy(pkg):
    N = 1 << 16
    rng = np.random.default_rng(3)
    iq = rng.integers(-30000, 30000, 2 * N).astype(np.int16)
    eng = pkg.PmDemodEngine(N)
    eng.load(iq)
    eng.fft_peak(0, N)
    got = eng.spectrum()
    want = np.fft.fft(iq[0::2].astype(np.float64) + 1j * iq[1::2].astype(np.float64))
    assert np.max(np.abs(got - want)) <= 1e-12 * np.max(np.abs(want))
    eng.close()


def _np_peak(x, first, last):
    """pmdemod.c:255-279 on numpy's double transform: the LAST maximum of |X|^2 in [first, last) and its neighbours."""
    X = np.fft.fft(x)
    e = X.real * X.real + X.imag * X.imag
    seg = e[first:last]
    k = first + (len(seg) - 1 - int(np.argmax(seg[::-1])))
    return k, X


def _search_case(kind, N, seed):
    rng = np.random.default_rng(seed)
    n = np.arange(N)
    first, last, lo = 0, N, None
    if kind == "noise":                   # no carrier at all: the largest of N exponentially distributed bins
        x = rng.integers(-30000, 30000, N) + 1j * rng.integers(-30000, 30000, N)
    elif kind == "tone":                  # a carrier between two bins, 10 dB under the noise per sample
        f = (rng.integers(1, N - 1) + rng.uniform(-0.5, 0.5)) / N
        x = np.rint(900.0 * np.exp(2j * np.pi * f * n) + rng.normal(0, 2000, N) + 1j * rng.normal(0, 2000, N))
    elif kind == "range":                 # a strong carrier OUTSIDE the searched bins, a weak one inside
        x = np.rint(8000.0 * np.exp(2j * np.pi * (N // 8 + 0.3) / N * n) + 300.0 * np.exp(2j * np.pi * (N // 2 + 77.2) / N * n)
                    + rng.normal(0, 500, N) + 1j * rng.normal(0, 500, N))
        first, last = N // 2 - 1000, N // 2 + 1000
    elif kind == "dechirp":               # a chirped carrier, taken out by the LO table (pmdemod.c:232-244)
        rate = 3.7e-9
        lo = np.exp(2j * np.pi * 0.5 * rate * n * n)
        x = np.rint(2500.0 * np.exp(2j * np.pi * ((N // 3 + 0.41) / N * n + 0.5 * rate * n * n))
                    + rng.normal(0, 1500, N) + 1j * rng.normal(0, 1500, N))
    elif kind == "edge":                  # the peak at bin 0: its lower neighbour is bin N - 1
        x = np.rint(3000.0 * np.exp(2j * np.pi * 0.12 / N * n) + rng.normal(0, 800, N) + 1j * rng.normal(0, 800, N))
    else:
        raise ValueError(kind)
    iq = np.empty(2 * N, np.int16)
    iq[0::2] = np.clip(x.real, -32768, 32767)
    iq[1::2] = np.clip(x.imag, -32768, 32767)
    return iq, first, last, lo


@pytest.mark.parametrize("kind,lg,seed", [("noise", 16, 1), ("noise", 16, 2), ("noise", 18, 3), ("noise", 20, 4), ("tone", 12, 5),
                                          ("tone", 16, 6), ("tone", 18, 7), ("tone", 22, 8), ("range", 18, 9), ("dechirp", 18, 10),
                                          ("edge", 16, 11), ("noise", 13, 12), ("tone", 15, 13), ("noise", 17, 14)])
def test_pmd_search_transform_names_the_double_transforms_peak(pkg, kind, lg, seed):
    """The peak search goes through a single-precision transform and evaluates the named bins exactly (include/
    isee3_dsp_hip.h, pmd_last_peak_path): same bin as the double transform under pmdemod.c's rule, values within 1e-12."""
    N = 1 << lg
    iq, first, last, lo = _search_case(kind, N, seed)
    eng = pkg.PmDemodEngine(N)
    if lo is not None:
        eng.set_dechirp(lo)
    eng.load(iq)
    pk = eng.fft_peak(first, last)
    assert eng.last_peak_path() == 1, "the search transform should have named the peak of this block itself"
    x = iq[0::2].astype(np.float64) + 1j * iq[1::2].astype(np.float64)
    if lo is not None:
        x = x * np.conj(lo)
    k, X = _np_peak(x, first, last)
    assert pk.peak == k
    scale = abs(X[k])
    for got, want in ((complex(pk.peak_re, pk.peak_im), X[k]), (complex(pk.next_re, pk.next_im), X[(k + 1) % N]),
                      (complex(pk.prev_re, pk.prev_im), X[(k - 1) % N])):
        assert abs(got - want) <= 1e-12 * scale
    assert abs(pk.maxenergy - scale * scale) <= 1e-12 * scale * scale
    # the double transform on request (test hook), and the next peak search is a search again
    sp = eng.spectrum()
    assert np.max(np.abs(sp - X)) <= 1e-12 * np.max(np.abs(X))
    assert eng.fft_peak(first, last).peak == k and eng.last_peak_path() == 1
    eng.close()


def test_pmd_search_transform_falls_back_where_single_precision_cannot_decide(pkg, monkeypatch):
    """An all-zero block (every bin ties: pmdemod.c's `>=` takes the last one), two tones of equal amplitude (one workgroup
    of the last pass or two: within 2^-10 of each other either way) and the forced fall-back."""
    N = 1 << 16
    eng = pkg.PmDemodEngine(N)
    eng.load(np.zeros(2 * N, np.int16))
    pk = eng.fft_peak(100, 5000)
    assert eng.last_peak_path() == 2 and pk.peak == 4999 and pk.maxenergy == 0.0
    # two exact tones of equal amplitude: bins a and b tie up to rounding -- whatever the double transform says, goes
    n = np.arange(N)
    for a, b in ((1000, 1000 + 256), (1000, 30000)):
        x = np.rint(4000.0 * (np.exp(2j * np.pi * a / N * n) + np.exp(2j * np.pi * b / N * n)))
        iq = np.empty(2 * N, np.int16)
        iq[0::2] = x.real
        iq[1::2] = x.imag
        eng.load(iq)
        pk = eng.fft_peak(0, N)
        path = eng.last_peak_path()
        sp = eng.spectrum()
        e = sp.real * sp.real + sp.imag * sp.imag
        k = N - 1 - int(np.argmax(e[::-1]))
        if path == 2:                     # decided by the double transform: exactly its arg-max, exactly its values
            assert pk.peak == k and pk.peak_re == sp[k].real and pk.peak_im == sp[k].imag
        else:                             # both bins evaluated exactly (quantisation leaves them ~1e-6 apart): the larger one
            assert path == 1 and pk.peak in (a, b) and (pk.peak == k or abs(e[a] - e[b]) <= 1e-12 * e[a])
    eng.close()
    monkeypatch.setenv("ISEE3DSP_FFT_F32_FORCE_FALLBACK", "1")
    code = ("import sys; sys.path.insert(0, %r); import importlib, numpy as np\n"
            "pkg = importlib.import_module('isee3-decoder_amd')\n"
            "N = 1 << 14; rng = np.random.default_rng(5); iq = rng.integers(-20000, 20000, 2 * N).astype(np.int16)\n"
            "e = pkg.PmDemodEngine(N); e.load(iq); pk = e.fft_peak(0, N)\n"
            "X = np.fft.fft(iq[0::2] + 1j * iq[1::2]); p = X.real ** 2 + X.imag ** 2\n"
            "assert e.last_peak_path() == 2 and pk.peak == N - 1 - int(np.argmax(p[::-1])), (e.last_peak_path(), pk.peak)\n"
            "print('ok')\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("name", [str(n) for n in np.load(PG)["names"]])
def test_pmdemod_cli_vs_oracle(pkg, name):
    z = np.load(PG)
    out = np.frombuffer(_run(pkg.cli_path("pmdemod"), _pm_args(z[name + "/cfg"]), z[name + "/iq"].tobytes()), np.int16)
    ref, pre = z[name + "/out"], z[name + "/pre"]
    assert len(out) == len(ref)
    bad = np.flatnonzero(out != ref)
    if "doppler" in name:
        # de-chirp runs the reference's own recurrence on the host: still identical up to boundaries
        pass
    for i in bad:
        assert abs(int(out[i]) - int(ref[i])) == 1 and abs(pre[i] - np.rint(pre[i])) <= 1e-5
    assert len(bad) <= max(2, len(ref) // 100000)


@pytest.mark.parametrize("name", [str(n) for n in np.load(PG)["names"]])
def test_pmdemod_cli_same_samples_whichever_transform_finds_the_peak(pkg, name, monkeypatch):
    """The stage's int16 output with the single-precision search transform (default) and with the double transform
    (ISEE3DSP_FFT_F64=1): Quinn's three bins are double-precision values either way, so the carrier estimates agree to
    ~1e-13 and the samples may differ by one LSB only where the value sits on an integer boundary."""
    z = np.load(PG)
    args, iq = _pm_args(z[name + "/cfg"]), z[name + "/iq"].tobytes()
    a = np.frombuffer(_run(pkg.cli_path("pmdemod"), args, iq), np.int16)
    monkeypatch.setenv("ISEE3DSP_FFT_F64", "1")
    b = np.frombuffer(_run(pkg.cli_path("pmdemod"), args, iq), np.int16)
    assert len(a) == len(b)
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    assert d.max(initial=0) <= 1 and np.count_nonzero(d) <= max(2, len(a) // 100000)


def test_full_chain_cli(pkg):
    """pmdemod | symdemod | vdecode on a synthetic PM capture recovers the sent telemetry bits."""
    fs = 32768.0
    iq, sent = orc.gen_iq(88, fs, 7.0, fc_hz=2345.6, amp=3000.0, cn0_dbhz=48.0)
    bb = _run(pkg.cli_path("pmdemod"), ["-q", "-r", str(fs), "-b", "1"], iq.tobytes())
    sy = _run(pkg.cli_path("symdemod"), ["-q", "-r", str(int(fs)), "-c", "1024"], bb)
    bits = _run(pkg.cli_path("vdecode"), ["-q"], sy)
    got = np.frombuffer(bits, np.uint8) - ord("0")
    assert len(got) > 2000
    s = "".join(map(str, sent))
    g = "".join(map(str, got[100:1100]))
    assert g in s, "decoded bit run not found in the transmitted stream"


def test_in_process_chain_equals_three_stage_pipeline(pkg):
    """bin/isee3chain (the three stages as threads of one process) == pmdemod | symdemod | vdecode"""
    fs = 32768.0
    iq, sent = orc.gen_iq(89, fs, 6.0, fc_hz=-3456.7, amp=3000.0, cn0_dbhz=48.0)
    bb = _run(pkg.cli_path("pmdemod"), ["-q", "-r", str(fs), "-b", "1"], iq.tobytes())
    sy = _run(pkg.cli_path("symdemod"), ["-q", "-r", str(int(fs)), "-c", "1024"], bb)
    bits = _run(pkg.cli_path("vdecode"), ["-q"], sy)
    chain = _run(pkg.cli_path("isee3chain"), ["-r", str(int(fs)), "-b", "1", "-c", "1024"], iq.tobytes())
    assert chain == bits and len(bits) > 2000
    # and the same through the library entry point on memory buffers, twice (context reuse)
    for _ in range(2):
        assert pkg.run_chain(iq, samprate=fs, binsize=1.0, symrate="1024") == bits


def test_chain_viterbi_stage_modes_give_the_same_bits(pkg, monkeypatch):
    """The chain's Viterbi stage can take its symbols block by block (long blocks shared between two decoders at the end),
    progressively (one stream fed as it arrives, second decoder joining at the planned cut, v224hip_progressive_*) or
    whole (wait for all, then split): the decoded bits are the same, with the capture in host memory and in HBM."""
    fs = 32768.0
    iq, sent = orc.gen_iq(93, fs, 40.0, fc_hz=2345.6, amp=3000.0, cn0_dbhz=48.0)
    d_iq = pkg.DeviceBuffer.from_numpy(iq)
    got = {}
    for mode in ("block", "progressive", "whole"):
        monkeypatch.setenv("ISEE3_CHAIN_MODE", mode)
        got[mode] = pkg.run_chain(iq, samprate=fs, binsize=1.0, symrate="1024")
        assert pkg.run_chain(d_iq, samprate=fs, binsize=1.0, symrate="1024") == got[mode], mode
    assert got["block"] == got["progressive"] == got["whole"] and len(got["block"]) > 19000
    monkeypatch.setenv("V224HIP_SPLIT_FORCE_FALLBACK", "1")          # the redo path inside the chain
    monkeypatch.setenv("ISEE3_CHAIN_MODE", "progressive")
    assert pkg.run_chain(d_iq, samprate=fs, binsize=1.0, symrate="1024") == got["block"]
    d_iq.free()


def test_chain_beside_a_busy_decoder_thread(pkg):
    """A chain run while another thread of the process keeps a decoder of its own busy (own stream, same GPU): same bits."""
    import threading
    fs = 32768.0
    iq, sent = orc.gen_iq(94, fs, 12.0, fc_hz=-1234.5, amp=3000.0, cn0_dbhz=48.0)
    want = pkg.run_chain(iq, samprate=fs, binsize=1.0, symrate="1024")
    syms, _ = orc.gen_coded_stream(9900, 60_000, 3.0, 24.0, 0)
    dec = pkg.Viterbi224(200 + 2040)
    dec.init(0)
    ref = dec.stream_decode(syms, 200)
    stop, bad = [False], []

    def busy():
        while not stop[0]:
            dec.init(0)
            if not np.array_equal(dec.stream_decode(syms, 200), ref):
                bad.append(1)

    th = threading.Thread(target=busy)
    th.start()
    try:
        for _ in range(3):
            assert pkg.run_chain(iq, samprate=fs, binsize=1.0, symrate="1024") == want
    finally:
        stop[0] = True
        th.join()
    assert not bad
    dec.close()


def test_stress_10msps_block_and_window(pkg):
    """BASELINE configs[4] shapes: 10 MS/s, 1 Hz bins => N = 2^23 FFT blocks, 9 760 samples per symbol
    (timesearch over 9 761 offsets).  One pmdemod block and one symdemod window against the oracle."""
    fs = 10_000_000.0
    N = 1 << 23
    iq, _ = orc.gen_iq(91, fs, N / fs + 1e-4, fc_hz=2_345_678.9, amp=3000.0, cn0_dbhz=45.0 + 10 * math.log10(fs / 250000.0))
    iq = iq[:2 * N]
    ref_out, ref_pre, ref_rep, n2 = orc.pmdemod(iq, samprate=fs, binsize=1.0)
    assert n2 == N and len(ref_rep) == 1
    out, pre, reps, _ = _gpu_pmdemod(pkg, iq, fs, 1.0)
    w = _check_pm(out, pre, reps, N, ref_out, ref_pre, ref_rep)
    print("N=2^23 worst relative deviation %.3g" % w)
    # symdemod at 10 MS/s: window 0.2 s (204 symbols), Symbolsamples 9760.43
    bb, _ = orc.gen_baseband(92, fs, 0.45, amp=600.0, noise_sigma=4000.0)
    sy_ref, ph_ref, en_ref = orc.symdemod(bb, samprate=int(fs), c_opt="1024", window=0.2)
    out = _run(pkg.cli_path("symdemod"), ["-q", "-r", str(int(fs)), "-c", "1024", "-w", "0.2"], bb.tobytes())
    assert out == sy_ref.tobytes() and len(out) >= 204


# ---- edge cases the reference handles implicitly: empty / short / ragged inputs ---------------------
def _ref(exe, args, data):
    p = os.path.join(orc.REF_DIR, exe)
    if not os.path.exists(p):
        pytest.skip("oracle/_ref/%s not prebuilt" % exe)
    return subprocess.run([p] + args, input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, check=True).stdout


@pytest.mark.parametrize("nsym", [0, 1, 2, 399, 401, 403])
def test_vdecode_cli_short_and_ragged_inputs(pkg, nsym):
    """fewer symbols than the decode delay, an odd count, exactly one output bit: same stdout as vdecode.c"""
    z = np.load(VG)
    data = z["forced_F/syms"][:nsym].tobytes()
    want = _ref("vdecode_port_ref", ["-q", "-F"], data)
    assert _run(pkg.cli_path("vdecode"), ["-q", "-F"], data) == want
    assert len(want) == max(0, nsym // 2 - 200)


@pytest.mark.parametrize("seconds", [0.0, 0.3, 1.0, 1.01])
def test_symdemod_cli_short_inputs(pkg, seconds):
    """less than one window of samples => no output, like symdemod.c:124-125"""
    bb, _ = orc.gen_baseband(55, 25000.0, seconds, amp=1500.0, noise_sigma=2000.0)
    args = ["-q", "-r", "25000", "-c", "1024", "-w", "1.0"]
    want = _ref("symdemod_ref", args, bb.tobytes())
    assert _run(pkg.cli_path("symdemod"), args, bb.tobytes()) == want


def test_pmdemod_cli_short_input_and_bad_option(pkg):
    """a partial block is dropped (pmdemod.c:206-216); unknown option => exit 1 (:111-113);
    carrier outside Nyquist => exit 1 (:116-121)"""
    iq, _ = orc.gen_iq(56, 16384.0, 0.2, fc_hz=500.0)
    assert _run(pkg.cli_path("pmdemod"), ["-q", "-r", "16384", "-b", "4"], iq.tobytes()) == b""
    assert _run(pkg.cli_path("pmdemod"), ["-q", "-r", "16384", "-b", "4"], b"") == b""
    p = subprocess.run([pkg.cli_path("pmdemod"), "-Z"], input=b"", stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 1 and b"unknown option" in p.stderr
    p = subprocess.run([pkg.cli_path("pmdemod"), "-r", "1000", "-S", "900"], input=b"", stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 1 and b"outside Nyquist" in p.stderr


def test_segmented_decode_matches_single_pass(pkg):
    """configs[4] concept: one capture cut into overlapped block-aligned segments, decoded independently
    (two chains at a time) and stitched == the single-pass decode, on a clean synthetic signal."""
    from importlib import import_module
    seg = import_module("isee3_decoder_amd.segment")
    fs = 32768.0
    iq, sent = orc.gen_iq(93, fs, 40.0, fc_hz=3210.9, amp=3000.0, cn0_dbhz=50.0)
    single = pkg.run_chain(iq, samprate=fs, binsize=1.0, symrate="1024")
    bits, ok, seams, processed = seg.decode_segmented(iq, fs, 1.0, 4, pkg.run_chain, warm_blocks=7, concurrency=2)
    assert seams == 3 and ok == 3
    # identical wherever both exist; the segmented stream may end a few bits earlier or later
    n = min(len(bits), len(single))
    assert n > 18000 and bits[:n - 64] == single[:n - 64]
    # planning: block-aligned, covering, warm-up clipped at the start
    assert seg.plan_segments(24, 4, 3) == [(0, 0, 6), (3, 6, 12), (9, 12, 18), (15, 18, 24)]
    # segments whose warm-up reaches back to block 0 are merged into the first one
    assert seg.plan_segments(8, 8, 3) == [(0, 0, 4), (1, 4, 5), (2, 5, 6), (3, 6, 7), (4, 7, 8)]


def test_vdecode_cli_file_input_uses_both_decoders_same_output(pkg, tmp_path):
    """`vdecode < file`: all input is known up front, so the stage decodes the stream in two halves on two decoders
    (verified split); `cat file | vdecode` decodes block by block on one.  Same bytes -- through a forced phase flip."""
    nbits = 100_000
    syms, _ = orc.gen_coded_stream(9500, nbits, 3.0, 24.0, 2)
    syms = np.concatenate([syms[:60001], syms[60002:]])            # a lost symbol: the phase tracker has to flip
    f = tmp_path / "syms.u8"
    f.write_bytes(syms.tobytes())
    exe = pkg.cli_path("vdecode")
    piped = subprocess.run([exe, "-q"], input=syms.tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    with open(f, "rb") as fh:
        filed = subprocess.run([exe, "-q"], stdin=fh, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert piped.returncode == 0 and filed.returncode == 0, filed.stderr
    assert len(piped.stdout) > 99_000 and piped.stdout == filed.stdout


def test_vdecode_cli_pipe_mode_shared_blocks_same_output(pkg):
    """`cat file | VDECODE_SHARE=1 vdecode`: long blocks shared between two decoders == the plain pipe mode, byte for byte"""
    nbits = 120_000
    syms, _ = orc.gen_coded_stream(9510, nbits, 3.0, 24.0, 2)
    exe = pkg.cli_path("vdecode")
    plain = subprocess.run([exe, "-q"], input=syms.tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    shared = subprocess.run([exe, "-q"], input=syms.tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600,
                            env=dict(os.environ, VDECODE_SHARE="1"))
    assert plain.returncode == 0 and shared.returncode == 0, shared.stderr
    assert len(plain.stdout) > 119_000 and plain.stdout == shared.stdout


def test_chain_objects_are_reused_and_released(pkg):
    """libisee3chain.so keeps its decoder / pmdemod / symdemod objects between calls (a kept pmdemod handle must not
    carry the previous call's de-chirp table or state): three calls, two settings, same answers as fresh; then release."""
    fs = 16384.0
    iq, _ = orc.gen_iq(4301, fs, 3.2, fc_hz=987.6, amp=3000.0, cn0_dbhz=50.0)
    first = pkg.run_chain(iq, samprate=fs, binsize=4.0, symrate="1024")
    other = pkg.run_chain(iq, samprate=fs, binsize=8.0, symrate="1024")          # another FFT size: another handle
    again = pkg.run_chain(iq, samprate=fs, binsize=4.0, symrate="1024")
    assert again == first and len(first) > 800 and len(other) > 800
    pkg.release_chain_objects()
    assert pkg.run_chain(iq, samprate=fs, binsize=4.0, symrate="1024") == first
    pkg.release_chain_objects()


@pytest.mark.parametrize("name", [str(n) for n in np.load(os.path.join(orc.GOLDEN, "vdecode_stderr.npz"))["names"]])
def test_vdecode_cli_symbol_error_statistic_matches_reference_stderr(pkg, name):
    """bin/vdecode's stderr == the reference vdecode's (vdecode.c:159-184 re-encode tally every -i bits, the phase-flip
    notice, the delay warning): fixture minted from oracle/_ref/vdecode_port_ref, C locale, program name stripped."""
    from test_vdecode_host import status_lines
    z = np.load(os.path.join(orc.GOLDEN, "vdecode_stderr.npz"))
    for whole in ("0", "1"):
        p = subprocess.run([pkg.cli_path("vdecode")] + [str(a) for a in z[name + "/args"]],
                           input=np.load(VG)[name + "/syms"].tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           timeout=600, env=dict(os.environ, LANG="C", LC_ALL="C", VDECODE_WHOLE=whole))
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        assert status_lines(p.stderr) == status_lines(z[name + "/stderr"].tobytes())
        assert p.stdout == np.load(VG)[name + "/stdout"].tobytes()


@pytest.mark.slow
def test_config2_full_size_chain_equals_oracle_chain(pkg):
    """BASELINE configs[2] at its own size: 60 s x 250 kS/s synthetic int16 IQ (bench.py's capture), 1 Hz bins
    (N = 2^18, 57 blocks), 1024 sym/s Manchester, vdecode -d 200.  libisee3chain.so on the GPU == oracle pmdemod ->
    oracle symdemod -> oracle vdecode, bit for bit (pmdemod leg UNPINNED: FFTW3 absent, see oracle/pmdemod_oracle.c;
    where oracle/_ref travelled along, the REFERENCE's symdemod binary is run on the oracle baseband as well)."""
    from importlib import import_module
    synth = import_module("isee3_decoder_amd.synth")
    fs = 250000.0
    iq, sent = synth.iq_capture(3, fs, 60.0, amp=None)
    assert len(iq) == 2 * 15_000_000
    got = pkg.run_chain(iq, samprate=fs, binsize=1.0, symrate="1024", decode_delay=200)
    bb, _, rep, N = orc.pmdemod(iq, samprate=fs, binsize=1.0, want_pre=False)
    assert N == 1 << 18 and len(rep) == 57
    sy, _, _ = orc.symdemod(bb, samprate=int(fs), c_opt="1024")
    if orc.have_ref():
        assert orc.ref_cli("symdemod_ref", ["-q", "-r", str(int(fs)), "-c", "1024"], bb.tobytes()) == sy.tobytes()
    want, _ = orc.vdecode(sy)
    assert len(want) > 29000
    assert got == want
    # and the decoded bits are the sent telemetry (after vdecode has settled its symbol-pair phase)
    s = "".join(map(str, sent))
    assert got[-1100:-100].decode() in s


@pytest.mark.slow
def test_config3_eight_captures_on_one_gpu(pkg, monkeypatch):
    """BASELINE configs[3]: the eight independent captures bench.py deals one per GPU (seeds 3 .. 10, 60 s x 250 kS/s each),
    here one after the other on the one GPU of the box -- the N > 1 form needs the node and differs only in which rank
    runs which seed (tests/test_dist_harness.py).  Every capture: the chain with the capture in HBM, in both Viterbi-stage
    modes, gives the same bits; ~30 000 of them; and a 1 000-bit run from the end is found in the telemetry that was sent.
    Seed 3 is the capture test_config2_full_size_chain_equals_oracle_chain pins to the oracle chain bit for bit."""
    from importlib import import_module
    synth = import_module("isee3_decoder_amd.synth")
    fs = 250000.0
    for seed in range(3, 11):
        iq, sent = synth.iq_capture(seed, fs, 60.0, amp=None)
        d_iq = pkg.DeviceBuffer.from_numpy(iq)
        got = {}
        for mode in ("progressive", "block"):
            monkeypatch.setenv("ISEE3_CHAIN_MODE", mode)
            got[mode] = pkg.run_chain(d_iq, samprate=fs, binsize=1.0, symrate="1024", decode_delay=200)
        d_iq.free()
        assert got["progressive"] == got["block"], "seed %d: the two Viterbi-stage modes differ" % seed
        assert 29500 < len(got["block"]) < 31000, (seed, len(got["block"]))
        assert got["block"][-1100:-100].decode() in "".join(map(str, sent)), "seed %d: decoded run not in the sent stream" % seed
    pkg.release_chain_objects()


def test_config4_form_64_overlapped_segments(pkg):
    """BASELINE configs[4] form at a shortened capture: ONE capture cut into 64 block-aligned segments, each extended to
    the left by a 7-block warm-up, decoded independently (two chains at a time on this GPU) and stitched: all seams
    verified, result == the single-pass decode of the same capture."""
    from importlib import import_module
    seg = import_module("isee3_decoder_amd.segment")
    fs = 16384.0
    iq, sent = orc.gen_iq(97, fs, 64.5, fc_hz=-2222.2, amp=3000.0, cn0_dbhz=50.0)
    single = pkg.run_chain(iq, samprate=fs, binsize=1.0, symrate="1024")
    bits, ok, seams, processed = seg.decode_segmented(iq, fs, 1.0, 64, pkg.run_chain, warm_blocks=7, concurrency=2)
    # 64 one-block segments; the eight whose 7-block warm-up reaches back to the start of the capture are one segment
    assert seams == 56 and ok == 56
    n = min(len(bits), len(single))
    assert n > 30000 and bits[:n - 64] == single[:n - 64]
    assert processed == (8 + 56 * 8) * 16384                  # every later segment really carried its warm-up


def test_symd_window_store_in_hbm(pkg):
    """symd_store_slide / _put / _scan == memmove / memcpy / prefix sums on a host copy of the buffer (symdemod.c:96-125),
    stale tail included; sources in host and in device memory; sizes that are not multiples of the scan tile."""
    rng = np.random.default_rng(77)
    cap = 3 * 4096 + 1234
    eng = pkg.SymDemodEngine(cap)
    host = np.zeros(cap, np.int16)
    n = 0
    first = True
    for slide, add, dev in ((0, 5000, False), (1200, 4321, True), (7000, 6000, False), (3, 1, True), (0, cap, False)):
        if slide:
            eng.store_slide(slide, n)
            host[:n - slide] = host[slide:n].copy()
            n -= slide
        add = min(add, cap - n)
        blk = rng.integers(-32768, 32768, add, dtype=np.int16)
        eng.store_put(n, pkg.DeviceBuffer.from_numpy(blk) if dev else blk)
        host[n:n + add] = blk
        n += add
        eng.store_scan(cap)                              # the whole buffer, stale tail included
        # check through the integrate-and-dump of one 'symbol' per probe: demod returns exact integer sums
        edges = np.array([0, cap // 3, cap], np.int32)
        _, e = eng.demod(edges, 1, 1, 0.0)
        want = -int(host[:cap // 3].astype(np.int64).sum()) + int(host[cap // 3:].astype(np.int64).sum())
        assert e == float(want) ** 2
        for a, b, c in ((1, 2, 3), (4095, 4096, 4097), (n - 2, n - 1, n), (0, n // 2, cap)):
            _, e = eng.demod(np.array([a, b, c], np.int32), 1, 1, 0.0)
            want = -int(host[a:b].astype(np.int64).sum()) + int(host[b:c].astype(np.int64).sum())
            assert e == float(want) ** 2
    eng.close()


def test_chain_on_a_capture_in_device_memory(pkg):
    """isee3_chain_run_dev (capture already in HBM) == isee3_chain_run_mem (capture in host memory) == the oracle chain;
    per-stage engine times are reported"""
    fs = 32768.0
    iq, _ = orc.gen_iq(4310, fs, 9.0, fc_hz=-777.7, amp=3000.0, cn0_dbhz=50.0)
    ms = []
    host = pkg.run_chain(iq, samprate=fs, binsize=1.0, symrate="1024")
    dev = pkg.run_chain(pkg.DeviceBuffer.from_numpy(iq), samprate=fs, binsize=1.0, symrate="1024", stage_ms=ms)
    assert dev == host and len(ms) == 3 and all(m > 0 for m in ms)
    bb, _, _, _ = orc.pmdemod(iq, samprate=fs, binsize=1.0, want_pre=False)
    sy, _, _ = orc.symdemod(bb, samprate=int(fs), c_opt="1024")
    want, _ = orc.vdecode(sy)
    assert dev == want and len(want) > 4000


def test_timesearch_beyond_2_pow_53_takes_the_ordered_path(pkg, monkeypatch):
    """10 MS/s shape: one symbol spans ~9 760 samples, the window's energy passes 2^53 and the exact-integer parallel sum no
    longer equals the reference's ordered double accumulation: the engine must fall back to the ordered kernel (first
    window) and go there directly afterwards (later windows), with the oracle's energies every time."""
    fs = 10_000_000
    bb, _ = orc.gen_baseband(61, float(fs), 0.45, amp=9000.0, noise_sigma=6000.0)
    ss = fs / 1024.545058
    nsym, first = 200, int(ss / 2)
    half = 0.5 * ss
    sw, sc = [0], half
    for _ in range(2 * nsym):
        sw.append(int(np.rint(sc))); sc += half
    sw = np.array(sw, np.int32)
    first_off = int(-ss / 2)
    noff = len([o for o in range(first_off, int(np.ceil(ss / 2))) if o < ss / 2])
    eng = pkg.SymDemodEngine(len(bb))
    eng.load(bb)
    lo = first + first_off
    # oracle energies for a few offsets: ordered double accumulation of (double)(sym*sym)
    P = np.concatenate([[0], np.cumsum(bb.astype(np.int64))])
    def ordered(t):
        e = 0.0
        for i in range(nsym):
            a, b, c = P[lo + t + sw[2 * i]], P[lo + t + sw[2 * i + 1]], P[lo + t + sw[2 * i + 2]]
            s = int(-(b - a) + (c - b))
            e += float(s * s)
        return e
    for _ in range(3):
        en = eng.timesearch(lo, sw, 1, nsym, noff)
        for t in (0, 1, noff // 2, noff - 1):
            assert en[t] == ordered(t)
    assert en.max() >= 2.0 ** 53
    eng.close()


def test_icesync_correlator_vs_oracle(pkg):
    """the FFT sync-vector correlator of icesync.c:55-208 on the GPU FFT (isync_* + include/isee3_icesync.h)
    against the oracle restatement: same peak index, peak value and whole Corr_result within 1e-9 of its maximum
    (PARITY UNPINNED: icesync.c needs FFTW3, absent).  250 kS/s, the reference's 2^20-point transforms, a frame cut out
    of synthetic telemetry baseband -- the peak must sit where the frame's sync symbols begin."""
    fs, symrate = 250000.0, 1024.475
    bb, sent = orc.gen_baseband(71, fs, 2.6, symrate=symrate, amp=1500.0, noise_sigma=3000.0)
    co = pkg.IcesyncCorrelator(fs, symrate)
    ss = fs / symrate
    assert co.synclen == int(34 * ss + 1) and abs(co.framesamples - ss * 2048) < 1e-9
    vec = orc.icesync_sync_vector(ss)
    n = int(np.ceil(co.framesamples))
    L = pkg.dsp_lib()
    for begin in (0, 77777, 140000):
        frame = bb[begin:begin + n + 8]
        got, gmax = co.search(frame, 0, n)
        want, wmax, wres = orc.icesync_search(vec, 1 << 20, frame, co.framesamples, 0, n, True)
        assert got == want and got != pkg.ICESYNC_FAIL
        assert abs(gmax - wmax) <= 1e-9 * wmax
        # the frame's tail + sync symbols really are there: symbols (1024 - 17) * 2 .. of a frame = the vector
        first_sync = ((2048 - 34) * ss - begin) % (2048 * ss)
        assert min(abs(got - first_sync), abs(got - first_sync + 2048 * ss), abs(got - first_sync - 2048 * ss)) < 3
    # whole result array through the primitive API
    import ctypes as C
    h = L.isync_create(1 << 20)
    assert h and L.isync_set_vector(h, vec.ctypes.data, len(vec)) == 0
    frame = np.ascontiguousarray(bb[5000:5000 + n])
    res = np.zeros(1 << 20, np.float64)
    pk, mp = C.c_int(0), C.c_double(0)
    assert L.isync_search(h, frame.ctypes.data, n, 0, 0, n, C.byref(pk), C.byref(mp), res.ctypes.data) == 0
    want, wmax, wres = orc.icesync_search(vec, 1 << 20, frame, float(n), 0, n, True)
    assert pk.value == want and np.max(np.abs(res - wres)) <= 1e-9 * wmax
    # device-resident samples, a search window, the fold rule, the two failure rules (icesync.c:157-158, 188-206)
    d = pkg.DeviceBuffer.from_numpy(frame)
    assert L.isync_search(h, d.ptr, n, 1, 0, n, C.byref(pk), C.byref(mp), None) == 0 and pk.value == want
    lo = want + 5
    assert L.isync_search(h, d.ptr, n, 1, lo, n, C.byref(pk), C.byref(mp), None) == 0
    assert pk.value == orc.icesync_search(vec, 1 << 20, frame, float(n), lo, n)[0] != want
    hi_lo = (1 << 20) - 3000
    assert L.isync_search(h, d.ptr, n, 1, hi_lo, 1 << 21, C.byref(pk), C.byref(mp), None) == 0
    assert pk.value == orc.icesync_search(vec, 1 << 20, frame, float(n), hi_lo, 1 << 21)[0]
    z = np.zeros(n, np.int16)
    assert L.isync_search(h, z.ctypes.data, n, 0, 0, n, C.byref(pk), C.byref(mp), None) == 0 and pk.value == pkg.ICESYNC_FAIL
    L.isync_destroy(h)
    co.close()


def test_two_host_threads_create_and_run_handles_at_once(pkg):
    """The bench's and the segment harness's pattern (harness.run_segments(concurrency=2), the chain's stage threads):
    decoders and pmdemod handles are CREATED and used from two host threads at the same time.  The one-off set-up both
    create paths share (the 15-step kernel's parity table, the FFT passes' LDS attribute) is under a lock / atomic flag;
    every thread's results must equal the oracle's / numpy's, round after round."""
    import threading
    nbits = 30 * 15 + 7
    N = 1 << 16
    work, errors = [], []
    for t in range(2):
        syms, _ = orc.gen_coded_frame(7100 + t, nbits, 2.5)
        o = orc.OracleV224(nbits, orc.FAST)
        o.init(0); o.update(syms, nbits)
        iq = np.random.default_rng(40 + t).integers(-30000, 30000, 2 * N).astype(np.int16)
        work.append((syms, o.chainback(nbits, 0), iq, np.fft.fft(iq[0::2].astype(np.float64) + 1j * iq[1::2].astype(np.float64))))
    gate = threading.Barrier(2)

    def worker(t):
        try:
            syms, want, iq, spec = work[t]
            for rnd in range(4):
                gate.wait(timeout=120)
                d = pkg.Viterbi224(nbits)                      # create -> init -> update -> chainback, 15-step engine
                eng = pkg.PmDemodEngine(N)
                d.init(0); d.update(syms, nbits)
                eng.load(iq); eng.fft_peak(0, N)
                got = d.chainback(nbits, 0)
                sp = eng.spectrum()
                d.close(); eng.close()
                assert np.array_equal(got, want), "thread %d round %d: decoded bytes differ from the oracle" % (t, rnd)
                assert np.max(np.abs(sp - spec)) <= 1e-12 * np.max(np.abs(spec)), "thread %d round %d: spectrum" % (t, rnd)
        except Exception as e:                                 # noqa: BLE001
            errors.append(e)
            gate.abort()

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def test_chain_output_buffer_too_small_is_an_error_in_every_mode(pkg, monkeypatch):
    """isee3_chain_run_mem / _dev with an output buffer that cannot take the decoded bits: rc 2 and a message that says
    so -- never a truncated result -- in block mode (the Viterbi stage stops at its first short write while symdemod is
    still producing symbols) and in progressive / whole mode (noticed when pass 2 writes at the end).  A good call
    afterwards on the same kept objects gives the full result."""
    fs = 32768.0
    iq, _ = orc.gen_iq(95, fs, 30.0, fc_hz=2345.6, amp=3000.0, cn0_dbhz=48.0)
    monkeypatch.setenv("ISEE3_CHAIN_MODE", "block")
    want = pkg.run_chain(iq, samprate=fs, binsize=1.0, symrate="1024")
    assert len(want) > 14000
    for mode in ("block", "progressive", "whole"):
        monkeypatch.setenv("ISEE3_CHAIN_MODE", mode)
        for cap in (1000, len(want) - 1):
            with pytest.raises(RuntimeError, match="could not be written"):
                pkg.run_chain(iq, samprate=fs, binsize=1.0, symrate="1024", out_cap=cap)
        assert pkg.run_chain(iq, samprate=fs, binsize=1.0, symrate="1024", out_cap=len(want) + 1) == want, mode


def test_c_host_survives_a_stage_that_fails_mid_stream(pkg, tmp_path):
    """The same failure seen from a C program, where SIGPIPE still has its default action (Python ignores it): the
    Viterbi stage gives up while symdemod is writing into the pipe between them.  The host must get rc 2 back and go
    on living (the stage drains the pipe before closing it; symdemod's thread blocks SIGPIPE)."""
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "chain_smallcap_host")
    lib = os.path.join(ROOT, "isee3-decoder_amd", "lib")
    subprocess.run(["gcc", "-O2", "-o", exe, os.path.join(ROOT, "tests", "csrc", "chain_smallcap_host.c"), "-L" + lib,
                    "-lisee3chain", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    fs = 32768.0
    iq, _ = orc.gen_iq(96, fs, 60.0, fc_hz=-3456.7, amp=3000.0, cn0_dbhz=48.0)
    for mode in ("block", "progressive"):
        p = subprocess.run([exe, str(fs), "1", "700"], input=iq.tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           timeout=300, env=dict(os.environ, ISEE3_CHAIN_MODE=mode))
        assert p.returncode == 0, (mode, p.returncode, p.stderr[-500:])
        assert p.stdout.startswith(b"rc=2 ") and b"could not be written" in p.stdout, p.stdout
    p = subprocess.run([exe, str(fs), "1", "40000"], input=iq.tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0 and p.stdout.startswith(b"rc=0 n=") and int(p.stdout.split()[1][2:]) > 29000, p.stdout


@pytest.mark.slow
def test_config4_chain_at_10_msps_end_to_end(pkg):
    """BASELINE configs[4] at its real shape: 10 MS/s int16 IQ, `-b 1` => N = 2^23 (pmdemod.c:129-131), `-W 0` full-band
    search (:273-276), 9 760 samples per symbol and 10 M-sample symdemod windows (symdemod.c:89-125).  24 blocks (20.1 s,
    805 MB of IQ) through the in-process chain -- block ring in HBM, device window store, progressive Viterbi stage --
    (a) from host memory and resident in HBM: identical; (b) == the ORACLE chain (pmdemod restatement | symdemod | vdecode,
    port semantics) bit for bit; (c) cut into 6 block-aligned segments with a 7-block warm-up (the first two start at
    block 0 and are one segment: 5 segments, 4 seams), decoded independently two at a time and stitched: every seam
    verified, result == the single pass."""
    from importlib import import_module
    seg = import_module("isee3_decoder_amd.segment")
    fs, N, nblk = 10_000_000.0, 1 << 23, 24
    iq, sent = orc.gen_iq(98, fs, nblk * N / fs + 1e-3, fc_hz=2_345_678.9, amp=3000.0,
                          cn0_dbhz=45.0 + 10 * math.log10(fs / 250000.0))
    iq = np.ascontiguousarray(iq[:2 * nblk * N])
    single = pkg.run_chain(iq, samprate=fs, binsize=1.0, symrate="1024")
    d_iq = pkg.DeviceBuffer.from_numpy(iq)
    assert pkg.run_chain(d_iq, samprate=fs, binsize=1.0, symrate="1024") == single
    d_iq.free()
    assert len(single) > 9000
    got = "".join(map(str, np.frombuffer(single, np.uint8)[3000:4000] - ord("0")))
    assert got in "".join(map(str, sent)), "decoded run not found in the sent telemetry"
    # (b) the oracle chain on the same capture
    bb, _, rep, n2 = orc.pmdemod(iq, samprate=fs, binsize=1.0, want_pre=False)
    assert n2 == N and len(rep) == nblk
    sy, _, _ = orc.symdemod(bb, samprate=int(fs), c_opt="1024")
    del bb
    ref, _ = orc.vdecode(sy)
    assert single == ref, "GPU chain differs from the oracle chain at 10 MS/s"
    # (c) overlapped segments
    bits, ok, seams, processed = seg.decode_segmented(iq, fs, 1.0, 6, pkg.run_chain, warm_blocks=7, concurrency=2)
    assert (seams, ok) == (4, 4)
    assert processed == (8 + 4 * 11) * N
    n = min(len(bits), len(single))
    assert n > 9000 and bits[:n - 64] == single[:n - 64]
    pkg.release_chain_objects()
