"""CPU: pin the oracle's port-semantics Viterbi to the fixtures minted from viterbi224_port.c
(tests/golden/make_golden.py).  No GPU, no reference needed at run time."""
import os

import numpy as np
import pytest

import orc

G = os.path.join(orc.GOLDEN, "viterbi_framed.npz")
S = os.path.join(orc.GOLDEN, "viterbi_stream.npz")


def _cases():
    z = np.load(G)
    return [str(n) for n in z["names"]]


def _run(dec, z, name):
    syms, nbits, length = z[name + "/syms"], int(z[name + "/nbits"]), int(z[name + "/length"])
    dec.init(int(z[name + "/start"]))
    dec.update(syms, nbits)
    return dec.chainback(nbits, int(z[name + "/end"]))


@pytest.mark.parametrize("name", _cases())
def test_fast_oracle_matches_port_fixture(name):
    z = np.load(G)
    nbits, length = int(z[name + "/nbits"]), int(z[name + "/length"])
    o = orc.OracleV224(length, orc.FAST)
    data = _run(o, z, name)
    assert np.array_equal(data, z[name + "/port_data"])
    # every decision row, not only the survivor path
    nrows = min(nbits, length)
    want = z[name + "/port_rowhash"]
    if nbits <= length:
        got = np.array([o.row_hash(i) for i in range(nrows)], dtype=np.uint64)
        assert np.array_equal(got, want)
    else:                                   # ring wrapped: row r holds step r + len*floor(..)
        got = np.array([o.row_hash(i) for i in range(length)], dtype=np.uint64)
        assert np.array_equal(got, want)
    assert o.dp() == int(z[name + "/port_dp"])
    assert o.spread() == int(z[name + "/port_spread"])
    best = [o.decodebit(d, -1) for d in (1, 24, min(nbits, length))]
    assert best == [int(b) for b in z[name + "/port_decodebit_best"]]
    o.close()


@pytest.mark.parametrize("name", ["erasure128", "wrap_len50"])
def test_literal_oracle_matches_port_fixture(name):
    z = np.load(G)
    o = orc.OracleV224(int(z[name + "/length"]), orc.LITERAL)
    assert np.array_equal(_run(o, z, name), z[name + "/port_data"])
    assert o.spread() == int(z[name + "/port_spread"])
    o.close()


def test_sse2_differs_from_port_on_ties():
    """the SSE2 decoder is NOT bit-identical to the port (tie rule / bias)."""
    z = np.load(G)
    differs = [n for n in _cases() if not np.array_equal(z[n + "/sse2_data"], z[n + "/port_data"])]
    assert "erasure128" in differs or "uniform256" in differs


def test_stream_decodebit_prefix():
    z = np.load(S)
    syms, delay, length = z["syms"], int(z["delay"]), int(z["length"])
    want = np.unpackbits(z["port_bits"])[: int(z["nout"])]
    n = 1500                                     # bits replayed here (full length on the GPU)
    o = orc.OracleV224(length, orc.FAST)
    o.init(0)
    got = []
    for u in range(n):
        o.update(syms[2 * u:2 * u + 2], 1)
        if u >= delay:
            got.append(o.decodebit(delay, 0))
    assert np.array_equal(np.array(got, np.uint8), want[: n - delay])
    o.close()


def test_encoder_known_answer():
    """sync_vector[34] of vdecode.c:27-30 = symbols 46..79 of encode({12 fc 81 9f be 00..})."""
    sync_vector = [0, 1, 1, 1, 1, 1, 1, 0, 1, 0, 1, 1, 1, 1, 0, 0, 1,
                   1, 0, 0, 1, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]
    data = np.array([0x12, 0xfc, 0x81, 0x9f, 0xbe, 0, 0, 0], dtype=np.uint8)
    syms, _ = orc.encode(data, 0)
    assert list(syms[46:80]) == sync_vector


def test_spread_bound():
    """Metric spread stays below 1000 + 23*510 (what makes 16-bit metrics exact)."""
    o = orc.OracleV224(64, orc.FAST)
    o.init(0)
    hard = (orc.gen_uniform(77, 128) & 1) * 255      # worst case: saturated symbols
    worst = 0
    for u in range(64):
        o.update(hard[2 * u:2 * u + 2].astype(np.uint8), 1)
        worst = max(worst, o.spread())
    assert worst < 1000 + 23 * 510
    o.close()


def test_oracle_decodeword_matches_sse2_reference_on_clean_streams():
    """decodeword exists only in viterbi224_sse2.c (:206-243).  On a clean coded stream the SSE2 and the
    port decoder keep the same survivor, so the oracle's restatement (port decisions) must return the reference's
    words: every query of tests/golden/decodeword_sse2.npz, incl. best-state search and ring wrap."""
    z = np.load(os.path.join(orc.GOLDEN, "decodeword_sse2.npz"))
    for name in [str(n) for n in z["names"]]:
        nbits, length = int(z[name + "/nbits"]), int(z[name + "/length"])
        o = orc.OracleV224(length, orc.FAST)
        o.init(0)
        o.update(z[name + "/syms"], nbits)
        got = [o.decodeword(int(d), int(e)) for d, e in z[name + "/queries"]]
        assert got == [int(w) for w in z[name + "/sse2_words"]], name
        o.close()


@pytest.mark.parametrize("mode", [orc.FAST, orc.LITERAL], ids=["fast", "literal"])
def test_oracle_resumes_from_exported_metrics(mode):
    """orc_v224_set_metrics (the hook the deep-stream GPU test uses): a decoder seeded with another decoder's path
    metrics -- shifted by a constant, only differences matter (port.c:171-181) -- writes the same decision rows and ends
    with the same relative metrics as the decoder that ran straight through."""
    n1, n2 = (120, 90) if mode == orc.FAST else (14, 10)
    syms, _ = orc.gen_coded_stream(77, n1 + n2, 2.0, 24.0, 0)
    a = orc.OracleV224(n1 + n2, mode)
    a.init(0); a.update(syms, n1 + n2)
    b = orc.OracleV224(n1, mode)
    b.init(0); b.update(syms[:2 * n1], n1)
    c = orc.OracleV224(n2, mode)
    c.set_metrics(b.get_metrics() + np.uint32(777))
    c.update(syms[2 * n1:], n2)
    assert c.dp() == 0                                    # ring of n2 rows, written once round
    for r in range(n2):
        assert c.row_hash(r) == a.row_hash(n1 + r), "row %d after the resume differs" % r
    assert np.array_equal(c.get_metrics(), a.get_metrics())
    assert int(a.get_metrics().min()) == 0 and int(a.get_metrics().max()) == a.spread()
