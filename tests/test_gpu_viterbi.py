"""GPU parity: libviterbi224_hip.so (through its C-ABI, via ctypes) against
  (a) the fixtures minted from the reference's viterbi224_port.c (tests/golden/*.npz), and
  (b) the CPU oracle on seeded inputs.
Bit-exact everywhere: decoded bytes/bits, EVERY decision row, final path metrics.
"""
import os

import numpy as np
import pytest

import orc
from conftest import load_pkg

pytestmark = pytest.mark.gpu

G = os.path.join(orc.GOLDEN, "viterbi_framed.npz")
S = os.path.join(orc.GOLDEN, "viterbi_stream.npz")

# (engine, k): simple engine + every fused pass width
ENGINES = [(0, 0)] + [(1, k) for k in (2, 3, 4, 5, 6, 7)]
IDS = ["simple"] + ["fused%d" % k for k in (2, 3, 4, 5, 6, 7)]


@pytest.fixture(scope="module")
def pkg():
    return load_pkg()


def _names():
    return [str(n) for n in np.load(G)["names"]]


def _fnv(a):
    import ctypes as C
    a = np.ascontiguousarray(a)
    return int(orc.lib().orc_fnv1a(a.ctypes.data_as(C.c_void_p), a.nbytes))


@pytest.mark.parametrize("engine,k", ENGINES, ids=IDS)
@pytest.mark.parametrize("name", _names())
def test_framed_fixture(pkg, name, engine, k):
    z = np.load(G)
    syms, nbits, length = z[name + "/syms"], int(z[name + "/nbits"]), int(z[name + "/length"])
    d = pkg.Viterbi224(length, engine, k)
    d.init(int(z[name + "/start"]))
    d.update(syms, nbits)
    data = d.chainback(nbits, int(z[name + "/end"]))
    assert np.array_equal(data, z[name + "/port_data"]), "chainback bytes differ from the port"
    want = z[name + "/port_rowhash"]
    rows = range(len(want)) if len(want) <= 200 else list(range(0, 40)) + list(range(len(want) - 40, len(want))) + list(range(40, len(want) - 40, 37))
    for r in rows:
        assert _fnv(d.export_row(r)) == int(want[r]), "decision row %d differs" % r
    m = d.export_metrics()
    assert int(m.min()) == 0
    assert int(m.max()) == int(z[name + "/port_spread"])
    assert _fnv(m) == int(z[name + "/port_metric_hash"]), "final path metrics differ"
    best = [d.decodebit(dl, -1) for dl in (1, 24, min(nbits, length))]
    assert best == [int(b) for b in z[name + "/port_decodebit_best"]]
    d.close()


@pytest.mark.parametrize("engine,k", [(0, 0), (1, 6), (1, 5), (1, 7)], ids=["simple", "fused6", "fused5", "fused7"])
def test_stream_fixture_full(pkg, engine, k):
    """vdecode's calling pattern (update 1 bit + decodebit(200,0)) over the whole 10^4-bit fixture."""
    z = np.load(S)
    syms, delay = z["syms"], int(z["delay"])
    want = np.unpackbits(z["port_bits"])[: int(z["nout"])]
    chunk = 510
    d = pkg.Viterbi224(delay + 2 * chunk, engine, k)
    d.set_option("chunk", chunk)
    d.init(0)
    out = d.stream_decode(syms, delay)
    assert np.all(out[:delay - 1] == 0xff)       # fewer than `delay` steps of history
    assert np.array_equal(out[delay:], want)
    d.close()


def test_reference_call_pattern_one_bit_at_a_time(pkg):
    """Exactly what vdecode.c:145,152 does through the nine-function API, ring len = delay+1."""
    z = np.load(S)
    syms, delay = z["syms"], int(z["delay"])
    want = np.unpackbits(z["port_bits"])[: int(z["nout"])]
    n = 420
    d = pkg.Viterbi224(delay + 1)
    d.init(0)
    got = []
    for u in range(n):
        assert d.update(syms[2 * u:2 * u + 2], 1) == 0
        if u >= delay:
            got.append(d.decodebit(delay, 0))
    assert np.array_equal(np.array(got, np.uint8), want[: n - delay])
    assert d.decodebit(0, 0) == -1                       # port.c:127-129
    d.close()


@pytest.mark.parametrize("engine,k", [(0, 0), (1, 6)], ids=["simple", "fused6"])
def test_seeded_stream_vs_oracle(pkg, engine, k):
    """Noisy coded stream with pure-noise blocks (ties, long unmerged paths) vs the CPU oracle."""
    nbits, delay = 3000, 200
    syms, _ = orc.gen_coded_stream(9001, nbits, 1.5, 24.0, 20)
    o = orc.OracleV224(delay + 1, orc.FAST)
    o.init(0)
    want = []
    for u in range(nbits):
        o.update(syms[2 * u:2 * u + 2], 1)
        want.append(o.decodebit(delay, 0) if u + 1 >= delay else 0xff)
    d = pkg.Viterbi224(delay + 2 * 256, engine, k)
    d.set_option("chunk", 256)
    d.init(0)
    # ragged: feed in uneven pieces so chunk/pass boundaries move around
    got = []
    pos = 0
    for piece in (1, 7, 300, 513, 1024, 5, 1150):
        piece = min(piece, nbits - pos)
        got.append(d.stream_decode(syms[2 * pos:2 * (pos + piece)], delay))
        pos += piece
    got = np.concatenate(got)
    assert pos == nbits
    assert np.array_equal(got, np.array(want, np.uint8))
    # metrics agree with the oracle too (relative to their minimum)
    m = d.export_metrics()
    for st in (0, 1, 12345, (1 << 23) - 1, 0x2aaaaa):
        assert int(m[st]) == o.metric_rel(st)
    o.close()
    d.close()


def test_engines_agree_long_and_decode_is_correct(pkg):
    """Size-independent checks at a larger size: simple and fused engines give identical streams;
    on a clean signal the decoded bits equal the sent bits; a decodeword equals 64 decodebits."""
    nbits, delay = 60000, 200
    syms, sent = orc.gen_coded_stream(9002, nbits, 4.0, 24.0, 0)
    outs = []
    for engine, k in ((0, 0), (1, 6), (1, 4)):
        d = pkg.Viterbi224(delay + 2 * 1020, engine, k)
        d.set_option("chunk", 1020)
        d.init(0)
        outs.append(d.stream_decode(syms, delay))
        if engine == 1 and k == 6:
            w = d.decodeword(64, 0)
            bits = [(w >> (63 - i)) & 1 for i in range(64)]     # bit 63 = last one read
            ref = []
            for dl in range(64, 0, -1):
                ref.append(d.decodebit(dl, 0))
            assert bits == ref
        d.close()
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    # decodebit(delay) after step u returns data bit u - delay - 22 (state holds the last 23 bits)
    dec = outs[0][delay + 23:]
    assert np.array_equal(dec, sent[1:1 + len(dec)]) or np.array_equal(dec, sent[:len(dec)])


def test_api_edge_cases(pkg):
    L = pkg.v224_lib()
    d = pkg.Viterbi224(32)
    assert d.update(np.zeros(0, np.uint8), 0) == 0
    assert L.chainback_viterbi224(d.h, None, 0, 0) == 0     # nbits == 0: nothing touched
    d.init(0x123456)
    m = d.export_metrics()
    assert int(m[0x123456]) == 0 and int(m[0]) == 1000 and int(m.max()) == 1000
    assert d.min_metric() == 0 and d.max_metric() == 1000
    d.init(0xffffff)                                        # masked to 23 bits (port.c:45)
    assert int(d.export_metrics()[0x7fffff]) == 0
    with pytest.raises(RuntimeError):
        d.stream_decode(np.zeros(64, np.uint8), 200)        # ring too short for delay
    d.close()
    with pytest.raises(RuntimeError):
        pkg.Viterbi224(0)


def test_min_max_metric_track_port_scale(pkg):
    """min/max_metric in the port's never-renormalised scale == oracle literal u32 metrics."""
    nbits = 150
    syms = orc.gen_uniform(9003, 2 * nbits)
    o = orc.OracleV224(nbits, orc.LITERAL)
    o.init(5)
    o.update(syms, nbits)
    lo, hi = orc.lib().orc_v224_metric_abs(o.h, 0), orc.lib().orc_v224_metric_abs(o.h, 1)
    for engine, k in ((0, 0), (1, 6), (1, 7)):
        d = pkg.Viterbi224(nbits, engine, k)
        d.init(5)
        d.update(syms, nbits)
        assert (d.min_metric(), d.max_metric()) == (lo, hi)
        d.close()
    o.close()


# ---- engine LDS (8 steps per launch through LDS tiles) -----------------------------------------
@pytest.mark.parametrize("name", _names())
def test_framed_fixture_lds_engine(pkg, name):
    test_framed_fixture(pkg, name, 2, 0)


def test_stream_fixture_full_lds_engine(pkg):
    test_stream_fixture_full(pkg, 2, 0)


def test_seeded_stream_vs_oracle_lds_engine(pkg):
    test_seeded_stream_vs_oracle(pkg, 2, 0)


# ---- engine LDS15 (15 steps per launch, tile-major metric order between launches) --------------
@pytest.mark.parametrize("name", _names())
def test_framed_fixture_lds15_engine(pkg, name):
    """1024-bit frames = 68 full passes + a 4-step remainder through the natural-order kernels: both
    order conversions, every decision row and the final metrics are checked."""
    test_framed_fixture(pkg, name, 3, 0)


def test_stream_fixture_full_lds15_engine(pkg):
    """chunk 510 = 34 passes; ring of 200 + 1020 rows is not a multiple of 15, so passes wrap it."""
    test_stream_fixture_full(pkg, 3, 0)


def test_seeded_stream_vs_oracle_lds15_engine(pkg):
    """ragged pieces and chunk 256 = 17 passes + 1: switches between the two metric orders all the time"""
    test_seeded_stream_vs_oracle(pkg, 3, 0)


def test_lds15_equals_lds8_on_a_long_stream(pkg):
    nbits, delay = 60_000, 200
    syms, _ = orc.gen_coded_stream(9200, nbits, 2.5, 24.0, 5)
    outs = []
    for engine in (2, 3):
        d = pkg.Viterbi224(delay + 2 * 1024, engine, 0)
        d.init(0)
        outs.append(d.stream_decode(syms, delay))
        m = d.export_metrics()
        outs.append(m - m.min())
        d.close()
    assert np.array_equal(outs[0], outs[2]) and np.array_equal(outs[1], outs[3])


def test_long_stream_vs_oracle_100k_bits(pkg):
    """BASELINE configs[0]/[1] at a larger size: 2*10^5 symbols (coded, 3 dB, 5 % pure-noise blocks) through
    the default engine and through the CPU oracle (port semantics) -- every decoded bit equal."""
    nbits, delay = 100_000, 200
    syms, _ = orc.gen_coded_stream(9100, nbits, 3.0, 24.0, 5)
    o = orc.OracleV224(delay + 1, orc.FAST)
    o.init(0)
    want = np.empty(nbits, np.uint8)
    for u in range(nbits):
        o.update(syms[2 * u:2 * u + 2], 1)
        want[u] = o.decodebit(delay, 0) if u + 1 >= delay else 0xff
    d = pkg.Viterbi224(delay + 2 * 1024)
    d.init(0)
    got = d.stream_decode(syms, delay)
    assert np.array_equal(got, want)
    assert d.min_metric() >= 0 and 0 < d.max_metric() - d.min_metric() < 1000 + 23 * 510
    o.close()
    d.close()


@pytest.mark.parametrize("dist", ["uniform", "coded3dB"])
def test_config1_framed_frames_vs_oracle(pkg, dist):
    """BASELINE configs[0] shape (vtest224: init / update(1000) / chainback per frame),
    12 frames per distribution through ONE decoder object, every frame's bytes equal to the oracle's."""
    framebits, nframes = 1000, 12
    d = pkg.Viterbi224(framebits)
    o = orc.OracleV224(framebits, orc.FAST)
    for f in range(nframes):
        if dist == "uniform":
            syms = orc.gen_uniform(7000 + f, 2 * framebits)
        else:
            syms, _ = orc.gen_coded_frame(7100 + f, framebits, 3.0, 24.0)
        for dec in (d, o):
            dec.init(0)
            dec.update(syms, framebits)
        assert np.array_equal(d.chainback(framebits, 0), o.chainback(framebits, 0)), "frame %d" % f
    d.close()
    o.close()


@pytest.mark.parametrize("ndec", [2, 3])
def test_decode_frames_large_batch(pkg, ndec, monkeypatch):
    """A batch of a dozen frames per decoder (frames padded to whole 15-step passes, tracebacks in 16 pieces): the same
    bytes with the one-wave traceback, and the oracle's for frames picked across the batch."""
    framebits, nframes = 600, 12 * ndec + 1
    frames = [orc.gen_coded_frame(7500 + f, framebits, 3.0, 24.0)[0] if f % 5 else orc.gen_uniform(7500 + f, 2 * framebits)
              for f in range(nframes)]
    syms = np.concatenate(frames)
    decs = [pkg.Viterbi224(2 * 600) for _ in range(ndec)]
    got = pkg.decode_frames(decs, syms, nframes, framebits, 0x819fbe, 0x155555)
    monkeypatch.setenv("V224HIP_SERIAL_CHAINBACK", "2")
    plain = pkg.decode_frames(decs, syms, nframes, framebits, 0x819fbe, 0x155555)
    assert np.array_equal(got, plain)
    o = orc.OracleV224(framebits, orc.FAST)
    for f in (0, 1, ndec, nframes // 2, nframes - 1):
        o.init(0x819fbe)
        o.update(frames[f], framebits)
        assert np.array_equal(got[f], o.chainback(framebits, 0x155555)), "frame %d" % f
    o.close()
    for d in decs:
        d.close()


@pytest.mark.parametrize("nbits", [511, 512, 519, 1024, 1031, 3000, 4099])
def test_chainback_in_pieces_equals_serial_walk(pkg, nbits):
    """chainback_viterbi224 walks frames of >= 512 bits in 16 pieces at once, each verified against the piece above
    (k_chainback_par).  Bytes equal the oracle's (port.c:72-101) for bit counts that are not multiples of 8 or 16 bytes, on a
    coded frame (pieces merge: nothing redone), and on pure noise / all-erasure input (paths do not merge within the
    warm-up: the seam check must catch it and walk those pieces again), for non-zero end states too."""
    cases = {
        "coded": orc.gen_coded_stream(7700 + nbits, nbits, 3.0, 24.0)[0],
        "noise": orc.gen_uniform(7800 + nbits, 2 * nbits),
        "erased": np.full(2 * nbits, 128, dtype=np.uint8),
    }
    d = pkg.Viterbi224(nbits)
    o = orc.OracleV224(nbits, orc.FAST)
    redone = {}
    for name, syms in cases.items():
        for dec in (d, o):
            dec.init(0)
            dec.update(syms, nbits)
        d.get_counter("chainback_redone")
        for end in (0, 0x5a5a5a, 0x7fffff):
            assert np.array_equal(d.chainback(nbits, end), o.chainback(nbits, end)), "%s end %x" % (name, end)
        redone[name] = d.get_counter("chainback_redone")
    if nbits < 512:
        assert sum(redone.values()) == 0                      # serial walk: the counter never moves
    else:
        assert redone["coded"] <= 1, redone                   # survivor paths of a decodable frame merge inside the warm-up (3 end states x 15 seams)
        if nbits >= 1024:
            assert redone["noise"] + redone["erased"] > 0, redone     # the check is not vacuous
    d.close()
    o.close()


def test_decode_frames_batch_on_two_decoders(pkg):
    """v224hip_decode_frames: 7 independent 1000-bit frames (vtest224.c:116-118 per frame) spread over two
    decoders on two streams -- every frame's bytes equal to the oracle's, for a non-zero start / end state too."""
    framebits, nframes = 1000, 7
    frames = []
    for f in range(nframes):
        if f % 3 == 2:
            frames.append(orc.gen_uniform(7300 + f, 2 * framebits))
        else:
            frames.append(orc.gen_coded_frame(7200 + f, framebits, 2.5, 24.0)[0])
    syms = np.concatenate(frames)
    o = orc.OracleV224(framebits, orc.FAST)
    decs = [pkg.Viterbi224(framebits), pkg.Viterbi224(framebits)]
    for start, end in ((0, 0), (0x819fbe, 0x2aaaaa)):
        want = []
        for f in range(nframes):
            o.init(start)
            o.update(frames[f], framebits)
            want.append(o.chainback(framebits, end))
        got = pkg.decode_frames(decs, syms, nframes, framebits, start, end)
        assert np.array_equal(got, np.stack(want))
        one = pkg.decode_frames(decs[:1], syms, nframes, framebits, start, end)     # one decoder: same answer
        assert np.array_equal(one, got)
    # the decoders are still good for the ordinary API afterwards
    decs[0].init(0)
    decs[0].update(frames[0], framebits)
    o.init(0); o.update(frames[0], framebits)
    assert np.array_equal(decs[0].chainback(framebits, 0), o.chainback(framebits, 0))
    with pytest.raises(RuntimeError):
        pkg.decode_frames(decs, syms, nframes - 1, framebits + 1)  # ring shorter than a frame
    for d in decs:
        d.close()
    # rings of two padded frames: frames alternate between the ring halves, every traceback runs under the next frame's
    # passes, the frame is padded with erasures to whole 15-step passes -- same bytes (1000 and 1024-bit frames, 1..3 decoders)
    for fb, nd in ((1000, 2), (1024, 2), (1024, 3), (1024, 1), (333, 2)):
        nfr = 9
        fr = [orc.gen_coded_frame(7400 + f, fb if fb % 8 == 0 else fb + (8 - fb % 8), 2.0, 24.0)[0][:2 * fb] if f % 4 else
              orc.gen_uniform(7500 + f, 2 * fb) for f in range(nfr)]
        oo = orc.OracleV224(fb, orc.FAST)
        want = []
        for f in range(nfr):
            oo.init(0x819fbe)
            oo.update(fr[f], fb)
            want.append(oo.chainback(fb, 0x155555))
        oo.close()
        big = [pkg.Viterbi224(2 * ((fb + 14) // 15 * 15)) for _ in range(nd)]
        got = pkg.decode_frames(big, np.concatenate(fr), nfr, fb, 0x819fbe, 0x155555)
        assert np.array_equal(got, np.stack(want)), (fb, nd)
        again = pkg.decode_frames(big, np.concatenate(fr), nfr, fb, 0x819fbe, 0x155555)
        assert np.array_equal(again, got)
        for d in big:
            d.close()
    o.close()


def test_two_decoders_interleaved_streams(pkg):
    """Two LDS15 decoders share the CUs when driven alternately (bench.py --segments-per-gpu 2): both streams
    still decode exactly as a lone decoder does."""
    nbits, delay = 20_400, 200
    streams = [orc.gen_coded_stream(9300 + i, nbits, 2.5, 24.0, 5)[0] for i in range(2)]
    lone = []
    for sy in streams:
        d = pkg.Viterbi224(delay + 2 * 1020)
        d.init(0)
        lone.append(d.stream_decode(sy, delay))
        d.close()
    decs = [pkg.Viterbi224(delay + 2 * 1020) for _ in range(2)]
    dsy = [pkg.DeviceBuffer.from_numpy(sy) for sy in streams]
    dout = [pkg.DeviceBuffer(nbits) for _ in range(2)]
    for d in decs:
        d.init(0)
    slab = 4 * 1020
    for pos in range(0, nbits, slab):
        n = min(slab, nbits - pos)
        for i in range(2):
            decs[i].stream_decode_dev(dsy[i], n, delay, dout[i], sym_offset=2 * pos, out_offset=pos)
    for i in range(2):
        decs[i].sync()
        assert np.array_equal(dout[i].to_numpy(np.uint8), lone[i])
        decs[i].close()


@pytest.mark.parametrize("length", [17, 31, 64])
def test_lds15_ragged_updates_small_ring(pkg, length):
    """update calls of every residue mod 15 on rings that a 15-step pass has to wrap (and that an update laps
    several times): after every call ALL ring rows, a traceback and the metrics equal the oracle's."""
    sizes = [15, 30, 1, 16, 45, 7, 29, 15, 14, 61]
    syms = orc.gen_uniform(7400 + length, 2 * sum(sizes))
    d = pkg.Viterbi224(length, 3, 0)
    o = orc.OracleV224(length, orc.FAST)
    d.init(3)
    o.init(3)
    pos = 0
    for n in sizes:
        d.update(syms[2 * pos:2 * (pos + n)], n)
        o.update(syms[2 * pos:2 * (pos + n)], n)
        pos += n
        for r in range(length):
            assert _fnv(d.export_row(r)) == o.row_hash(r), "row %d after %d steps" % (r, pos)
        dl = min(length - 1, 13)
        assert d.decodebit(dl, 0) == o.decodebit(dl, 0)
        assert d.decodebit(dl, -1) == o.decodebit(dl, -1)
    m = d.export_metrics()
    for st in (0, 3, 255, 256, 32767, 32768, 0x555555, (1 << 23) - 1):
        assert int(m[st]) == o.metric_rel(st)
    d.close()
    o.close()


def test_split_stream_is_verified_and_exact(pkg, monkeypatch):
    """v224hip_stream_decode_split: one stream over 2 and 3 decoders == the same stream through one decoder, bit for
    bit (0xff start-up marks included).  On a coded stream the seams verify (no part redone); whatever the
    comparison at a seam says -- pure noise with a short warm-up, or a forced mismatch -- the output is exact."""
    delay = 200
    for seed, nbits, noise_pct, warm, ndec, force in ((9400, 61200, 5, 14280, 2, False), (9401, 91800, 5, 8160, 3, False),
                                                      (9402, 30600, 100, 2040, 2, False), (9403, 40800, 5, 14280, 2, True),
                                                      (9404, 5000, 5, 14280, 2, False),      # too short to split: one decoder
                                                      (9405, 33333, 5, 4000, 2, False),      # ragged length, warm-up rounded up
                                                      (9406, 25000, 5, 4080, 1, False)):     # ndec = 1
        syms, _ = orc.gen_coded_stream(seed, nbits, 2.5, 24.0, noise_pct)
        d = pkg.Viterbi224(delay + 2 * 1020)
        d.init(0)
        want = d.stream_decode(syms, delay)
        d.close()
        decs = [pkg.Viterbi224(delay + 2 * 1020) for _ in range(ndec)]
        dsy = pkg.DeviceBuffer.from_numpy(syms)
        dout = pkg.DeviceBuffer(nbits)
        if force:
            monkeypatch.setenv("V224HIP_SPLIT_FORCE_FALLBACK", "1")
        nfb = pkg.stream_decode_split(decs, dsy, nbits, delay, dout, warm_bits=warm)
        monkeypatch.delenv("V224HIP_SPLIT_FORCE_FALLBACK", raising=False)
        got = dout.to_numpy(np.uint8)
        assert np.array_equal(got, want), "seed %d" % seed
        if force:
            assert nfb == 1
        elif noise_pct < 100:
            assert nfb == 0, "seed %d: %d parts redone" % (seed, nfb)
        for x in decs:
            x.close()


@pytest.mark.slow
def test_split_equals_single_at_full_bench_size(pkg):
    """BASELINE configs[1] at full size (10^7 symbols, bench.py's own input): two decoders on consecutive parts,
    verified at the seam, give byte for byte what one decoder gives."""
    from importlib import import_module
    synth = import_module("isee3_decoder_amd.synth")
    nbits, delay = 5_000_000, 200
    syms, _, _ = synth.coded_stream(1000, nbits, 3.0, 24.0, 1.0)
    dsy = pkg.DeviceBuffer.from_numpy(syms)
    decs = [pkg.Viterbi224(delay + 2 * 1020) for _ in range(2)]
    d1 = pkg.DeviceBuffer(nbits)
    decs[0].init(0)
    for pos in range(0, nbits, 8160):
        decs[0].stream_decode_dev(dsy, min(8160, nbits - pos), delay, d1, sym_offset=2 * pos, out_offset=pos)
    decs[0].sync()
    one = d1.to_numpy(np.uint8).copy()
    d2 = pkg.DeviceBuffer(nbits)
    assert pkg.stream_decode_split(decs, dsy, nbits, delay, d2) == 0
    assert np.array_equal(d2.to_numpy(np.uint8), one)
    for x in decs:
        x.close()


DW = os.path.join(orc.GOLDEN, "decodeword_sse2.npz")


@pytest.mark.parametrize("engine,k", [(0, 0), (1, 6), (2, 0), (3, 0)], ids=["simple", "fused6", "lds8", "lds15"])
def test_decodeword_vs_sse2_reference_fixture(pkg, engine, k):
    """decodeword_viterbi224 against words minted from the reference's viterbi224_sse2.c:206-243 (the only
    file that implements it) on clean coded streams -- where the SSE2 and the port decoder keep the same survivor --
    incl. the best-state search (end < 0), delays 1..len and a wrapped ring."""
    z = np.load(DW)
    for name in [str(n) for n in z["names"]]:
        nbits, length = int(z[name + "/nbits"]), int(z[name + "/length"])
        d = pkg.Viterbi224(length, engine, k)
        d.init(0)
        d.update(z[name + "/syms"], nbits)
        got = [d.decodeword(int(dl), int(e)) for dl, e in z[name + "/queries"]]
        assert got == [int(w) for w in z[name + "/sse2_words"]], name
        d.close()
    if orc.have_ref():                 # the reference library itself, where oracle/_ref travelled along
        syms, _ = orc.gen_coded_stream(9601, 400, 7.0, 24.0, 0)
        r = orc.RefV224(256, "sse2")
        d = pkg.Viterbi224(256, engine, k)
        for x in (r, d):
            x.init(0)
            x.update(syms, 400)
        for dl, e in ((64, 0), (64, -1), (200, -1), (3, 0x123456)):
            assert d.decodeword(dl, e) == r.decodeword(dl, e)
        r.close()
        d.close()


@pytest.mark.parametrize("engine,k", [(0, 0), (1, 5), (2, 0), (3, 0)], ids=["simple", "fused5", "lds8", "lds15"])
def test_decodeword_vs_oracle_on_noisy_streams(pkg, engine, k):
    """the same function where ties and unmerged paths matter: noisy stream with pure-noise blocks, against the oracle's
    restatement of sse2.c:206-243 on the port's decisions (pinned by test_oracle_decodeword_matches_sse2_reference...)"""
    nbits, length = 700, 320                                   # the ring wraps twice
    syms, _ = orc.gen_coded_stream(9602, nbits, 1.0, 24.0, 30)
    o = orc.OracleV224(length, orc.FAST)
    d = pkg.Viterbi224(length, engine, k)
    for x in (o, d):
        x.init(0)
    pos = 0
    for piece in (1, 14, 15, 16, 300, 354):                    # words are asked for at ragged positions
        o.update(syms[2 * pos:2 * (pos + piece)], piece)
        d.update(syms[2 * pos:2 * (pos + piece)], piece)
        pos += piece
        for dl, e in ((64, 0), (64, -1), (min(pos, length), -1), (1, -1), (33, 0x7fffff)):
            if dl <= min(pos, length):
                assert d.decodeword(dl, e) == o.decodeword(dl, e), (pos, dl, e)
    assert pos == nbits
    o.close()
    d.close()


def test_split_with_long_decode_delays(pkg):
    """vdecode allows any -d (vdecode.c:86-91 only warns above 1024).  The seam window of the split decode must cover
    the traceback depth: delays of 1024 (three rows past one chunk), 1500, and one longer than the requested warm-up
    all give exactly the single decoder's output, 0xff start-up marks included."""
    for seed, nbits, delay, warm in ((9410, 40800, 1024, 14280), (9411, 51000, 1500, 14280), (9412, 61200, 5000, 4080),
                                     (9413, 30600, 1021, 2040)):
        syms, _ = orc.gen_coded_stream(seed, nbits, 2.5, 24.0, 5)
        d = pkg.Viterbi224(delay + 2 * 1020)
        d.init(0)
        want = d.stream_decode(syms, delay)
        d.close()
        decs = [pkg.Viterbi224(delay + 2 * 1020) for _ in range(2)]
        dsy, dout = pkg.DeviceBuffer.from_numpy(syms), pkg.DeviceBuffer(nbits)
        nfb = pkg.stream_decode_split(decs, dsy, nbits, delay, dout, warm_bits=warm)
        got = dout.to_numpy(np.uint8)
        assert np.array_equal(got, want), "delay %d" % delay
        assert np.all(got[delay:] <= 1) and nfb == 0
        # a second call on the same decoder objects reuses their seam buffers
        assert pkg.stream_decode_split(decs, dsy, nbits, delay, dout, warm_bits=warm) == 0
        assert np.array_equal(dout.to_numpy(np.uint8), want)
        for x in decs:
            x.close()


@pytest.mark.parametrize("tail", [0, 2], ids=["size_4k_multiple", "size_4k_plus_2"])
def test_symbols_ending_at_the_last_byte_of_an_allocation(pkg, tail):
    """the 15-step kernel fetches its 30 symbols as aligned dwords; the last pass of a buffer that ends exactly at its
    allocation's last byte must read nothing it may not and decode the same (buffer sizes = k*4096 and k*4096 + 2:
    last pass misaligned by 2 and by 0)."""
    nbits, delay = 4 * 1020, 200
    size = 3 * 4096 + tail
    syms, _ = orc.gen_coded_stream(9420 + tail, nbits, 3.0, 24.0, 5)
    d = pkg.Viterbi224(delay + 2 * 1020)
    d.init(0)
    want = d.stream_decode(syms, delay)
    buf = pkg.DeviceBuffer(size)
    off = size - 2 * nbits
    host = np.zeros(size, np.uint8)
    host[off:] = syms
    assert pkg.v224_lib().v224hip_h2d(buf.ptr, host.ctypes.data, size) == 0
    dout = pkg.DeviceBuffer(nbits)
    d.init(0)
    d.stream_decode_dev(buf, nbits, delay, dout, sym_offset=off)
    d.sync()
    assert np.array_equal(dout.to_numpy(np.uint8), want)
    d.init(0)
    d.update_dev(buf, nbits, byte_offset=off)
    o = orc.OracleV224(delay + 2 * 1020, orc.FAST)
    o.init(0)
    o.update(syms, nbits)
    assert d.chainback(1000, 0).tobytes() == o.chainback(1000, 0).tobytes()
    o.close()
    d.close()


@pytest.mark.parametrize("chunk,warms", [(1020, (3060, 4080)), (510, (1530, 1530))], ids=["chunk1020", "chunk510_as_the_chain"])
def test_stream_blocks_shared_between_two_decoders(pkg, chunk, warms):
    """v224hip_stream_decode_shared: a stream fed block by block, long blocks shared between the decoder that carries the
    stream and a second one that starts fresh inside the block (seam verified): byte for byte what ONE decoder gives for
    the same blocks -- ragged block sizes (not multiples of 15 or of the chunk), short blocks in between, the holder
    changing hands several times, a block of pure noise (seam may fail: redone), 0xff start-up marks."""
    delay = 200
    nbits = 150_000
    syms, _ = orc.gen_coded_stream(9700, nbits, 2.5, 24.0, 3)
    rng = np.random.default_rng(4)
    syms[2 * 70_000:2 * 88_000] = rng.integers(0, 256, 36_000, dtype=np.uint8)        # 18 000 bits of noise
    one = pkg.Viterbi224(delay + 2 * 1020)
    one.init(0)
    decs = [pkg.Viterbi224(delay + 2 * 1020) for _ in range(2)]
    for d in decs + [one]:
        d.set_option("chunk", chunk)
    decs[0].init(0)
    holder, pos, holders = 0, 0, []
    for n in (511, 13_001, 1, 25_000, 40, 12_240, 30_000, 19_999, 9_000, 3_384, 36_824):
        blk = syms[2 * pos:2 * (pos + n)]
        want = one.stream_decode(blk, delay)
        got, holder = pkg.stream_decode_shared(decs, holder, blk, delay, warms[0] if pos < 60_000 else warms[1])
        assert np.array_equal(got, want), "block at bit %d (%d bits)" % (pos, n)
        holders.append(holder)
        pos += n
    assert pos == nbits and len(set(holders)) == 2              # the stream really changed hands
    # afterwards the holder continues through the plain API like any decoder
    tail, _ = orc.gen_coded_stream(9701, 600, 3.0, 24.0, 0)
    assert np.array_equal(decs[holder].stream_decode(tail, delay), one.stream_decode(tail, delay))
    for d in decs + [one]:
        d.close()


def _progressive(pkg, decs, syms, nbits, expected, delay, pieces, warm=3060):
    p = pkg.ProgressiveDecode(decs, expected, delay, warm)
    pos = 0
    for n in pieces:
        n = min(n, nbits - pos)
        if n <= 0:
            break
        p.feed(syms[2 * pos:2 * (pos + n)])
        pos += n
    while pos < nbits:
        n = min(4096, nbits - pos)
        p.feed(syms[2 * pos:2 * (pos + n)])
        pos += n
    return p.end()


@pytest.mark.parametrize("case", ["as_expected", "shorter_than_the_cut", "just_past_the_cut", "longer_than_expected",
                                  "too_short_to_split", "one_decoder", "noise", "forced_redo"])
def test_progressive_two_decoder_stream_equals_one_decoder(pkg, case, monkeypatch):
    """v224hip_progressive_*: a stream fed piece by piece while two decoders work on it (cut planned for an EXPECTED length,
    one warm-up, seam verified) == v224hip_stream_decode of the whole stream on one decoder, byte for byte: ragged pieces,
    a stream that ends before / just after / far beyond the planned cut, a plan too short to split, a single decoder,
    pure noise across the seam, and the redo path forced."""
    delay, chunk = 200, 1020
    nbits, expected = {"as_expected": (60_001, 60_000), "shorter_than_the_cut": (25_003, 60_000),
                       "just_past_the_cut": (31_700, 60_000), "longer_than_expected": (90_010, 60_000),
                       "too_short_to_split": (7_000, 7_000), "one_decoder": (40_000, 40_000),
                       "noise": (50_000, 50_000), "forced_redo": (50_000, 50_000)}[case]
    syms, _ = orc.gen_coded_stream(9800, nbits, 2.5, 24.0, 2)
    if case == "noise":
        syms[2 * 15_000:2 * 45_000] = np.random.default_rng(5).integers(0, 256, 60_000, dtype=np.uint8)
    if case == "forced_redo":
        monkeypatch.setenv("V224HIP_SPLIT_FORCE_FALLBACK", "1")
    one = pkg.Viterbi224(delay + 2 * chunk)
    decs = [pkg.Viterbi224(delay + 2 * chunk) for _ in range(1 if case == "one_decoder" else 2)]
    for d in decs + [one]:
        d.set_option("chunk", chunk)
    one.init(0)
    want = one.stream_decode(syms, delay)
    rng = np.random.default_rng(6)
    pieces = [int(x) for x in rng.integers(1, 3000, 200)] + [1, 2, 3]
    got, redone = _progressive(pkg, decs, syms, nbits, expected, delay, pieces)
    assert len(got) == nbits and np.array_equal(got, want)
    got_min, redone_min = _progressive(pkg, decs, syms, nbits, expected, delay, pieces, warm=1)      # raised to 2 chunks
    assert np.array_equal(got_min, want) and (case != "as_expected" or redone_min == 0)
    assert redone == (1 if case == "forced_redo" else redone)
    if case in ("as_expected", "longer_than_expected", "just_past_the_cut"):
        assert redone == 0
    # the decoders are ordinary decoders again afterwards, and a second progressive run on them works
    got2, _ = _progressive(pkg, decs, syms, min(nbits, 20_000), 20_000, delay, [777] * 100)
    assert np.array_equal(got2, want[:min(nbits, 20_000)])
    for d in decs + [one]:
        d.close()


def test_progressive_api_refuses_bad_arguments(pkg):
    """Error behaviour of v224hip_progressive_*: NULL handle / decoders, mismatched chunk sizes, a ring too short for the
    delay, an output buffer too small -- -1 / NULL with a message, nothing left behind, decoders usable afterwards."""
    import ctypes as C
    L = pkg.v224_lib()
    u8p = C.POINTER(C.c_uint8)
    a, b = pkg.Viterbi224(200 + 2040), pkg.Viterbi224(200 + 2040)
    b.set_option("chunk", 1020)                                    # a keeps 2040
    hs = (C.c_void_p * 2)(a.h, b.h)
    assert not L.v224hip_progressive_begin(hs, 2, 50_000, 200, 3060) and b"chunk" in L.v224hip_last_error()
    assert not L.v224hip_progressive_begin(None, 2, 50_000, 200, 3060)
    assert not L.v224hip_progressive_begin(hs, 1, 50_000, 5000, 3060) and b"short ring" in L.v224hip_last_error()
    assert L.v224hip_progressive_feed(None, None, 0) == -1
    assert L.v224hip_progressive_end(None, None, 0, None, None) == -1
    L.v224hip_progressive_abort(None)
    b.set_option("chunk", 2040)
    syms, _ = orc.gen_coded_stream(9850, 9000, 3.0, 24.0, 0)
    h = L.v224hip_progressive_begin(hs, 2, 9000, 200, 3060)
    assert h
    assert L.v224hip_progressive_feed(h, syms.ctypes.data_as(u8p), 9000) == 0
    small = np.zeros(100, dtype=np.uint8)
    n, r = C.c_longlong(7), C.c_int(7)
    assert L.v224hip_progressive_end(h, small.ctypes.data_as(u8p), 100, C.byref(n), C.byref(r)) == -1      # handle is gone now
    assert n.value == 0 and b"too small" in L.v224hip_last_error()
    a.init(0)
    want = a.stream_decode(syms, 200)
    got, _ = _progressive(pkg, [a, b], syms, 9000, 9000, 200, [4000, 5000])
    assert np.array_equal(got, want)
    a.close(); b.close()


@pytest.mark.slow
def test_deep_stream_pinned_to_the_oracle_at_bench_size(pkg):
    """BASELINE configs[1] / BASELINE.md B3: the bench's own 10^7-symbol stream, checked against the ORACLE deep inside,
    not only against itself.  The u16 path metrics are renormalised every second launch and the running offset is
    carried through 333 k launches; a defect that needs millions of steps to show would pass every shorter test.  At
    bits ~2.5*10^6 and ~4.9*10^6 the decoder's path metrics are exported (v224hip_export_metrics), the oracle (port
    semantics, viterbi224_port.c:159-195) is seeded with them (orc_v224_set_metrics: the recursion has no other state)
    and runs the next 20 000 trellis steps with decodebit(200, 0) per bit: the product's bits over that stretch, the
    decision rows of its last pass there and all 2^23 relative metrics at its end must equal the oracle's.  The run
    with the exports must also equal a plain run of the whole stream (exporting switches the metric order and back)."""
    from importlib import import_module
    synth = import_module("isee3_decoder_amd.synth")
    nbits, delay, slab, span = 5_000_000, 200, 8160, 20_400       # span = 10 chunks of 2 040 = whole 15-step passes
    syms, _, _ = synth.coded_stream(1000, nbits, 3.0, 24.0, 1.0)
    dsy = pkg.DeviceBuffer.from_numpy(syms)
    d = pkg.Viterbi224(delay + 2040)
    plain = pkg.DeviceBuffer(nbits)
    d.init(0)
    for pos in range(0, nbits, slab):
        d.stream_decode_dev(dsy, min(slab, nbits - pos), delay, plain, sym_offset=2 * pos, out_offset=pos)
    d.sync()
    want_all = plain.to_numpy(np.uint8).copy()

    marks = [2_500_000 // slab * slab, 4_900_000 // slab * slab]
    stops = sorted(set(marks + [m + span for m in marks]))
    out = pkg.DeviceBuffer(nbits)
    snap, last_rows = {}, {}
    d.init(0)
    pos = 0
    while pos < nbits:
        nxt = min([s for s in stops if s > pos] + [nbits, pos + slab])
        d.stream_decode_dev(dsy, nxt - pos, delay, out, sym_offset=2 * pos, out_offset=pos)
        pos = nxt
        if pos in stops:
            if pos - span in marks:                    # end of a pinned stretch: rows first (the export below may not move them)
                dp = d.get_counter("dp")
                last_rows[pos] = [(j, _fnv(d.export_row((dp - 1 - j) % d.length))) for j in (0, 1, 7, 14, 15, 29)]
            snap[pos] = d.export_metrics()
            assert d.get_counter("steps") == pos
    d.sync()
    assert np.array_equal(out.to_numpy(np.uint8), want_all), "exporting metrics mid-stream changed the decode"

    for m in marks:
        o = orc.OracleV224(delay + 1, orc.FAST)
        o.set_metrics(snap[m])
        got = want_all[m:m + span]
        rows_o = {}
        keep = {span - 1 - j for j, _ in last_rows[m + span]}
        for u in range(span):
            o.update(syms[2 * (m + u):2 * (m + u) + 2], 1)
            if u in keep:
                rows_o[u] = o.row_hash((o.dp() - 1) % (delay + 1))
            if u >= delay:                             # the traceback stays inside the rows written since the seed
                assert o.decodebit(delay, 0) == got[u], "bit %d (%d after the seed at %d) differs from the oracle" % (m + u, u, m)
        for j, h in last_rows[m + span]:
            assert rows_o[span - 1 - j] == h, "decision row %d steps before bit %d differs from the oracle" % (j, m + span)
        assert np.array_equal(o.get_metrics(), snap[m + span]), "path metrics at bit %d differ from the oracle" % (m + span)
        o.close()
    d.close()
