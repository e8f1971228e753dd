"""Synthetic inputs for bench.py and the CLI demos (numpy only; independent of oracle/).

Code constants: reference code.h:54-63 (MCQLI24: POLY1 = 073665667, POLY2 = 073665665, K = 24,
second symbol inverted).  Encoder semantics: reference encode.c:17-35 (shift left, MSB-first data,
symbol 0 from POLY1, symbol 1 from POLY2)."""
import numpy as np

POLY1 = 0o73665667
POLY2 = 0o73665665
K = 24


def encode_bits(bits):
    """bits: uint8 0/1 array -> interleaved hard symbols (2 per bit), encoder starting at state 0."""
    bits = np.asarray(bits, dtype=np.uint8)
    n = len(bits)
    pad = np.concatenate([np.zeros(K - 1, np.uint8), bits])
    y = []
    for poly, flip in ((POLY1, 0), (POLY2, 1)):
        acc = np.full(n, flip, dtype=np.uint8)
        for k in range(K):
            if (poly >> k) & 1:                   # tap k looks k bits into the past
                acc ^= pad[K - 1 - k: K - 1 - k + n]
        y.append(acc)
    out = np.empty(2 * n, dtype=np.uint8)
    out[0::2], out[1::2] = y[0], y[1]
    return out


def coded_stream(seed, nbits, ebn0_db=3.0, amplitude=24.0, noise_block_pct=1.0):
    """8-bit offset-128 soft symbols of a continuously encoded random bit stream through AWGN
    (noise level as reference vtest224.c:93-95); `noise_block_pct` % of 1024-symbol blocks are
    replaced by pure noise.  Returns (symbols uint8[2*nbits], bits uint8[nbits])."""
    rng = np.random.default_rng(seed)
    bits = rng.integers(0, 2, nbits, dtype=np.uint8)
    hard = encode_bits(bits).astype(np.float32)
    esn0 = ebn0_db + 10 * np.log10(0.5)
    sigma = amplitude * np.sqrt(0.5) / 10 ** (0.05 * esn0)
    sig = amplitude * (2 * hard - 1)
    nblk = (2 * nbits + 1023) // 1024
    noisy = rng.random(nblk) < noise_block_pct / 100.0
    mask = np.repeat(noisy, 1024)[: 2 * nbits]
    sig[mask] = 0
    x = 128 + sig + rng.normal(0, sigma, 2 * nbits).astype(np.float32)
    return np.clip(np.rint(x), 0, 255).astype(np.uint8), bits, mask


SYNCWORD = 0x12fc819fbe      # reference framer.c:18 / decode.c:24
ACTUALCLOCK = 1024.545058    # reference symdemod.c:18


def telemetry_bits(seed, nbits):
    """1024-bit frames whose last 40 bits are the sync word (reference framer.c:12-18,69)."""
    rng = np.random.default_rng(seed)
    bits = rng.integers(0, 2, nbits, dtype=np.uint8)
    sync = np.array([(SYNCWORD >> (39 - i)) & 1 for i in range(40)], dtype=np.uint8)
    for f in range(0, nbits - 1023, 1024):
        bits[f + 984:f + 1024] = sync
    return bits


def _iq_range(sy, s, e, samprate, fc_hz, beta, symrate, amp, sigma, phi0, rng, out):
    """samples [s, e) of the capture into out[0 : 2 (e - s)] (signal = a function of the absolute sample index)"""
    i = np.arange(s, e, dtype=np.float64)
    ph = i * (symrate / samprate)
    k = ph.astype(np.int64)
    m = np.where(ph - k < 0.5, -1.0, 1.0) * sy[k]
    th = np.mod(i * (2 * np.pi * fc_hz / samprate), 2 * np.pi) + phi0 + beta * m
    x = amp * np.cos(th) + rng.normal(0, sigma, e - s)
    y = amp * np.sin(th) + rng.normal(0, sigma, e - s)
    out[0:2 * (e - s):2] = np.clip(np.rint(x), -32767, 32767).astype(np.int16)
    out[1:2 * (e - s):2] = np.clip(np.rint(y), -32767, 32767).astype(np.int16)


def iq_capture(seed, samprate, seconds, fc_hz=12345.678, beta=1.1, symrate=ACTUALCLOCK, amp=3000.0,
               cn0_dbhz=45.0, chunk=1 << 22):
    """Synthetic PM capture: int16 interleaved I,Q of
    A*exp(j(2 pi fc t + phi0 + beta*m(t))) + complex AWGN, m(t) = Manchester of the r=1/2 encoded
    telemetry (first half -, second half + for symbol 1: reference symdemod.c:227-235); modulation
    index 1.1 rad (pmdemod.c:83).  A = 3000 keeps 5 sigma of the 45 dB-Hz noise inside int16 at
    250 kS/s; amp=None picks A so that the per-component noise sigma is 6000 at any sample rate.
    Returns (iq int16[2n], sent bits)."""
    if amp is None:
        amp = 6000.0 / np.sqrt(samprate / (2.0 * 10 ** (cn0_dbhz / 10.0)))
    n = int(samprate * seconds)
    nsym = int(seconds * symrate) + 4
    bits = telemetry_bits(seed, nsym // 2 + 2)
    sy = encode_bits(bits)[:nsym].astype(np.int8) * 2 - 1
    rng = np.random.default_rng(seed + 7)
    phi0 = rng.random() * 2 * np.pi
    sigma = amp * np.sqrt(samprate / (2.0 * 10 ** (cn0_dbhz / 10.0)))
    out = np.empty(2 * n, dtype=np.int16)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        _iq_range(sy, s, e, samprate, fc_hz, beta, symrate, amp, sigma, phi0, rng, out[2 * s:2 * e])
    return out, bits


_SHARED_OUT = None          # the capture being filled (anonymous shared mapping, inherited by the forked workers)


def _iq_piece(args):
    seed, j, s, e, sy, samprate, fc_hz, beta, symrate, amp, sigma, phi0 = args
    _iq_range(sy, s, e, samprate, fc_hz, beta, symrate, amp, sigma, phi0, np.random.default_rng([seed, j]),
              _SHARED_OUT[2 * s:2 * e])
    return j


def iq_capture_parallel(seed, samprate, nsamples, workers=8, piece=1 << 22, fc_hz=12345.678, beta=1.1, symrate=ACTUALCLOCK,
                        cn0_dbhz=45.0):
    """The same signal model for LONG captures (configs[4]: 64 blocks of 2^23 samples at 10 MS/s = 2.1 GB): carrier and
    telemetry are functions of the absolute sample index, the noise of piece j comes from its own generator
    default_rng([seed, j]), so the pieces can be made by `workers` forked processes in any order -- they write straight
    into one shared anonymous mapping -- and the capture does not depend on how many there are.  amp as
    iq_capture(amp=None).  Returns (iq int16[2 nsamples], sent bits)."""
    global _SHARED_OUT
    import mmap
    amp = 6000.0 / np.sqrt(samprate / (2.0 * 10 ** (cn0_dbhz / 10.0)))
    seconds = nsamples / samprate
    nsym = int(seconds * symrate) + 4
    bits = telemetry_bits(seed, nsym // 2 + 2)
    sy = encode_bits(bits)[:nsym].astype(np.int8) * 2 - 1
    phi0 = np.random.default_rng(seed + 7).random() * 2 * np.pi
    sigma = amp * np.sqrt(samprate / (2.0 * 10 ** (cn0_dbhz / 10.0)))
    jobs = [(seed, j, s, min(nsamples, s + piece), sy, samprate, fc_hz, beta, symrate, amp, sigma, phi0)
            for j, s in enumerate(range(0, nsamples, piece))]
    buf = mmap.mmap(-1, 4 * nsamples)                  # MAP_SHARED | MAP_ANONYMOUS
    _SHARED_OUT = np.frombuffer(buf, dtype=np.int16)
    try:
        if workers <= 1:
            for job in jobs:
                _iq_piece(job)
        else:
            import multiprocessing as mp
            with mp.get_context("fork").Pool(workers) as pool:
                for _ in pool.imap_unordered(_iq_piece, jobs, chunksize=1):
                    pass
        out = _SHARED_OUT
    finally:
        _SHARED_OUT = None
    return out, bits
