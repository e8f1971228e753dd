"""CPU: the sync vector of the icesync correlator (icesync.c:55-97; host code of the product, include/isee3_icesync.h)
against the oracle's restatement and against the reference's own constants, and the oracle's fft_sync_search
(icesync.c:139-208) against a direct numpy correlation.  PARITY UNPINNED for the search itself: icesync.c needs FFTW3
(icesync.c:17), which the image lacks -- the reference program cannot be built here."""
import numpy as np

import orc
from conftest import load_pkg

# vdecode.c:27-30 sync_vector[] = the expected sign of the last 34 encoded tail + sync symbols (+1 = symbol 1): the SAME 34
# symbols icesync.c:64-66 takes from the encoder (symbols 46..79 of encode({12 fc 81 9f be 00...}))
SYNC_SIGN = [-1, 1, 1, 1, 1, 1, 1, -1, 1, -1, 1, 1, 1, 1, -1, -1, 1,
             1, -1, -1, 1, 1, -1, 1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1]


def test_sync_vector_product_equals_oracle_and_reference_constants():
    pkg = load_pkg()
    for ss in (250000 / 1024.475, 25000 / 1024.475, 16.0, 9760.3):
        v = pkg.icesync_sync_vector(ss)
        assert np.array_equal(v, orc.icesync_sync_vector(ss))
        assert len(v) == int(34 * ss + 1)
        # Manchester: symbol 1 = first half -1, second half +1 (icesync.c:90-97); sample at 1/4 and 3/4 of every symbol
        for k in range(34):
            a, b = v[int((k + 0.25) * ss)], v[int((k + 0.75) * ss)]
            assert (a, b) == (-SYNC_SIGN[k], SYNC_SIGN[k]), k
        assert np.all(v[int(34 * ss) + 1:] == 0) if len(v) > int(34 * ss) + 1 else True


def test_oracle_search_equals_direct_correlation():
    ss = 25000 / 1024.475
    v = orc.icesync_sync_vector(ss)
    N = 1 << 16
    frames = ss * 2048
    n = int(np.ceil(frames))
    rng = np.random.default_rng(2)
    x = rng.normal(0, 300, n + 64).astype(np.int16)
    pos = 23456
    x[pos:pos + len(v)] += (400 * v).astype(np.int16)
    pk, mp, res = orc.icesync_search(v, N, x, frames, 0, n, True)
    xp = np.zeros(N); xp[:n] = x[:n]
    vp = np.zeros(N); vp[:len(v)] = v
    ref = np.real(np.fft.ifft(np.fft.fft(xp) * np.conj(np.fft.fft(vp)))) * N
    assert pk == pos == int(np.argmax(ref[:n]))
    assert np.max(np.abs(res - ref)) <= 1e-9 * ref.max() and abs(mp - ref[pos]) <= 1e-9 * ref.max()
    # direct definition at the peak: N * sum_m x[m + pos] * v[m] (FFTW's c2r is the unnormalised inverse)
    assert abs(mp - N * float(np.dot(x[pos:pos + len(v)].astype(np.float64), v))) <= 1e-9 * mp
    # window, failure and fold rules (icesync.c:188-206)
    assert orc.icesync_search(v, N, x, frames, pos + 1, n)[0] != pos
    assert orc.icesync_search(v, N, np.zeros(n + 64, np.int16), frames, 0, n)[0] == orc.ICESYNC_FAIL
    neg = -np.abs(x)
    neg[:len(v)] = (-300 * np.abs(v)).astype(np.int16)
    pk2, _ = orc.icesync_search(v, N, x, frames, N - 100, N)      # circular tail: indices above N/2 fold to N - index
    assert pk2 == orc.ICESYNC_FAIL or 0 < pk2 <= 100
