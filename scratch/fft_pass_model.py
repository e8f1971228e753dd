"""numpy model of k_fft_pass (csrc/dsp_hip.hip): the LDS-staged Stockham stage of radix R = R1 * R2, checked against
numpy.fft for the pass plans pmd_fft_peak uses.  Index algebra only (run on the CPU before the first GPU run)."""
import numpy as np

SHAPE = {5: (8, 4), 6: (8, 8), 7: (16, 8), 8: (16, 16)}


def fft_pass(x, N, s, lg):
    R1, R2 = SHAPE[lg]
    R = R1 * R2
    stride = N // R
    y = np.zeros(N, complex)
    for t in range(N // R):
        Z = np.zeros((R2, R1), complex)
        for a in range(R2):
            v = np.array([x[t + (a + R2 * b) * stride] for b in range(R1)])
            Y = np.fft.fft(v)                                  # R1-point DFT over b
            for k1 in range(R1):
                Z[a, k1] = Y[k1] * np.exp(-2j * np.pi * (a * k1) / R)
        q = t & (s - 1)
        ps = t - q
        for k1 in range(R1):
            X = np.fft.fft(Z[:, k1])                           # R2-point DFT over a
            for k2 in range(R2):
                k = k1 + R1 * k2
                y[q + R * ps + k * s] = X[k2] * np.exp(-2j * np.pi * (ps * k) / N)
    return y


def plan(logN):
    npass = (logN + 7) // 8
    base, extra = divmod(logN, npass)
    return [base + (1 if i < extra else 0) for i in range(npass)]


rng = np.random.default_rng(1)
for logN in (12, 13, 15, 16, 17):
    N = 1 << logN
    x = rng.normal(size=N) + 1j * rng.normal(size=N)
    cur, s = x, 1
    for lg in plan(logN):
        cur = fft_pass(cur, N, s, lg)
        s <<= lg
    err = np.max(np.abs(cur - np.fft.fft(x))) / np.max(np.abs(np.fft.fft(x)))
    print(logN, plan(logN), "rel err %.2e" % err)
    assert err < 1e-12
