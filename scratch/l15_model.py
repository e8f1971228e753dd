"""numpy model of the 15-step tile pass (engine LDS15): checks the index algebra -- tile layout,
in-place butterfly network, branch-parity masks, thread/register mapping and the decision-bit
addressing -- against the CPU oracle, before any of it is trusted on the GPU.

    python scratch/l15_model.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import orc  # noqa: E402

K, XB = 15, 8
P1H = 0o73665667 >> 1
NS = 1 << 23


def parity(a):
    a = a.copy()
    for s in (16, 8, 4, 2, 1):
        a ^= a >> s
    return a & 1


def qpos(j, t):
    return j - (K - t) if j >= K - t else j + t + XB


def state_of(q, x, t):
    """state number of slot q, tile x after t steps"""
    s = x.astype(np.int64) << t
    for j in range(K):
        s = s | (((q >> j) & 1).astype(np.int64) << qpos(j, t))
    return s


# ---- thread / register mapping per level ----------------------------------------------------
# level L holds steps 4L..4L+3 (level 3: 12..14).  Returns slot for (thread th, register a, half h).
def slot_of(level, th, a, h):
    if level == 0:      # reg <-> slot 14..11, pack = slot 0, th <-> slot 10..1
        return (a << 11) | (th << 1) | h
    if level == 1:      # reg <-> slot 10..7, th[9:6] <-> slot 14..11, th[5:0] <-> slot 6..1
        return ((th >> 6) << 11) | (a << 7) | ((th & 63) << 1) | h
    if level == 2:      # reg <-> slot 6..3, th[9:2] <-> slot 14..7, th[1:0] <-> slot 2..1
        return ((th >> 2) << 7) | (a << 3) | ((th & 3) << 1) | h
    # level 3 (direct output rows): reg a = b9:b2:b1:b0, pack = slot 8; th[9:6]=slot 6..3 (wave), th[5:2]=slot 14..11,
    # th[1]=slot 10, th[0]=slot 7
    b9, low3 = a >> 3, a & 7
    return (((th >> 2) & 15) << 11) | (((th >> 1) & 1) << 10) | (b9 << 9) | (h << 8) | ((th & 1) << 7) | (((th >> 6) & 15) << 3) | low3


def reg_pair_bit(level, t):
    """register-index bit paired at step t"""
    return (3 - (t % 4)) if level < 3 else (2 - (t - 12))


def locate(level, q):
    """inverse of slot_of: (th, a, h) for slot q"""
    if level == 0:
        return (q >> 1) & 1023, q >> 11, q & 1
    if level == 1:
        return ((q >> 11) << 6) | ((q >> 1) & 63), (q >> 7) & 15, q & 1
    if level == 2:
        return ((q >> 7) << 2) | ((q >> 1) & 3), (q >> 3) & 15, q & 1
    th = (((q >> 3) & 15) << 6) | ((q >> 11) << 2) | (((q >> 10) & 1) << 1) | ((q >> 7) & 1)
    a = (((q >> 9) & 1) << 3) | (q & 7)
    return th, a, (q >> 8) & 1


def decision_bit(level, t, a, h):
    """bit position inside the thread's stage dword (perm/bfi packing: byte 2*which + half, bit = butterfly)"""
    lb = reg_pair_bit(level, t)
    bf = ((a >> (lb + 1)) << lb) | (a & ((1 << lb) - 1))
    which = (a >> lb) & 1
    return 8 * (2 * which + h) + bf


def get_decision_addr(tau, s):
    """what the device's lds15_get_decision computes: (word index in the row, bit)"""
    t = tau - 1
    x = (s >> tau) & 0xff
    q = ((s & ((1 << tau) - 1)) << (K - tau)) | (s >> (tau + XB))
    level = min(t // 4, 3)
    th, a, h = locate(level, q)
    return x * 1024 + th, decision_bit(level, t, a, h)


def main():
    nsteps = 15
    rng = np.random.default_rng(7)
    syms = rng.integers(0, 256, 2 * (nsteps + 10), dtype=np.uint8)
    # run a few warm-up steps in the oracle so that metrics are not uniform
    warm = 10
    o = orc.OracleV224(nsteps + warm + 1, orc.LITERAL)
    o.init(5)
    o.update(syms[:2 * warm], warm)
    # fetching all oracle metrics one by one is slow: re-run a numpy natural-order trellis instead
    m = np.full(NS, 1000, np.int64)
    m[5] = 0
    i = np.arange(NS // 2, dtype=np.int64)
    par_i = parity(i & P1H)

    def nat_step(m, s0, s1):
        c = np.where(i & 1, s0 + s1, s0 + 255 - s1)
        bm = np.where(par_i == 1, 510 - c, c)
        A = m[:NS // 2] + bm; B = m[NS // 2:] + 510 - bm
        C = m[:NS // 2] + 510 - bm; D = m[NS // 2:] + bm
        new = np.empty(NS, np.int64); dec = np.empty(NS, np.uint8)
        new[0::2] = np.minimum(A, B); new[1::2] = np.minimum(C, D)
        dec[0::2] = A >= B; dec[1::2] = C >= D
        return new, dec

    for w in range(warm):
        m, dec = nat_step(m, int(syms[2 * w]), int(syms[2 * w + 1]))
    # natural-order model agrees with the oracle on the warm-up rows?
    rowbits = np.unpackbits(o.row(warm - 1), bitorder="little")
    assert np.array_equal(rowbits, dec), "natural model != oracle (formula error)"
    print("natural-order formula matches the oracle row", warm - 1)

    # ---- tile pass ----------------------------------------------------------------------------
    x = np.repeat(np.arange(256, dtype=np.int64), 1 << K)
    q = np.tile(np.arange(1 << K, dtype=np.int64), 256)
    # L15 layout: phys = (s & 0xff) << 15 | s >> 8 ; tile x holds states s = q << 8 | x
    T = m[(q << 8) | x].copy()                        # T[x*32768 + q]
    mnat = m.copy()
    rows_model = []
    for t in range(nsteps):
        s0, s1 = int(syms[2 * (warm + t)]), int(syms[2 * (warm + t) + 1])
        mnat, decnat = nat_step(mnat, s0, s1)
        pj = K - 1 - t
        lo = (q >> pj) & 1 == 0
        idx_lo = np.nonzero(lo)[0]
        idx_hi = idx_lo + (1 << pj)
        S = state_of(q[idx_lo], x[idx_lo], t)         # MSB clear by construction
        assert (S >> 22).max() == 0
        par = parity(S & P1H)
        c = np.where(S & 1, s0 + s1, s0 + 255 - s1)
        bm = np.where(par == 1, 510 - c, c)
        oi, oj = T[idx_lo], T[idx_hi]
        A = oi + bm; B = oj + 510 - bm; C = oi + 510 - bm; D = oj + bm
        T[idx_lo] = np.minimum(A, B); T[idx_hi] = np.minimum(C, D)
        dslot = np.empty(NS, np.uint8)
        dslot[idx_lo] = A >= B; dslot[idx_hi] = C >= D
        # the new states those slots now hold
        S1 = state_of(q, x, t + 1)
        chk = np.empty(NS, np.uint8); chk[S1] = dslot
        assert np.array_equal(chk, decnat), "tile network decisions differ at step %d" % t
        # ---- per-thread dwords exactly as the kernel packs them, then the device address formula
        level = min(t // 4, 3)
        th, a, h = locate(level, q)
        word = x * 1024 + th
        bit = np.array([decision_bit(level, t, int(aa), int(hh)) for aa in range(16) for hh in range(2)]).reshape(16, 2)[a, h]
        rowwords = np.zeros(NS // 32, np.uint32)
        np.bitwise_or.at(rowwords, word, (dslot.astype(np.uint32) << bit.astype(np.uint32)))
        # device lookup for a sample of states
        ss = rng.integers(0, NS, 200000)
        got = np.empty(len(ss), np.uint8)
        for n, s in enumerate(ss[:3000]):
            w, b = get_decision_addr(t + 1, int(s))
            got[n] = (rowwords[w] >> b) & 1
        assert np.array_equal(got[:3000], decnat[ss[:3000]]), "get_decision formula wrong at tau %d" % (t + 1)
        # the mapping is a bijection (every bit of every word used exactly once)
        cnt = np.zeros(NS, np.uint8)
        np.add.at(cnt, word * 32 + bit, 1)
        assert cnt.min() == 1 and cnt.max() == 1
        # static-ness of i0 (= S & 1) per level: must depend only on what the kernel assumes
        print("step %2d ok (level %d)" % (t, level))
    # output layout: tile x, slot q -> natural s' = x << 15 | q ; L15 phys = (q & 0xff) << 15 | x << 7 | q >> 8
    snew = (x << 15) | q
    assert np.array_equal(T, mnat[snew])
    phys = ((snew & 0xff) << 15) | (snew >> 8)
    assert np.array_equal(phys, ((q & 0xff) << 15) | (x << 7) | (q >> 8))
    print("final metrics match; output layout consistent")


if __name__ == "__main__":
    main()
