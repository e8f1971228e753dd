"""How many trellis steps until a freshly initialised decoder's path metrics equal (up to a constant) those of a decoder
that has followed the stream from its start?  That is what the seam check of the split / shared decode verifies; the
answer sizes the warm-up.  For each Eb/N0: fraction of states that still differ after w steps."""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from conftest import load_pkg
import orc
pkg = load_pkg()
T0 = 6000
for ebn0, noise in ((3.0, 0), (1.5, 0), (3.0, 10), (0.0, 0), (3.0, 100)):
    syms, _ = orc.gen_coded_stream(31, T0 + 4096, ebn0, 24.0, noise)
    a = pkg.Viterbi224(64)
    a.init(0)
    a.update(syms[:2 * T0], T0)
    res = []
    prev = 0
    for w in (120, 255, 390, 510, 765, 1020, 1530, 2040, 3060, 4080):
        a.update(syms[2 * (T0 + prev):2 * (T0 + w)], w - prev)      # A stands at T0 + w
        prev = w
        b = pkg.Viterbi224(64)
        b.init(0)
        b.update(syms[2 * T0:2 * (T0 + w)], w)                      # B started fresh at T0
        ma, mb = a.export_metrics(), b.export_metrics()
        res.append((w, int(np.count_nonzero(ma != mb))))
        b.close()
    a.close()
    print("Eb/N0 %.1f dB, %d %% noise blocks: states differing after w steps:" % (ebn0, noise), res, flush=True)
