"""Framed chainback in pieces: how often a piece has to be walked again, and what a frame costs, against the warm-up
(V224HIP_CB_WARM) -- coded 1024-bit frames at several Eb/N0."""
import os, sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import orc
from conftest import load_pkg
pkg = load_pkg()
nb = 1024
d = pkg.Viterbi224(nb)
for ebn0 in (1.0, 2.0, 3.0, 4.0):
    red, t = 0, 0.0
    for f in range(12):
        syms, _ = orc.gen_coded_frame(5000 + f, nb, ebn0, 24.0)
        d.init(0); d.update(syms, nb); d.sync()
        t0 = time.perf_counter(); d.chainback(nb, 0); t += time.perf_counter() - t0
        red += d.get_counter("chainback_redone")
    print("warm %s: Eb/N0 %.1f dB: %d of %d pieces walked again, chainback call %.1f us" % (os.environ.get("V224HIP_CB_WARM", "192"), ebn0, red, 12 * 15, t / 12 * 1e6), flush=True)
