"""Verified split on two decoders with decoder 0 at the top wave priority (option "prio" = argv[1]): total rate and each
decoder's launch period while both run."""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from conftest import load_pkg
from importlib import import_module
pkg = load_pkg()
synth = import_module("isee3_decoder_amd.synth")
prio = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nbits = 2_000_000
syms, bits, _ = synth.coded_stream(1000, nbits, 3.0, 24.0, 1.0)
dsy, dout = pkg.DeviceBuffer.from_numpy(syms), pkg.DeviceBuffer(nbits)
decs = [pkg.Viterbi224(200 + 2040) for _ in range(2)]
if prio:      # option "prio" (every wave of the decoder's launches at s_setprio 3) existed only while this was measured: commit history
    decs[0].set_option("prio", prio)
ref = None
for rep in range(4):
    for d in decs:
        d.set_option("profile", 4); d.acs_stats(reset=True)
    t0 = time.perf_counter()
    pkg.stream_decode_split(decs, dsy, nbits, 200, dout, 14280)
    dt = time.perf_counter() - t0
    us = []
    for d in decs:
        l, ms, st = d.acs_stats(); d.set_option("profile", 0); us.append(ms / max(l, 1) * 1e3)
    out = dout.to_numpy(np.uint8)
    if ref is None: ref = out.copy()
    print("prio %d: split %.3f Msym/s; launch period decoder 0 %.2f us, decoder 1 %.2f us (whole part, incl. what it ran alone); same bits %s"
          % (prio, 2 * nbits / dt / 1e6, us[0], us[1], np.array_equal(out, ref)), flush=True)
