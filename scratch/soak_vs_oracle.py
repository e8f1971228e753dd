"""One-off evidence run (not part of tests/): the first N bits of bench.py's own 10^7-symbol input, decoded on the GPU
(one decoder, and two decoders with the verified split), against the CPU oracle's vdecode-style loop, bit for bit.
    OMP_NUM_THREADS=16 python scratch/soak_vs_oracle.py 1000000
"""
import os, sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
os.environ.setdefault("OMP_NUM_THREADS", "16")
import numpy as np
import orc
from conftest import load_pkg
from importlib import import_module
pkg = load_pkg()
synth = import_module("isee3_decoder_amd.synth")
nbits = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 600.0
delay = 200
syms, bits, _ = synth.coded_stream(1000, 5_000_000, 3.0, 24.0, 1.0)
syms = syms[:2 * nbits]
d = pkg.Viterbi224(delay + 2 * 1020); d.init(0)
t0 = time.perf_counter(); one = d.stream_decode(syms, delay); t1 = time.perf_counter()
decs = [d, pkg.Viterbi224(delay + 2 * 1020)]
dsy, dout = pkg.DeviceBuffer.from_numpy(syms), pkg.DeviceBuffer(nbits)
redone = pkg.stream_decode_split(decs, dsy, nbits, delay, dout)
two = dout.to_numpy(np.uint8)
print("GPU: %d bits, one decoder %.2f s; split: parts redone %d, identical to one decoder: %s" % (nbits, t1 - t0, redone, np.array_equal(one, two)), flush=True)
o = orc.OracleV224(delay + 1, orc.FAST); o.init(0)
t0 = time.perf_counter(); bad = 0; done = 0
for u in range(nbits):
    o.update(syms[2 * u:2 * u + 2], 1)
    w = o.decodebit(delay, 0) if u + 1 >= delay else 0xff
    if w != one[u]:
        bad += 1
    done = u + 1
    if done % 50000 == 0:
        el = time.perf_counter() - t0
        print("oracle %d bits, %.0f s, mismatches %d" % (done, el, bad), flush=True)
        if el > budget:
            break
print("RESULT: %d bits compared against the CPU oracle, %d mismatches" % (done, bad), flush=True)
