"""The verified split on a decoder pair whose streams sit on later hardware queues: K dummy decoders (each owns a stream) are
created first.  K = argv[1]."""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from conftest import load_pkg
from importlib import import_module
pkg = load_pkg()
synth = import_module("isee3_decoder_amd.synth")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nbits = 1_500_000
syms, bits, _ = synth.coded_stream(1000, nbits, 3.0, 24.0, 1.0)
dsy, dout = pkg.DeviceBuffer.from_numpy(syms), pkg.DeviceBuffer(nbits)
dummies = [pkg.Viterbi224(32) for _ in range(K)]
decs = [pkg.Viterbi224(200 + 2040) for _ in range(2)]
rates = []
for rep in range(4):
    t0 = time.perf_counter()
    pkg.stream_decode_split(decs, dsy, nbits, 200, dout, 14280)
    rates.append(2 * nbits / (time.perf_counter() - t0) / 1e6)
decs[0].init(0)
t0 = time.perf_counter(); decs[0].stream_decode_dev(dsy, nbits, 200, dout); decs[0].sync(); lone = 2 * nbits / (time.perf_counter() - t0) / 1e6
print("%d stream(s) created before the pair: split %s Msym/s, decoder 0 alone %.3f" % (K, " ".join("%.3f" % r for r in rates), lone), flush=True)
