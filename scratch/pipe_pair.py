import sys, time, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from conftest import load_pkg
pkg = load_pkg()
from importlib import import_module
synth = import_module("isee3_decoder_amd.synth")
nbits = 200_000
syms, _, _ = synth.coded_stream(5, nbits, 3.0, 24.0, 0.0)
dsy, dout = pkg.DeviceBuffer.from_numpy(syms), pkg.DeviceBuffer(nbits)
k = int(sys.argv[1])
a = pkg.Viterbi224(200 + 2 * 1020)
dummies = [pkg.SymDemodEngine(4096) for _ in range(k)]
b = pkg.Viterbi224(200 + 2 * 1020)
pkg.stream_decode_split([a, b], dsy, nbits, 200, dout, 4080)
t0 = time.perf_counter(); pkg.stream_decode_split([a, b], dsy, nbits, 200, dout, 4080); a.sync(); b.sync()
print("own traceback streams, %d queues between the decoders: pair %.3f Msym/s" % (k, 2 * nbits / (time.perf_counter() - t0) / 1e6), flush=True)
