"""pair rate of two decoders as a function of how many other streams the process holds (kept alive, idle)"""
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
dummies = [pkg.SymDemodEngine(4096) for _ in range(k)]          # one stream each
a = pkg.Viterbi224(200 + 2 * 1020)
b = pkg.Viterbi224(200 + 2 * 1020)
pkg.stream_decode_split([a, b], dsy, nbits, 200, dout, 4080)
t0 = time.perf_counter(); pkg.stream_decode_split([a, b], dsy, nbits, 200, dout, 4080); a.sync(); b.sync()
r = 2 * nbits / (time.perf_counter() - t0) / 1e6
print("dummy streams %d (prio %s, max queues %s): pair %.3f Msym/s" % (k, "normal" if os.environ.get("ISEE3DSP_NORMAL_PRIORITY") else "high", os.environ.get("GPU_MAX_HW_QUEUES", "8 (lib)"), r), flush=True)
