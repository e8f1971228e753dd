import sys, time, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from conftest import load_pkg
pkg = load_pkg()
from importlib import import_module
synth = import_module("isee3_decoder_amd.synth")
nbits = 100_000
syms, _, _ = synth.coded_stream(5, nbits, 3.0, 24.0, 0.0)
dsy, dout = pkg.DeviceBuffer.from_numpy(syms), pkg.DeviceBuffer(nbits)
a = pkg.Viterbi224(200 + 2 * 1020)
a.init(0); a.stream_decode_dev(dsy, nbits, 200, dout); a.sync()
t0 = time.perf_counter(); a.init(0); a.stream_decode_dev(dsy, nbits, 200, dout); a.sync()
print("gap %s: single %.3f Msym/s" % (os.environ.get("V224HIP_TEST_GAP"), 2 * nbits / (time.perf_counter() - t0) / 1e6), flush=True)
