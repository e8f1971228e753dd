"""Host cost of enqueueing ACS launches: wall time of one v224hip_stream_decode_dev call of 8 chunks (1 088 launches + 8
traceback kernels) on an idle decoder -- the call returns when everything is enqueued -- against the GPU time of the same work."""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from conftest import load_pkg
from importlib import import_module
pkg = load_pkg()
synth = import_module("isee3_decoder_amd.synth")
L = 200_000
syms, bits, _ = synth.coded_stream(1000, L, 3.0, 24.0, 1.0)
dsy, out = pkg.DeviceBuffer.from_numpy(syms), pkg.DeviceBuffer(L)
d = pkg.Viterbi224(200 + 2040)
d.init(0); d.stream_decode_dev(dsy, 16320, 200, out); d.sync()
for rep in range(5):
    d.init(0); d.sync()
    t0 = time.perf_counter()
    d.stream_decode_dev(dsy, 16320, 200, out)
    t1 = time.perf_counter()
    d.sync()
    t2 = time.perf_counter()
    print("enqueue of 1088 launches: %.2f ms = %.2f us per launch; GPU done after %.2f ms = %.2f us per launch" %
          ((t1 - t0) * 1e3, (t1 - t0) / 1096 * 1e6, (t2 - t0) * 1e3, (t2 - t0) / 1088 * 1e6), flush=True)
