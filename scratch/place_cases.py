"""Is the bad pair rate a matter of WHERE the second decoder's buffers land?  Case W of queue_cases.py, then the second
decoder is re-created behind spacers of different sizes."""
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
mk = lambda: pkg.Viterbi224(200 + 2 * 1020)
def rate2(a, b):
    pkg.stream_decode_split([a, b], dsy, nbits, 200, dout, 4080)
    t0 = time.perf_counter(); pkg.stream_decode_split([a, b], dsy, nbits, 200, dout, 4080); a.sync(); b.sync()
    return 2 * nbits / (time.perf_counter() - t0) / 1e6
def rate1(a):
    a.init(0); a.stream_decode_dev(dsy, nbits, 200, dout); a.sync()
    t0 = time.perf_counter(); a.init(0); a.stream_decode_dev(dsy, nbits, 200, dout); a.sync()
    return 2 * nbits / (time.perf_counter() - t0) / 1e6
d0 = mk(); rate1(d0); d0.close()
pm = pkg.PmDemodEngine(1 << 18); sy = pkg.SymDemodEngine(600000)
a = mk(); rate1(a)
spacers = []
for sp in (0, 1 << 20, 16 << 20, 17 << 20, 32 << 20, 3 << 20, 64 << 20, 2 << 20):
    if sp:
        spacers.append(pkg.DeviceBuffer(sp))
    b = mk()
    print("spacer %3d MiB: single b %.3f  pair %.3f Msym/s" % (sp >> 20, rate1(b), rate2(a, b)), flush=True)
    b.close()
