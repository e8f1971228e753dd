"""Where does the run-to-run spread of the two-decoder rate come from?  Same 4M-symbol stream, verified split on two decoders:
several timed repeats per decoder pair, several freshly created pairs per process."""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from conftest import load_pkg
from importlib import import_module
pkg = load_pkg()
synth = import_module("isee3_decoder_amd.synth")
nbits = 2_000_000
syms, bits, _ = synth.coded_stream(1000, nbits, 3.0, 24.0, 1.0)
dsy, dout = pkg.DeviceBuffer.from_numpy(syms), pkg.DeviceBuffer(nbits)
for trial in range(4):
    decs = [pkg.Viterbi224(200 + 2040) for _ in range(2)]
    for d in decs:
        d.set_option("chunk", 2040)
    rates = []
    for rep in range(6):
        t0 = time.perf_counter()
        pkg.stream_decode_split(decs, dsy, nbits, 200, dout, 14280)
        dt = time.perf_counter() - t0
        rates.append(2 * nbits / dt / 1e6)
    decs[0].init(0)
    t0 = time.perf_counter(); decs[0].stream_decode_dev(dsy, nbits, 200, dout); decs[0].sync(); dt = time.perf_counter() - t0
    print("pair %d: split %s Msym/s; decoder 0 alone %.3f" % (trial, " ".join("%.3f" % r for r in rates), 2 * nbits / dt / 1e6), flush=True)
    for d in decs:
        d.close()
