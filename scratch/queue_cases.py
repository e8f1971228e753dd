"""Which creation histories make two decoders NOT overlap?  (HIP stream -> hardware queue mapping)"""
import sys, time, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from conftest import load_pkg
pkg = load_pkg()
from importlib import import_module
synth = import_module("isee3_decoder_amd.synth")
nbits = 400_000
syms, _, _ = synth.coded_stream(5, nbits, 3.0, 24.0, 0.0)
dsy, dout = pkg.DeviceBuffer.from_numpy(syms), pkg.DeviceBuffer(nbits)
case = sys.argv[1]

def mk():
    d = pkg.Viterbi224(200 + 2 * 1020)
    return d

def rate2(a, b):
    pkg.stream_decode_split([a, b], dsy, nbits, 200, dout, 4080)
    t0 = time.perf_counter(); pkg.stream_decode_split([a, b], dsy, nbits, 200, dout, 4080); a.sync(); b.sync()
    return 2 * nbits / (time.perf_counter() - t0) / 1e6

def rate1(a):
    a.init(0); a.stream_decode_dev(dsy, nbits, 200, dout); a.sync()
    t0 = time.perf_counter(); a.init(0); a.stream_decode_dev(dsy, nbits, 200, dout); a.sync()
    return 2 * nbits / (time.perf_counter() - t0) / 1e6

if case == "X":
    a, b = mk(), mk()
elif case == "Y":
    d0 = mk(); rate1(d0); d0.close(); a, b = mk(), mk()
elif case == "Z":
    d0 = mk(); rate1(d0); d0.close(); a = mk(); rate1(a); b = mk()
elif case == "W":       # with front-end handles in between, as the chain has them
    d0 = mk(); rate1(d0); d0.close()
    pm = pkg.PmDemodEngine(1 << 18); sy = pkg.SymDemodEngine(600000)
    a = mk(); rate1(a); b = mk()
elif case == "V":
    pm = pkg.PmDemodEngine(1 << 18); sy = pkg.SymDemodEngine(600000)
    a = mk(); rate1(a); b = mk()
print("case", case, "queues", os.environ.get("GPU_MAX_HW_QUEUES"), ": single %.3f  pair %.3f Msym/s" % (rate1(a), rate2(a, b)), flush=True)
