"""One v224hip_decode_frames batch (32 frames of 1024 bits, two decoders with rings of two padded frames) for a kernel trace."""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from conftest import load_pkg
from importlib import import_module
pkg = load_pkg()
synth = import_module("isee3_decoder_amd.synth")
nb = 1024
syms, bits, _ = synth.coded_stream(1, nb, 4.0, 24.0, 0.0)
frames = np.tile(syms[:2 * nb], 32)
decs = [pkg.Viterbi224(2 * 1035) for _ in range(2)]
pkg.decode_frames(decs, frames, 4, nb)
t0 = time.perf_counter(); out = pkg.decode_frames(decs, frames, 32, nb); t1 = time.perf_counter()
print("%.3f ms per frame" % ((t1 - t0) * 1e3 / 32))
