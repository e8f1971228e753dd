"""Is the spread of the two-decoder rate a matter of which two decoders (where their buffers landed)?  N decoders in one
process, the verified split on every pair (i, j), both orders."""
import sys, time, itertools
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from conftest import load_pkg
from importlib import import_module
pkg = load_pkg()
synth = import_module("isee3_decoder_amd.synth")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5
nbits = 1_000_000
syms, bits, _ = synth.coded_stream(1000, nbits, 3.0, 24.0, 1.0)
dsy, dout = pkg.DeviceBuffer.from_numpy(syms), pkg.DeviceBuffer(nbits)
decs = [pkg.Viterbi224(200 + 2040) for _ in range(N)]
M = np.zeros((N, N))
for i, j in itertools.permutations(range(N), 2):
    best = 0
    for rep in range(2):
        t0 = time.perf_counter()
        pkg.stream_decode_split([decs[i], decs[j]], dsy, nbits, 200, dout, 14280)
        best = max(best, 2 * nbits / (time.perf_counter() - t0) / 1e6)
    M[i, j] = best
print("split rate, Msymbols/s: row = first decoder (part 0), column = second")
for i in range(N):
    print("  " + " ".join("  -  " if i == j else "%.3f" % M[i, j] for j in range(N)), flush=True)
