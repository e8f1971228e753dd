"""One-off evidence run: the bench's 10^7-symbol stream through v224hip_progressive_* (fed in pieces of 1..8191 bits, as fast
as the host can) against the verified split and one decoder."""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from conftest import load_pkg
from importlib import import_module
pkg = load_pkg()
synth = import_module("isee3_decoder_amd.synth")
nbits, delay = 5_000_000, 200
syms, bits, _ = synth.coded_stream(1000, nbits, 3.0, 24.0, 1.0)
decs = [pkg.Viterbi224(delay + 2040) for _ in range(2)]
dsy, dout = pkg.DeviceBuffer.from_numpy(syms), pkg.DeviceBuffer(nbits)
t0 = time.perf_counter(); redone = pkg.stream_decode_split(decs, dsy, nbits, delay, dout); t1 = time.perf_counter()
split = dout.to_numpy(np.uint8).copy()
decs[0].init(0)
t2 = time.perf_counter(); decs[0].stream_decode_dev(dsy, nbits, delay, dout); decs[0].sync(); t3 = time.perf_counter()
one = dout.to_numpy(np.uint8).copy()
rng = np.random.default_rng(7)
for expected in (nbits, nbits // 2, 2 * nbits):
    p = pkg.ProgressiveDecode(decs, expected, delay, 2040)
    pos = 0
    t4 = time.perf_counter()
    while pos < nbits:
        n = min(int(rng.integers(1, 8192)), nbits - pos)
        p.feed(syms[2 * pos:2 * (pos + n)]); pos += n
    got, red = p.end()
    t5 = time.perf_counter()
    print("progressive, expected %d: %.3f s (%.3f Msymbols/s), second part decoded again: %d, identical to one decoder: %s, to the split: %s"
          % (expected, t5 - t4, 2 * nbits / (t5 - t4) / 1e6, red, np.array_equal(got, one), np.array_equal(got, split)), flush=True)
print("split %.3f s (%.3f Msymbols/s, parts redone %d); one decoder %.3f s (%.3f Msymbols/s); split identical to one decoder: %s"
      % (t1 - t0, 2 * nbits / (t1 - t0) / 1e6, redone, t3 - t2, 2 * nbits / (t3 - t2) / 1e6, np.array_equal(split, one)))
