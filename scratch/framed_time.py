import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from conftest import load_pkg
pkg = load_pkg()
from importlib import import_module
synth = import_module("isee3_decoder_amd.synth")
nb = 1024
syms, bits, _ = synth.coded_stream(1, nb, 4.0, 24.0, 0.0)
d = pkg.Viterbi224(nb)
for rep in range(3):
    d.init(0); d.sync()
    t0 = time.perf_counter(); d.update(syms, nb); d.sync(); t1 = time.perf_counter()
    out = d.chainback(nb, 0); t2 = time.perf_counter()
    print("update %.3f ms, chainback %.3f ms" % ((t1-t0)*1e3, (t2-t1)*1e3))
# two decoders alternating frames
d2 = pkg.Viterbi224(nb)
t0 = time.perf_counter()
for rep in range(4):
    d.init(0); d2.init(0)
    d.update(syms, nb); d2.update(syms, nb)
    a = d.chainback(nb, 0); b = d2.chainback(nb, 0)
t1 = time.perf_counter()
print("2 decoders x 4 frames: %.3f ms per frame" % ((t1-t0)*1e3/8))
# batch entry point: 32 frames over 1, 2, 3 decoders
d3 = pkg.Viterbi224(nb)
frames = np.tile(syms[:2 * nb], 32)
for decs in ([d], [d, d2], [d, d2, d3]):
    pkg.decode_frames(decs, frames, 32, nb)      # (same size as the timed call: staging buffers grow once)
    t0 = time.perf_counter(); out = pkg.decode_frames(decs, frames, 32, nb); t1 = time.perf_counter()
    print("decode_frames, %d decoder(s): %.3f ms per 1024-bit frame = %.3f Msymbols/s" % (len(decs), (t1 - t0) * 1e3 / 32, 32 * 2 * nb / (t1 - t0) / 1e6))
# rings of two padded frames: tracebacks under the next frame's passes, no remainder pass
big = [pkg.Viterbi224(2 * 1035) for _ in range(3)]
for decs in (big[:1], big[:2], big[:3]):
    pkg.decode_frames(decs, frames, 32, nb)      # (same size as the timed call: staging buffers grow once)
    t0 = time.perf_counter(); out2 = pkg.decode_frames(decs, frames, 32, nb); t1 = time.perf_counter()
    assert np.array_equal(out2, out)
    print("decode_frames, %d decoder(s) with 2 x 1035 rows: %.3f ms per 1024-bit frame = %.3f Msymbols/s" % (len(decs), (t1 - t0) * 1e3 / 32, 32 * 2 * nb / (t1 - t0) / 1e6))
