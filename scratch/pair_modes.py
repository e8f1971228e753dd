"""Why does the verified split run its two decoders at ~21.5 us per launch pair when two independent streams on the same two
decoders run at ~27.3?  Same pair of decoders, one process: the split, then independent streams in several arrangements."""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from conftest import load_pkg
from importlib import import_module
pkg = load_pkg()
synth = import_module("isee3_decoder_amd.synth")
L = 2_448_000                                   # 150 slabs of 16 320 bits
syms, bits, _ = synth.coded_stream(1000, L, 3.0, 24.0, 1.0)
dsy = pkg.DeviceBuffer.from_numpy(syms)
decs = [pkg.Viterbi224(200 + 2040) for _ in range(2)]
outs = [pkg.DeviceBuffer(L) for _ in range(2)]
slab, half = 16320, L // 2

def independent(name, off1, first1=0, slab1=slab, order=(0, 1)):
    """decoder 0 decodes [0, half), decoder 1 decodes [off1, off1 + half); decoder 1's first call takes first1 bits"""
    for d in decs: d.init(0)
    pos, end = [0, off1], [half, off1 + half]
    t0 = time.perf_counter()
    if first1:
        decs[1].stream_decode_dev(dsy, first1, 200, outs[1], sym_offset=2 * pos[1], out_offset=pos[1]); pos[1] += first1
    while pos[0] < end[0] or pos[1] < end[1]:
        for j in order:
            if pos[j] < end[j]:
                n = min(slab1 if j == 1 else slab, end[j] - pos[j])
                decs[j].stream_decode_dev(dsy, n, 200, outs[j], sym_offset=2 * pos[j], out_offset=pos[j]); pos[j] += n
    for d in decs: d.sync()
    dt = time.perf_counter() - t0
    print("%-58s %.3f Msym/s aggregate (%.2f us per launch pair)" % (name, 2 * L / dt / 1e6, dt / (half / 15) * 1e6), flush=True)

for rep in range(2):
    t0 = time.perf_counter()
    pkg.stream_decode_split(decs, dsy, L, 200, outs[0], 14280)
    dt = time.perf_counter() - t0
    print("%-58s %.3f Msym/s (%.2f us per launch pair)" % ("verified split", 2 * L / dt / 1e6, dt / ((L + 14280) / 2 / 15) * 1e6), flush=True)
    independent("independent, both from bit 0 (same data)", 0)
    independent("independent, decoder 1 from the middle (other data)", half)
    independent("independent, decoder 1 from the middle, first call 6 chunks", half, first1=6 * 2040)
    independent("independent, decoder 1 from the middle, slabs of 7 chunks", half, slab1=7 * 2040)
    independent("independent, decoder 1 enqueued first", half, order=(1, 0))
