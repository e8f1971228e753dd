"""A/B of builds of libviterbi224_hip.so (isee3-decoder_amd/lib_alt/<name>/, made with -D flags): lone-decoder and
two-decoder rates on the same stream, and a hash of the decoded bits (every build must print the same one).
usage: python scratch/l15_variants.py <name under lib_alt | default> [nbits]"""
import os, sys, time, zlib
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from conftest import load_pkg
from importlib import import_module
pkg = load_pkg()
name = sys.argv[1] if len(sys.argv) > 1 else "default"
if name != "default":
    pkg.LIB_DIR = os.path.join(os.path.dirname(pkg.LIB_DIR), "lib_alt", name)
    assert os.path.exists(pkg.lib_path("libviterbi224_hip.so")), pkg.LIB_DIR
synth = import_module("isee3_decoder_amd.synth")
nbits = int(sys.argv[2]) if len(sys.argv) > 2 else 1_500_000
syms, bits, _ = synth.coded_stream(1000, nbits, 3.0, 24.0, 1.0)
dsy, dout = pkg.DeviceBuffer.from_numpy(syms), pkg.DeviceBuffer(nbits)
decs = [pkg.Viterbi224(200 + 2040) for _ in range(2)]
lone, us = [], []
for rep in range(4):
    d = decs[0]
    d.init(0)
    if rep:
        d.set_option("profile", 4); d.acs_stats(reset=True)
    t0 = time.perf_counter()
    for pos in range(0, nbits, 16320):
        d.stream_decode_dev(dsy, min(16320, nbits - pos), 200, dout, sym_offset=2 * pos, out_offset=pos)
    d.sync()
    dt = time.perf_counter() - t0
    if rep:
        l, ms, st = d.acs_stats(); d.set_option("profile", 0)
        lone.append(2 * nbits / dt / 1e6); us.append(ms / l * 1e3)
h1 = zlib.crc32(dout.to_numpy(np.uint8).tobytes())
pair = []
for rep in range(6):
    t0 = time.perf_counter()
    pkg.stream_decode_split(decs, dsy, nbits, 200, dout, 14280)
    dt = time.perf_counter() - t0
    if rep:
        pair.append(2 * nbits / dt / 1e6)
h2 = zlib.crc32(dout.to_numpy(np.uint8).tobytes())
print("%-12s lone %s Msym/s (%s us/launch) | pair %s Msym/s | crc %08x %08x" %
      (name, " ".join("%.3f" % r for r in lone), " ".join("%.2f" % u for u in us), " ".join("%.3f" % r for r in pair), h1, h2), flush=True)
