"""configs[3]'s eight captures (seeds 3..10, one per GPU in the real configuration) through the chain on ONE GPU, one after
the other: progressive and block mode give the same bits, and the decoded run is found in the sent stream."""
import os, sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from conftest import load_pkg
from importlib import import_module
pkg = load_pkg()
synth = import_module("isee3_decoder_amd.synth")
sys.path.insert(0, '.')
import bench
for seed in range(3, 11):
    iq, sent = synth.iq_capture(seed, 250000.0, 60.0, amp=None)
    d_iq = pkg.DeviceBuffer.from_numpy(iq)
    out = {}
    for mode in ("progressive", "block"):
        os.environ["ISEE3_CHAIN_MODE"] = mode
        pkg.run_chain(d_iq, samprate=250000.0, binsize=1.0, symrate="1024")
        t0 = time.perf_counter(); out[mode] = pkg.run_chain(d_iq, samprate=250000.0, binsize=1.0, symrate="1024"); dt = time.perf_counter() - t0
        out[mode + "_ms"] = dt * 1e3
    nbits, ok = bench.chain_check(out["progressive"], sent)
    print("seed %d: %d bits, progressive %.2f ms, block %.2f ms, identical %s, decoded run found in the sent stream %s"
          % (seed, nbits, out["progressive_ms"], out["block_ms"], out["progressive"] == out["block"], ok), flush=True)
    d_iq.free()
