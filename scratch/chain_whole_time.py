import sys, time, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from conftest import load_pkg
pkg = load_pkg()
from importlib import import_module
synth = import_module("isee3_decoder_amd.synth")
fs = 250000.0
iq, bits = synth.iq_capture(3, fs, 60.0)[:2]
for mode in ("0", "1", "0", "1"):
    os.environ["ISEE3_CHAIN_WHOLE"] = mode
    t0 = time.perf_counter(); out = pkg.run_chain(iq, samprate=fs, binsize=1.0, symrate="1024"); t1 = time.perf_counter()
    print("whole", mode, "%.1f ms" % ((t1 - t0) * 1e3), len(out))
