"""Stability run on the final build: many chain runs (all three Viterbi-stage modes, capture in HBM and in host memory, two
chains at a time), framed batches and split decodes interleaved in one process; every result compared with the first."""
import os, sys, time, threading
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from conftest import load_pkg
from importlib import import_module
pkg = load_pkg()
synth = import_module("isee3_decoder_amd.synth")
iq, sent = synth.iq_capture(3, 250000.0, 20.0, amp=None)
d_iq = pkg.DeviceBuffer.from_numpy(iq)
ref = pkg.run_chain(d_iq, samprate=250000.0, binsize=1.0, symrate="1024")
nb = 1000
fsyms = np.concatenate([synth.coded_stream(6000 + f, nb, 3.0, 24.0, 0.0)[0][:2 * nb] for f in range(12)])
decs = [pkg.Viterbi224(2 * 1005) for _ in range(3)]
fref = pkg.decode_frames(decs, fsyms, 12, nb)
ssyms, _, _ = synth.coded_stream(1000, 200_000, 3.0, 24.0, 1.0)
sd = [pkg.Viterbi224(200 + 2040) for _ in range(2)]
dsy, dout = pkg.DeviceBuffer.from_numpy(ssyms), pkg.DeviceBuffer(200_000)
pkg.stream_decode_split(sd, dsy, 200_000, 200, dout); sref = dout.to_numpy(np.uint8).copy()
bad = 0
t0 = time.perf_counter()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
for it in range(n):
    mode = ("progressive", "block", "whole")[it % 3]
    os.environ["ISEE3_CHAIN_MODE"] = mode
    if it % 5 == 4:
        res = [None, None]
        def one(k):
            res[k] = pkg.run_chain(d_iq if k == 0 else iq, samprate=250000.0, binsize=1.0, symrate="1024")
        th = [threading.Thread(target=one, args=(k,)) for k in range(2)]
        [t.start() for t in th]; [t.join() for t in th]
        bad += sum(r != ref for r in res)
    else:
        bad += pkg.run_chain(d_iq if it % 2 else iq, samprate=250000.0, binsize=1.0, symrate="1024") != ref
    if it % 3 == 0:
        bad += not np.array_equal(pkg.decode_frames(decs[:2 + it % 2], fsyms, 12, nb), fref)
    if it % 4 == 0:
        pkg.stream_decode_split(sd, dsy, 200_000, 200, dout); bad += not np.array_equal(dout.to_numpy(np.uint8), sref)
    if it % 25 == 24:
        print("iteration %d, %.1f s, mismatches %d" % (it + 1, time.perf_counter() - t0, bad), flush=True)
print("RESULT: %d iterations, %d mismatches" % (n, bad))
