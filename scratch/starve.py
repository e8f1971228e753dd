"""How much do two busy decoders slow the chain's front end?  The chain runs with ONE decoder (ISEE3_CHAIN_SHARE=0) while a
background thread keeps a second decoder busy on its own stream; stage engine times with / without, and with stream
priorities (BG_PRIO=low: background decoder low; CHAIN_DEC_PRIO=low: the chain's own decoder low; ISEE3DSP_HIGH_PRIORITY=1)."""
import os, sys, time, threading
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
os.environ["ISEE3_CHAIN_SHARE"] = "0"
import numpy as np
from conftest import load_pkg
from importlib import import_module
pkg = load_pkg()
synth = import_module("isee3_decoder_amd.synth")
iq, sent = synth.iq_capture(3, 250000.0, 60.0, amp=None)
d_iq = pkg.DeviceBuffer.from_numpy(iq)
nbits = 1_000_000
syms, bits, _ = synth.coded_stream(1000, nbits, 3.0, 24.0, 1.0)
dsy, dout = pkg.DeviceBuffer.from_numpy(syms), pkg.DeviceBuffer(nbits)
bg = os.environ.get("BG", "1") == "1"
if os.environ.get("BG_PRIO"):
    os.environ["V224HIP_STREAM_PRIORITY"] = os.environ["BG_PRIO"]
dec = pkg.Viterbi224(200 + 2040); dec.set_option("chunk", 2040)
os.environ.pop("V224HIP_STREAM_PRIORITY", None)
if os.environ.get("CHAIN_DEC_PRIO"):
    os.environ["V224HIP_STREAM_PRIORITY"] = os.environ["CHAIN_DEC_PRIO"]
stop = [False]
def busy():
    while not stop[0]:
        dec.init(0)
        dec.stream_decode_dev(dsy, 200000, 200, dout); dec.sync()
th = threading.Thread(target=busy)
if bg:
    th.start()
ms = []
for rep in range(5):
    t0 = time.perf_counter()
    out = pkg.run_chain(d_iq, samprate=250000.0, binsize=1.0, symrate="1024", decode_delay=200, stage_ms=ms)
    dt = time.perf_counter() - t0
    if rep >= 2:
        print("%s: chain %.2f ms, pmdemod %.2f symdemod %.2f vdecode %.2f" % (os.environ.get("LABEL", ""), dt * 1e3, ms[0], ms[1], ms[2]), flush=True)
stop[0] = True
if bg:
    th.join()
