import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from conftest import load_pkg
import orc
pkg = load_pkg()
from importlib import import_module
synth = import_module("isee3_decoder_amd.synth")
fs = 10e6; secs = 5.0
iq, sent = synth.iq_capture(3, fs, secs, amp=None)
print("iq rms", np.sqrt(np.mean(iq.astype(np.float64)**2)), "clip frac", np.mean(np.abs(iq) >= 32767))
bits = pkg.run_chain(iq, samprate=fs, binsize=1.0, symrate="1024")
got = np.frombuffer(bits, np.uint8) - ord('0')
print("decoded", len(got))
# oracle path
out, pre, rep, N = orc.pmdemod(iq, samprate=fs, binsize=1.0)
print("oracle pmdemod blocks", [(r['peak'], round(r['carrier_freq'],3), round(r['cn0'],2)) for r in rep])
sy, ph, en = orc.symdemod(out, samprate=int(fs), c_opt="1024")
print("oracle symdemod symbols", len(sy), "phases", ph[:6])
ob, st = orc.vdecode(sy)
og = np.frombuffer(ob, np.uint8) - ord('0')
print("oracle decoded", len(og), "equal to gpu:", np.array_equal(og, got[:len(og)]) if len(og) <= len(got) else False)
s = "".join(map(str, sent))
for name, g in (("gpu", got), ("oracle", og)):
    for start in (100, 300, 600, 1000):
        seg = "".join(map(str, g[start:start+200]))
        print(name, start, "found at", s.find(seg))
