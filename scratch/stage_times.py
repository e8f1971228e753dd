import sys, os, time, subprocess
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from conftest import load_pkg
pkg = load_pkg()
from importlib import import_module
synth = import_module("isee3_decoder_amd.synth")
fs = float(sys.argv[1]); secs = float(sys.argv[2])
iq, sent = synth.iq_capture(5, fs, secs, amp=3000.0, cn0_dbhz=45.0 + 10*np.log10(fs/250000.0))
def run(exe, args, data):
    t = time.perf_counter()
    p = subprocess.run([pkg.cli_path(exe)] + args, input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-500:]
    return p.stdout, time.perf_counter() - t
bb, t1 = run("pmdemod", ["-q", "-r", str(fs), "-b", "1"], iq.tobytes())
sy, t2 = run("symdemod", ["-q", "-r", str(int(fs)), "-c", "1024"], bb)
bits, t3 = run("vdecode", ["-q"], sy)
ch, t4 = run("isee3chain", ["-r", str(int(fs)), "-b", "1", "-c", "1024"], iq.tobytes())
n = len(iq)//2
print("fs %g, %g s, %d samples: pmdemod %.3f s (%.1f MS/s), symdemod %.3f s (%.1f MS/s in), vdecode %.3f s (%d bits), chain %.3f s (%.1f MS/s) same=%s" % (fs, secs, n, t1, n/t1/1e6, t2, len(bb)/2/t2/1e6, t3, len(bits), t4, n/t4/1e6, ch == bits))
