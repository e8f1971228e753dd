"""Two decoders on two independent streams at the same time (two host threads), decoder 0 with option "prio" = argv[1]:
each decoder's own rate, and what a stream shared in proportion to those rates would reach."""
import sys, time, threading
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from conftest import load_pkg
from importlib import import_module
pkg = load_pkg()
synth = import_module("isee3_decoder_amd.synth")
prio = int(sys.argv[1]) if len(sys.argv) > 1 else 0
stagger_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0      # decoder 1 starts this much later
nbits = 1_200_000
syms, bits, _ = synth.coded_stream(1000, nbits, 3.0, 24.0, 1.0)
dsy = pkg.DeviceBuffer.from_numpy(syms)
decs = [pkg.Viterbi224(200 + 2040) for _ in range(2)]
outs = [pkg.DeviceBuffer(nbits) for _ in range(2)]
if prio:      # option "prio" (every wave of the decoder's launches at s_setprio 3) existed only while this was measured: commit history
    decs[0].set_option("prio", prio)
for rep in range(3):
    t_end = [0.0, 0.0]
    gate = threading.Barrier(2)
    us = [0.0, 0.0]
    def run(j):
        d = decs[j]
        d.init(0)
        d.set_option("profile", 4); d.acs_stats(reset=True)
        gate.wait()
        if j == 1 and stagger_ms > 0: time.sleep(stagger_ms * 1e-3)
        t0 = time.perf_counter()
        for lap in range(2 if j == (2 + rep) % 2 else 1):       # the other decoder keeps going: this one always has company
            for pos in range(0, nbits, 16320):
                d.stream_decode_dev(dsy, min(16320, nbits - pos), 200, outs[j], sym_offset=2 * pos, out_offset=pos)
            if lap == 0:
                d.sync()
                t_end[j] = time.perf_counter() - t0
                l, ms, st = d.acs_stats(); us[j] = ms / l * 1e3
        d.sync(); d.set_option("profile", 0)
    ts = [threading.Thread(target=run, args=(j,)) for j in range(2)]
    [t.start() for t in ts]; [t.join() for t in ts]
    # both ran side by side for min(t) seconds: rates over that common stretch
    tc = min(t_end)
    r = [2 * nbits / t / 1e6 for t in t_end]
    print("stagger %.3f ms " % stagger_ms, end="")
    print("prio %d rep %d (decoder %d runs two laps): decoder 0 %.3f Msym/s %.2f us/launch, decoder 1 %.3f Msym/s %.2f us/launch" % (prio, rep, (2 + rep) % 2, r[0], us[0], r[1], us[1]), flush=True)
same = np.array_equal(outs[0].to_numpy(np.uint8), outs[1].to_numpy(np.uint8))
print("identical outputs:", same)
