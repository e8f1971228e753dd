"""Two decoders, two independent streams of lengths f L and (1 - f) L enqueued slab by slab from ONE host thread (the
split's pattern), no profiling events: aggregate rate L / max(finish times), decoder 0 at option "prio" = argv[1]."""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from conftest import load_pkg
from importlib import import_module
pkg = load_pkg()
synth = import_module("isee3_decoder_amd.synth")
prio = int(sys.argv[1]) if len(sys.argv) > 1 else 0
lead_chunks = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0     # decoder 0 gets this many chunks of 2 040 bits enqueued first
L = 2_400_000
syms, bits, _ = synth.coded_stream(1000, L, 3.0, 24.0, 1.0)
dsy = pkg.DeviceBuffer.from_numpy(syms)
decs = [pkg.Viterbi224(200 + 2040) for _ in range(2)]
outs = [pkg.DeviceBuffer(L) for _ in range(2)]
if prio:      # option "prio" (every wave of the decoder's launches at s_setprio 3) existed only while this was measured: commit history
    decs[0].set_option("prio", prio)
slab = 16320
for f in ((0.5,) if lead_chunks else (0.5, 0.54, 0.58, 0.62, 0.5)):
    n = [int(L * f) // slab * slab, int(L * (1 - f)) // slab * slab]
    for rep in range(2):
        for d in decs: d.init(0)
        pos = [0, 0]
        t0 = time.perf_counter()
        if lead_chunks:                      # a head start for decoder 0: its chunk boundaries then fall between decoder 1's
            lead = int(lead_chunks * 2040) // 15 * 15
            decs[0].stream_decode_dev(dsy, lead, 200, outs[0], sym_offset=0, out_offset=0); pos[0] = lead
            n[0] = lead + (n[0] - lead) // slab * slab
        while pos[0] < n[0] or pos[1] < n[1]:
            for j in (0, 1):
                if pos[j] < n[j]:
                    decs[j].stream_decode_dev(dsy, slab, 200, outs[j], sym_offset=2 * pos[j], out_offset=pos[j]); pos[j] += slab
        t_enq = time.perf_counter() - t0
        decs[0].sync(); ta = time.perf_counter() - t0
        decs[1].sync(); tb = time.perf_counter() - t0
    print("lead %.2f chunks " % lead_chunks, end="")
    print("prio %d f %.2f: decoder 0 done at %.1f ms (%d bits), decoder 1 at >= %.1f ms (%d bits), enqueue %.1f ms: aggregate %.3f Msym/s"
          % (prio, f, ta * 1e3, n[0], tb * 1e3, n[1], t_enq * 1e3, 2 * (n[0] + n[1]) / max(ta, tb) / 1e6), flush=True)
