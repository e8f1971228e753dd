"""Why do two decoders' launches take longer per pair in the framed pattern than in one long stream?  Passes only
(update_dev is asynchronous), two decoders fed alternately in slabs of 1035 bits: (a) one long stream, (b) init before
every slab, (c) init before every slab + the ring pointer back at 0 (what a frame does), with rings of 1035 / 2070 / 4140 rows."""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from conftest import load_pkg
pkg = load_pkg()
slab, nslab = 1035, 24
syms = np.random.default_rng(1).integers(0, 256, 2 * slab, dtype=np.uint8)
ds = pkg.DeviceBuffer.from_numpy(syms)
for rows in (1035, 2070, 4140):
    decs = [pkg.Viterbi224(rows) for _ in range(2)]
    for mode in ("stream", "init per slab"):
        for nd in (1, 2):
            best = 1e9
            for rep in range(3):
                for d in decs[:nd]:
                    d.init(0); d.sync()
                t0 = time.perf_counter()
                for s in range(nslab):
                    for d in decs[:nd]:
                        if mode != "stream":
                            d.init(0)
                        d.update_dev(ds, slab)
                for d in decs[:nd]:
                    d.sync()
                best = min(best, time.perf_counter() - t0)
            print("rows %4d, %-13s, %d decoder(s): %.2f us per pass and decoder pair" % (rows, mode, nd, best * 1e6 / (nslab * 69)), flush=True)
    for d in decs:
        d.close()
