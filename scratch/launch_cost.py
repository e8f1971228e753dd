"""Host cost of one ACS launch: enqueue time of update_viterbi224_blk (asynchronous) for a few hundred launches."""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from conftest import load_pkg
pkg = load_pkg()
d = pkg.Viterbi224(3000)
for nl in (50, 100, 200, 400):
    syms = np.full(2 * 15 * nl, 128, dtype=np.uint8)
    ds = pkg.DeviceBuffer.from_numpy(syms)
    d.init(0); d.sync()
    best = 1e9
    for rep in range(5):
        d.sync()
        t0 = time.perf_counter(); d.update_dev(ds, 15 * nl); t1 = time.perf_counter(); d.sync(); t2 = time.perf_counter()
        best = min(best, t1 - t0)
    print("%d launches: enqueue %.1f us = %.2f us per launch; with the GPU %.1f us = %.2f per launch" % (nl, best * 1e6, best * 1e6 / nl, (t2 - t0) * 1e6, (t2 - t0) * 1e6 / nl), flush=True)
