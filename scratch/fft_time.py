"""Time pmdemod's transform at N = 2^23 (or argv[2]) on a build of libisee3dsp_hip.so under lib_alt/<argv[1]> (default: lib/).
Under `rocprofv3 --kernel-trace --stats` the per-pass averages are in the kernel stats."""
import os, sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from conftest import load_pkg
pkg = load_pkg()
name = sys.argv[1] if len(sys.argv) > 1 else "default"
if name != "default":
    pkg.LIB_DIR = os.path.join(os.path.dirname(pkg.LIB_DIR), "lib_alt", name)
N = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 23)
iq = np.random.default_rng(1).integers(-20000, 20000, 2 * N).astype(np.int16)
d = pkg.DeviceBuffer.from_numpy(iq)
eng = pkg.PmDemodEngine(N)
L = pkg.dsp_lib()
import ctypes as C
def once():
    assert L.pmd_load(eng.h, C.c_void_p(d.ptr), 1, 0) == 0
    return eng.fft_peak(0, N)
for _ in range(3):
    once()
t0 = time.perf_counter()
for _ in range(30):
    once()
print("%-8s N=2^%d: %.1f us per load + fft_peak call" % (name, int(np.log2(N)), (time.perf_counter() - t0) / 30 * 1e6), flush=True)
