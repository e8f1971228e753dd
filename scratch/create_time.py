import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from conftest import load_pkg
pkg = load_pkg()
for n in (2241, 2241, 1024):
    t0 = time.perf_counter(); d = pkg.Viterbi224(n); d.sync(); t1 = time.perf_counter(); d.close(); t2 = time.perf_counter()
    print("viterbi len %d: create %.2f ms, close %.2f ms" % (n, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
D = pkg.dsp_lib()
import ctypes as C
for rep in range(3):
    t0 = time.perf_counter(); h = D.pmd_create(1 << 18); t1 = time.perf_counter(); D.pmd_destroy(C.c_void_p(h)); t2 = time.perf_counter()
    print("pmd N=2^18: create %.2f ms, destroy %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
    t0 = time.perf_counter(); h = D.symd_create(500000 + 400); t1 = time.perf_counter(); D.symd_destroy(C.c_void_p(h)); t2 = time.perf_counter()
    print("symd 500k: create %.2f ms, destroy %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
