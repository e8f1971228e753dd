import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from conftest import load_pkg
pkg = load_pkg()
for n in (2241, 2241, 2241, 1024, 201):
    t0 = time.perf_counter(); d = pkg.Viterbi224(n); d.sync(); t1 = time.perf_counter(); d.close(); t2 = time.perf_counter()
    print("len %d: create %.2f ms, close %.2f ms" % (n, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
