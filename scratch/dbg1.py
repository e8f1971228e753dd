import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from conftest import load_pkg
import orc
pkg = load_pkg()
z = np.load('tests/golden/viterbi_framed.npz')
syms = z['uniform256/syms']
n = 48
ref = pkg.Viterbi224(n, 0, 0); ref.init(0); ref.update(syms, n)
rrows = [ref.export_row(r) for r in range(n)]
for K in (2, 3, 4, 6):
    d = pkg.Viterbi224(n, 1, K); d.init(0); d.update(syms, n)
    for r in range(n):
        a = d.export_row(r)
        if not np.array_equal(a, rrows[r]):
            x = np.unpackbits(a ^ rrows[r], bitorder='little')
            idx = np.flatnonzero(x)
            print('K', K, 'first diff row', r, 'nbits differing', len(idx), 'states', [bin(i) for i in idx[:6]])
            break
    else:
        print('K', K, 'all rows equal')
    # metrics after n steps
    print('  metrics equal:', np.array_equal(d.export_metrics(), ref.export_metrics()))
    d.close()
# step-by-step metric comparison for K=2
for K in (2,):
    for steps in (22, 24, 26):
        a = pkg.Viterbi224(64, 0, 0); a.init(0); a.update(syms, steps)
        b = pkg.Viterbi224(64, 1, K); b.init(0); b.update(syms, steps)
        ma, mb = a.export_metrics(), b.export_metrics()
        print('K', K, 'steps', steps, 'metrics equal', np.array_equal(ma, mb), 'min/max', ma.min(), ma.max(), mb.min(), mb.max(), 'minmetric', a.min_metric(), b.min_metric())
        a.close(); b.close()
