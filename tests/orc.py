"""ctypes bindings to the CPU oracle (oracle/liboracle.so) and, where present, to the real
reference built by oracle/Makefile into oracle/_ref/.

TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() -- never by the product path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")
GOLDEN = os.path.join(ROOT, "tests", "golden")

NSTATES = 1 << 23
ROWBYTES = NSTATES // 8
LITERAL, FAST = 0, 1

u8p = C.POINTER(C.c_uint8)
i16p = C.POINTER(C.c_int16)


def _ptr(a, t):
    return a.ctypes.data_as(t)


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "all"], check=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(so):
            build_oracle()
        # the GPU box exposes far more hardware threads than its CPU share: cap OpenMP
        os.environ.setdefault("OMP_NUM_THREADS", str(min(8, os.cpu_count() or 1)))
        L = C.CDLL(so)
        L.orc_v224_create.restype = C.c_void_p
        L.orc_v224_create.argtypes = [C.c_int, C.c_int]
        L.orc_v224_init.argtypes = [C.c_void_p, C.c_int]
        L.orc_v224_update.argtypes = [C.c_void_p, u8p, C.c_int]
        L.orc_v224_chainback.argtypes = [C.c_void_p, u8p, C.c_uint, C.c_uint]
        L.orc_v224_decodebit.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_v224_delete.argtypes = [C.c_void_p]
        L.orc_v224_decodeword.restype = C.c_uint64
        L.orc_v224_decodeword.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_v224_row.restype = u8p
        L.orc_v224_row.argtypes = [C.c_void_p, C.c_int]
        L.orc_v224_dp.argtypes = [C.c_void_p]
        L.orc_v224_metric_rel.restype = C.c_uint32
        L.orc_v224_metric_rel.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_v224_spread.restype = C.c_uint32
        L.orc_v224_spread.argtypes = [C.c_void_p]
        L.orc_v224_metric_abs.restype = C.c_uint32
        L.orc_v224_metric_abs.argtypes = [C.c_void_p, C.c_int]
        L.orc_v224_set_metrics.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.orc_v224_get_metrics.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.orc_fnv1a.restype = C.c_uint64
        L.orc_fnv1a.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_encode.restype = C.c_uint64
        L.orc_encode.argtypes = [u8p, u8p, C.c_uint, C.c_uint64]
        L.orc_gen_uniform_bytes.argtypes = [C.c_uint64, u8p, C.c_size_t]
        L.orc_gen_coded_stream.argtypes = [C.c_uint64, C.c_size_t, C.c_double, C.c_double, C.c_int, u8p, u8p]
        L.orc_gen_coded_frame.argtypes = [C.c_uint64, C.c_int, C.c_double, C.c_double, u8p, u8p]
        L.orc_gen_iq.restype = C.c_size_t
        L.orc_gen_iq.argtypes = [C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                 C.c_double, C.c_double, i16p, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.orc_gen_baseband.restype = C.c_size_t
        L.orc_gen_baseband.argtypes = [C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_double,
                                       C.c_double, i16p, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.orc_vdecode.restype = C.c_size_t
        L.orc_vdecode.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_void_p]
        L.orc_symdemod_default.argtypes = [C.c_void_p]
        L.orc_symdemod_set_c.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_symdemod.restype = C.c_size_t
        L.orc_symdemod.argtypes = [C.c_void_p, i16p, C.c_size_t, u8p, C.c_size_t,
                                   C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]
        L.orc_timesearch.restype = C.c_double
        L.orc_timesearch.argtypes = [C.POINTER(C.c_int), i16p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int]
        L.orc_trial_demod.restype = C.c_double
        L.orc_trial_demod.argtypes = [i16p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_double, u8p]
        L.orc_pmdemod_default.argtypes = [C.c_void_p]
        L.orc_pmdemod_fftsize.argtypes = [C.c_void_p]
        L.orc_pmdemod.restype = C.c_size_t
        L.orc_pmdemod.argtypes = [C.c_void_p, i16p, C.c_size_t, i16p, C.POINTER(C.c_double),
                                  C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.orc_fft_forward.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]
        L.orc_icesync_sync_vector.argtypes = [C.c_double, C.c_void_p, C.c_int]
        L.orc_icesync_search.argtypes = [C.c_void_p, C.c_int, C.c_int, i16p, C.c_double, C.c_int, C.c_int,
                                         C.POINTER(C.c_double), C.c_void_p]
        _lib = L
    return _lib


class SymdemodCfg(C.Structure):
    _fields_ = [("samprate_unused", C.c_double), ("samprate", C.c_int), ("symrate", C.c_double),
                ("symbolclocks", C.c_int), ("window", C.c_double), ("clocktrack", C.c_int)]


class PmdemodCfg(C.Structure):
    _fields_ = [("samprate", C.c_double), ("binsize", C.c_double), ("search_freq", C.c_double),
                ("search_width", C.c_double), ("doppler_rate", C.c_double),
                ("cn0_threshold", C.c_double), ("flip", C.c_int)]


class PmdemodBlk(C.Structure):
    _fields_ = [("peak", C.c_int), ("carrier_freq", C.c_double), ("cn0", C.c_double),
                ("amplitude", C.c_double)]


class VdecodeStats(C.Structure):
    _fields_ = [("bits", C.c_uint64), ("symerrs", C.c_uint64), ("flips", C.c_int)]


# ------------------------------------------------------------------ Viterbi wrappers
class _V224Base:
    """Common python face over a create/init/update/chainback/decodebit C API."""

    def init(self, start=0):
        return self._init(self.h, int(start))

    def update(self, syms, nbits=None):
        syms = np.ascontiguousarray(syms, dtype=np.uint8)
        if nbits is None:
            nbits = len(syms) // 2
        assert len(syms) >= 2 * nbits
        return self._update(self.h, _ptr(syms, u8p), int(nbits))

    def chainback(self, nbits, endstate=0):
        out = np.zeros((nbits + 7) // 8, dtype=np.uint8)
        rc = self._chainback(self.h, _ptr(out, u8p), int(nbits), int(endstate) & 0xFFFFFFFF)
        assert rc == 0
        return out

    def decodebit(self, delay, endstate=0):
        return self._decodebit(self.h, int(delay), int(endstate))

    def close(self):
        if self.h:
            self._delete(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class OracleV224(_V224Base):
    def __init__(self, length, mode=FAST):
        L = lib()
        self._init, self._update = L.orc_v224_init, L.orc_v224_update
        self._chainback, self._decodebit, self._delete = L.orc_v224_chainback, L.orc_v224_decodebit, L.orc_v224_delete
        self.h = L.orc_v224_create(int(length), int(mode))
        assert self.h
        self.length = length

    def decodeword(self, delay, endstate=0):
        return int(lib().orc_v224_decodeword(self.h, int(delay), int(endstate)))

    def row(self, r):
        return np.ctypeslib.as_array(lib().orc_v224_row(self.h, int(r)), shape=(ROWBYTES,))

    def row_hash(self, r):
        return int(lib().orc_fnv1a(lib().orc_v224_row(self.h, int(r)), ROWBYTES))

    def dp(self):
        return lib().orc_v224_dp(self.h)

    def spread(self):
        return int(lib().orc_v224_spread(self.h))

    def metric_rel(self, state):
        return int(lib().orc_v224_metric_rel(self.h, int(state)))

    def set_metrics(self, m):
        """resume from the given 2^23 path metrics (uint32, any common offset); rewinds the ring"""
        m = np.ascontiguousarray(m, dtype=np.uint32)
        assert m.size == NSTATES
        assert lib().orc_v224_set_metrics(self.h, m.ctypes.data_as(C.POINTER(C.c_uint32))) == 0

    def get_metrics(self):
        out = np.empty(NSTATES, dtype=np.uint32)
        assert lib().orc_v224_get_metrics(self.h, out.ctypes.data_as(C.POINTER(C.c_uint32))) == 0
        return out


def have_ref():
    return os.path.exists(os.path.join(REF_DIR, "libv224_port_ref.so"))


class RefV224(_V224Base):
    """The real reference decoder (viterbi224_port.c or viterbi224_sse2.c compiled as-is)."""

    def __init__(self, length, variant="port"):
        so = os.path.join(REF_DIR, "libv224_%s_ref.so" % variant)
        # a private copy of the library per instance is NOT needed: the file-static branch
        # table (port.c:16) is read-only after create and identical for every instance
        L = C.CDLL(so)
        L.create_viterbi224.restype = C.c_void_p
        L.create_viterbi224.argtypes = [C.c_int]
        L.init_viterbi224.argtypes = [C.c_void_p, C.c_int]
        L.update_viterbi224_blk.argtypes = [C.c_void_p, u8p, C.c_int]
        L.chainback_viterbi224.argtypes = [C.c_void_p, u8p, C.c_uint, C.c_uint]
        L.decodebit_viterbi224.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.delete_viterbi224.argtypes = [C.c_void_p]
        self.L = L
        self._init, self._update = L.init_viterbi224, L.update_viterbi224_blk
        self._chainback, self._decodebit, self._delete = (L.chainback_viterbi224, L.decodebit_viterbi224,
                                                          L.delete_viterbi224)
        self.h = L.create_viterbi224(int(length))
        assert self.h
        self.length = length
        if variant.startswith("sse2"):             # the port implements six of the nine functions
            L.decodeword_viterbi224.restype = C.c_uint64
            L.decodeword_viterbi224.argtypes = [C.c_void_p, C.c_int, C.c_int]
            L.max_metric_viterbi224.argtypes = [C.c_void_p]
            L.min_metric_viterbi224.argtypes = [C.c_void_p]

    def decodeword(self, delay, endstate=0):
        return int(self.L.decodeword_viterbi224(self.h, int(delay), int(endstate)))


# ------------------------------------------------------------------ generators
def gen_uniform(seed, n):
    out = np.zeros(n, dtype=np.uint8)
    lib().orc_gen_uniform_bytes(seed, _ptr(out, u8p), n)
    return out


def gen_coded_stream(seed, nbits, ebn0_db=3.0, amplitude=24.0, noise_blocks_pct=0):
    syms = np.zeros(2 * nbits, dtype=np.uint8)
    bits = np.zeros(nbits, dtype=np.uint8)
    lib().orc_gen_coded_stream(seed, nbits, ebn0_db, amplitude, noise_blocks_pct, _ptr(syms, u8p), _ptr(bits, u8p))
    return syms, bits


def gen_coded_frame(seed, framebits, ebn0_db=3.0, amplitude=24.0):
    syms = np.zeros(2 * framebits, dtype=np.uint8)
    data = np.zeros(framebits // 8, dtype=np.uint8)
    lib().orc_gen_coded_frame(seed, framebits, ebn0_db, amplitude, _ptr(syms, u8p), _ptr(data, u8p))
    return syms, data


def encode(data, encstate=0):
    data = np.ascontiguousarray(data, dtype=np.uint8)
    syms = np.zeros(16 * len(data), dtype=np.uint8)
    st = lib().orc_encode(_ptr(syms, u8p), _ptr(data, u8p), len(data), encstate)
    return syms, int(st)


def gen_baseband(seed, samprate, seconds, symrate=1024.545058, amp=2000.0, noise_sigma=4000.0):
    n = int(samprate * seconds)
    out = np.zeros(n, dtype=np.int16)
    cap = int(seconds * symrate) + 16
    sent = np.zeros(cap, dtype=np.uint8)
    ns = C.c_size_t(0)
    lib().orc_gen_baseband(seed, samprate, seconds, symrate, amp, noise_sigma, _ptr(out, i16p),
                           _ptr(sent, u8p), cap, C.byref(ns))
    return out, sent[:min(ns.value, cap)]


def gen_iq(seed, samprate, seconds, fc_hz=12345.678, beta=1.1, symrate=1024.545058, amp=3000.0,
           cn0_dbhz=45.0):
    n = int(samprate * seconds)
    iq = np.zeros(2 * n, dtype=np.int16)
    cap = int(seconds * symrate) + 16
    sent = np.zeros(cap, dtype=np.uint8)
    ns = C.c_size_t(0)
    lib().orc_gen_iq(seed, samprate, seconds, fc_hz, beta, symrate, amp, cn0_dbhz, _ptr(iq, i16p),
                     _ptr(sent, u8p), cap, C.byref(ns))
    return iq, sent[:min(ns.value, cap)]


# ------------------------------------------------------------------ stage wrappers
def vdecode(syms, delay=200, start_phase=0, dontflip=False, mode=FAST):
    syms = np.ascontiguousarray(syms, dtype=np.uint8)
    buf = C.create_string_buffer(len(syms) // 2 + 2)
    st = VdecodeStats()
    n = lib().orc_vdecode(_ptr(syms, u8p), len(syms), delay, start_phase, int(dontflip), mode, buf, C.byref(st))
    return buf.raw[:n], st


def symdemod(samples, samprate=250000, c_opt=None, window=1.0, clocktrack=False, symbolclocks=None):
    samples = np.ascontiguousarray(samples, dtype=np.int16)
    cfg = SymdemodCfg()
    lib().orc_symdemod_default(C.byref(cfg))
    cfg.samprate = int(samprate)
    if c_opt is not None:
        lib().orc_symdemod_set_c(C.byref(cfg), str(c_opt).encode())
    if symbolclocks is not None:
        cfg.symbolclocks = symbolclocks
    cfg.window = window
    cfg.clocktrack = int(clocktrack)
    cap = int(len(samples) / samprate * cfg.symrate) + 4096
    out = np.zeros(cap, dtype=np.uint8)
    logcap = int(len(samples) / (samprate * window)) + 8
    ph = (C.c_int * logcap)()
    en = (C.c_double * logcap)()
    nw = C.c_int(0)
    n = lib().orc_symdemod(C.byref(cfg), _ptr(samples, i16p), len(samples), _ptr(out, u8p), cap, ph, en, logcap,
                           C.byref(nw))
    k = min(nw.value, logcap)
    return out[:n], list(ph[:k]), list(en[:k])


def pmdemod(iq, samprate=250000.0, binsize=4.0, search_freq=0.0, search_width=0.0, doppler_rate=0.0,
            cn0_threshold=21.0, flip=False, want_pre=True):
    iq = np.ascontiguousarray(iq, dtype=np.int16)
    cfg = PmdemodCfg()
    lib().orc_pmdemod_default(C.byref(cfg))
    cfg.samprate, cfg.binsize = samprate, binsize
    cfg.search_freq, cfg.search_width, cfg.doppler_rate = search_freq, search_width, doppler_rate
    cfg.cn0_threshold, cfg.flip = cn0_threshold, int(flip)
    N = lib().orc_pmdemod_fftsize(C.byref(cfg))
    nsamp = len(iq) // 2
    nb = nsamp // N
    out = np.zeros(nb * N, dtype=np.int16)
    pre = np.zeros(nb * N, dtype=np.float64) if want_pre else None
    blk = (PmdemodBlk * max(nb, 1))()
    nblk = C.c_int(0)
    lib().orc_pmdemod(C.byref(cfg), _ptr(iq, i16p), nsamp, _ptr(out, i16p),
                      pre.ctypes.data_as(C.POINTER(C.c_double)) if want_pre else None, blk, nb, C.byref(nblk))
    rep = [dict(peak=b.peak, carrier_freq=b.carrier_freq, cn0=b.cn0, amplitude=b.amplitude) for b in blk[:nb]]
    return out, pre, rep, N


def fft_forward(x):
    x = np.ascontiguousarray(x, dtype=np.complex128)
    out = np.zeros_like(x)
    lib().orc_fft_forward(x.view(np.float64).ctypes.data_as(C.POINTER(C.c_double)),
                          out.view(np.float64).ctypes.data_as(C.POINTER(C.c_double)), len(x))
    return out


def ref_cli(name, args, stdin_bytes, timeout=3600):
    """Run one of the reference's own pipe stages (oracle/_ref/<name>) on a byte string."""
    exe = os.path.join(REF_DIR, name)
    p = subprocess.run([exe] + list(args), input=stdin_bytes, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout, check=True)
    return p.stdout


ICESYNC_FAIL = -1234567890


def icesync_sync_vector(symbolsamples):
    cap = int(34 * symbolsamples + 2)
    v = np.zeros(cap, np.float64)
    n = lib().orc_icesync_sync_vector(float(symbolsamples), v.ctypes.data, cap)
    assert n > 0
    return v[:n]


def icesync_search(vec, corr_size, samples, framesamples, low, high, want_result=False):
    """oracle fft_sync_search (icesync.c:139-208): (peakindex or ICESYNC_FAIL, maxpeak[, Corr_result])"""
    vec = np.ascontiguousarray(vec, dtype=np.float64)
    samples = np.ascontiguousarray(samples, dtype=np.int16)
    mp = C.c_double(0)
    res = np.zeros(corr_size, np.float64) if want_result else None
    pk = lib().orc_icesync_search(vec.ctypes.data, len(vec), int(corr_size), _ptr(samples, i16p), float(framesamples),
                                  int(low), int(high), C.byref(mp), res.ctypes.data if want_result else None)
    return (pk, mp.value, res) if want_result else (pk, mp.value)
