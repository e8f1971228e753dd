"""isee3-decoder_amd -- MI355X-native hot path of the ISEE-3/ICE receive chain.

This package is a thin ctypes face over the C-ABI shared libraries built from csrc/ (hand-written
HIP for gfx950).  There is NO CPU fallback: if a library is missing, or no GPU is visible when a
decoder is created, the call raises.

The directory name contains a hyphen, so load it with `tests/conftest.py::load_pkg()` /
`importlib` under the module name ``isee3_decoder_amd``.
"""
import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(PKG_DIR, "lib")
BIN_DIR = os.path.join(PKG_DIR, "bin")
INCLUDE_DIR = os.path.join(os.path.dirname(PKG_DIR), "include")

ENGINE_SIMPLE, ENGINE_FUSED = 0, 1
NSTATES = 1 << 23
ROWBYTES = NSTATES // 8

u8p = C.POINTER(C.c_uint8)

# every symbol include/viterbi224.h and include/viterbi224_hip.h declare
V224_SYMBOLS = [
    "init_viterbi224", "create_viterbi224", "chainback_viterbi224", "delete_viterbi224",
    "update_viterbi224_blk", "max_metric_viterbi224", "min_metric_viterbi224",
    "decodebit_viterbi224", "decodeword_viterbi224",
    "v224hip_device_count", "v224hip_set_device", "v224hip_create", "v224hip_last_error",
    "v224hip_update_dev", "v224hip_stream_decode", "v224hip_stream_decode_dev",
    "v224hip_stream_chunk", "v224hip_decode_frames", "v224hip_stream_decode_split", "v224hip_stream_decode_shared", "v224hip_progressive_begin", "v224hip_progressive_feed", "v224hip_progressive_end", "v224hip_progressive_abort", "v224hip_set_option", "v224hip_get_counter", "v224hip_sync", "v224hip_acs_stats",
    "v224hip_export_row", "v224hip_export_metrics", "v224hip_dev_alloc", "v224hip_dev_free",
    "v224hip_h2d", "v224hip_d2h",
]


class NativeLibraryMissing(RuntimeError):
    pass


def lib_path(name):
    return os.path.join(LIB_DIR, name)


_v224 = None


def v224_lib():
    """Load libviterbi224_hip.so (raises NativeLibraryMissing when it has not been built)."""
    global _v224
    if _v224 is not None:
        return _v224
    path = lib_path("libviterbi224_hip.so")
    if not os.path.exists(path):
        raise NativeLibraryMissing(
            "%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)" % path)
    L = C.CDLL(path)
    L.create_viterbi224.restype = C.c_void_p
    L.create_viterbi224.argtypes = [C.c_int]
    L.v224hip_create.restype = C.c_void_p
    L.v224hip_create.argtypes = [C.c_int, C.c_int, C.c_int]
    L.v224hip_last_error.restype = C.c_char_p
    L.init_viterbi224.argtypes = [C.c_void_p, C.c_int]
    L.update_viterbi224_blk.argtypes = [C.c_void_p, u8p, C.c_int]
    L.chainback_viterbi224.argtypes = [C.c_void_p, u8p, C.c_uint, C.c_uint]
    L.decodebit_viterbi224.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.decodeword_viterbi224.restype = C.c_ulonglong
    L.decodeword_viterbi224.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.max_metric_viterbi224.argtypes = [C.c_void_p]
    L.min_metric_viterbi224.argtypes = [C.c_void_p]
    L.delete_viterbi224.argtypes = [C.c_void_p]
    L.delete_viterbi224.restype = None
    L.v224hip_set_device.argtypes = [C.c_int]
    L.v224hip_update_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.v224hip_stream_decode.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, u8p]
    L.v224hip_stream_decode_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.v224hip_stream_chunk.argtypes = [C.c_void_p]
    L.v224hip_stream_decode_shared.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), u8p, C.c_int, C.c_int, u8p, C.c_int]
    L.v224hip_stream_decode_split.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                              C.c_int, C.POINTER(C.c_int)]
    L.v224hip_decode_frames.argtypes = [C.POINTER(C.c_void_p), C.c_int, u8p, C.c_int, C.c_int, C.c_int,
                                        C.c_uint, u8p]
    L.v224hip_progressive_begin.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_longlong, C.c_int, C.c_int]
    L.v224hip_progressive_begin.restype = C.c_void_p
    L.v224hip_progressive_feed.argtypes = [C.c_void_p, u8p, C.c_int]
    L.v224hip_progressive_end.argtypes = [C.c_void_p, u8p, C.c_longlong, C.POINTER(C.c_longlong), C.POINTER(C.c_int)]
    L.v224hip_progressive_abort.argtypes = [C.c_void_p]
    L.v224hip_progressive_abort.restype = None
    L.v224hip_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_long]
    L.v224hip_sync.argtypes = [C.c_void_p]
    L.v224hip_get_counter.argtypes = [C.c_void_p, C.c_char_p]
    L.v224hip_get_counter.restype = C.c_long
    L.v224hip_acs_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_double),
                                    C.POINTER(C.c_ulonglong), C.c_int]
    L.v224hip_export_row.argtypes = [C.c_void_p, C.c_int, u8p]
    L.v224hip_export_metrics.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    L.v224hip_dev_alloc.restype = C.c_void_p
    L.v224hip_dev_alloc.argtypes = [C.c_size_t]
    L.v224hip_dev_free.argtypes = [C.c_void_p]
    L.v224hip_dev_free.restype = None
    L.v224hip_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.v224hip_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    _v224 = L
    return L


def last_error():
    return (v224_lib().v224hip_last_error() or b"").decode()


class DeviceBuffer:
    """A chunk of HBM owned through the C-ABI (no torch types cross the boundary)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.ptr = v224_lib().v224hip_dev_alloc(self.nbytes)
        if not self.ptr:
            raise MemoryError("v224hip_dev_alloc(%d) failed" % nbytes)

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        if v224_lib().v224hip_h2d(b.ptr, a.ctypes.data, a.nbytes) != 0:
            raise RuntimeError("h2d failed")
        return b

    def to_numpy(self, dtype=np.uint8, count=None):
        dt = np.dtype(dtype)
        n = self.nbytes // dt.itemsize if count is None else count
        out = np.empty(n, dtype=dt)
        if v224_lib().v224hip_d2h(out.ctypes.data, self.ptr, out.nbytes) != 0:
            raise RuntimeError("d2h failed")
        return out

    def free(self):
        if self.ptr:
            v224_lib().v224hip_dev_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Viterbi224:
    """Python mirror of the reference decoder API (viterbi224.h:8-16) on the HIP library.

    Method names and argument meaning follow the C functions; errors raise instead of
    returning -1/NULL.
    """

    def __init__(self, length, engine=-1, k=0):
        L = v224_lib()
        self.L = L
        self.h = L.v224hip_create(int(length), int(engine), int(k))
        if not self.h:
            raise RuntimeError("create_viterbi224(%d) failed: %s" % (length, last_error()))
        self.length = int(length)

    def _chk(self, rc, what):
        if rc < 0:
            raise RuntimeError("%s failed: %s" % (what, last_error()))
        return rc

    def init(self, starting_state=0):
        return self._chk(self.L.init_viterbi224(self.h, int(starting_state)), "init_viterbi224")

    def update(self, syms, nbits=None):
        syms = np.ascontiguousarray(syms, dtype=np.uint8)
        if nbits is None:
            nbits = len(syms) // 2
        if len(syms) < 2 * nbits:
            raise ValueError("need 2*nbits symbols")
        return self._chk(self.L.update_viterbi224_blk(self.h, syms.ctypes.data_as(u8p), int(nbits)),
                         "update_viterbi224_blk")

    def update_dev(self, dbuf, nbits, byte_offset=0):
        return self._chk(self.L.v224hip_update_dev(self.h, dbuf.ptr + byte_offset, int(nbits)),
                         "v224hip_update_dev")

    def chainback(self, nbits, endstate=0):
        out = np.zeros((nbits + 7) // 8, dtype=np.uint8)
        self._chk(self.L.chainback_viterbi224(self.h, out.ctypes.data_as(u8p), int(nbits),
                                              int(endstate) & 0xFFFFFFFF), "chainback_viterbi224")
        return out

    def decodebit(self, delay, endstate=0):
        return self.L.decodebit_viterbi224(self.h, int(delay), int(endstate))

    def decodeword(self, delay, endstate=0):
        return int(self.L.decodeword_viterbi224(self.h, int(delay), int(endstate)))

    def min_metric(self):
        return self.L.min_metric_viterbi224(self.h)

    def max_metric(self):
        return self.L.max_metric_viterbi224(self.h)

    def stream_decode(self, syms, delay=200, nbits=None):
        syms = np.ascontiguousarray(syms, dtype=np.uint8)
        if nbits is None:
            nbits = len(syms) // 2
        out = np.empty(nbits, dtype=np.uint8)
        self._chk(self.L.v224hip_stream_decode(self.h, syms.ctypes.data_as(u8p), int(nbits), int(delay),
                                               out.ctypes.data_as(u8p)), "v224hip_stream_decode")
        return out

    def stream_decode_dev(self, d_syms, nbits, delay, d_out, sym_offset=0, out_offset=0):
        return self._chk(self.L.v224hip_stream_decode_dev(self.h, d_syms.ptr + sym_offset, int(nbits),
                                                          int(delay), d_out.ptr + out_offset),
                         "v224hip_stream_decode_dev")

    def set_option(self, key, value):
        return self._chk(self.L.v224hip_set_option(self.h, key.encode(), int(value)), "set_option " + key)

    def stream_chunk(self):
        return self.L.v224hip_stream_chunk(self.h)

    def get_counter(self, key):
        """Read and reset a counter ("chainback_redone": pieces of parallel chainbacks walked again after a failed seam check)."""
        r = self.L.v224hip_get_counter(self.h, key.encode())
        if r < 0:
            raise RuntimeError("v224hip_get_counter %s failed: %s" % (key, last_error()))
        return int(r)

    def sync(self):
        return self._chk(self.L.v224hip_sync(self.h), "v224hip_sync")

    def acs_stats(self, reset=False):
        n, ms, st = C.c_ulonglong(0), C.c_double(0), C.c_ulonglong(0)
        self._chk(self.L.v224hip_acs_stats(self.h, C.byref(n), C.byref(ms), C.byref(st), int(reset)), "acs_stats")
        return n.value, ms.value, st.value

    def export_row(self, row):
        out = np.empty(ROWBYTES, dtype=np.uint8)
        self._chk(self.L.v224hip_export_row(self.h, int(row), out.ctypes.data_as(u8p)), "export_row")
        return out

    def export_metrics(self):
        out = np.empty(NSTATES, dtype=np.uint32)
        self._chk(self.L.v224hip_export_metrics(self.h, out.ctypes.data_as(C.POINTER(C.c_uint32))), "export_metrics")
        return out

    def close(self):
        if getattr(self, "h", None):
            self.L.delete_viterbi224(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------
# libisee3dsp_hip.so : pmdemod / symdemod kernels (include/isee3_dsp_hip.h)
# ------------------------------------------------------------------------------------------------
DSP_SYMBOLS = [
    "isee3dsp_last_error", "isee3dsp_set_device", "isee3dsp_share_stream", "isee3dsp_dev_alloc", "isee3dsp_dev_free", "isee3dsp_h2d", "isee3dsp_d2h",
    "symd_create", "symd_destroy", "symd_load", "symd_timesearch", "symd_demod",
    "symd_store_reset", "symd_store_slide", "symd_store_put", "symd_store_scan", "symd_window",
    "pmd_create", "pmd_destroy", "pmd_set_dechirp", "pmd_load", "pmd_fft_peak", "pmd_mix_quantise",
    "pmd_fft_peak_begin", "pmd_fft_peak_end", "pmd_mix_begin", "pmd_mix_end",
    "pmd_get_spectrum",
    "pmd_last_peak_path",
    "isync_create", "isync_destroy", "isync_set_vector", "isync_search",
]



def stream_decode_split(decoders, d_syms, nbits, delay, d_out, warm_bits=14280, sym_offset=0, out_offset=0):
    """v224hip_stream_decode_split: one stream over several decoders with a single decoder's result (verified at
    the seams; see include/viterbi224_hip.h).  Device buffers.  Returns the number of parts that had to be redone."""
    L = v224_lib()
    hs = (C.c_void_p * len(decoders))(*[d.h for d in decoders])
    nfb = C.c_int(0)
    rc = L.v224hip_stream_decode_split(hs, len(decoders), d_syms.ptr + sym_offset, int(nbits), int(delay),
                                       d_out.ptr + out_offset, int(warm_bits), C.byref(nfb))
    if rc != 0:
        raise RuntimeError("v224hip_stream_decode_split: " + L.v224hip_last_error().decode())
    return nfb.value


def stream_decode_shared(decoders, holder, syms, delay, warm_bits=4080):
    """v224hip_stream_decode_shared: the next block (host numpy uint8 symbols) of the stream decoders[holder] is in the
    middle of.  Returns (out uint8[nbits], new holder)."""
    L = v224_lib()
    syms = np.ascontiguousarray(syms, dtype=np.uint8)
    nbits = len(syms) // 2
    out = np.empty(nbits, dtype=np.uint8)
    hs = (C.c_void_p * len(decoders))(*[d.h for d in decoders])
    h = C.c_int(holder)
    rc = L.v224hip_stream_decode_shared(hs, len(decoders), C.byref(h), syms.ctypes.data_as(u8p), nbits, int(delay),
                                        out.ctypes.data_as(u8p), int(warm_bits))
    if rc != 0:
        raise RuntimeError("v224hip_stream_decode_shared: " + L.v224hip_last_error().decode())
    return out, h.value


class ProgressiveDecode:
    """v224hip_progressive_*: a stream that is still arriving, on two decoders with one warm-up, exactly as one decoder
    would decode it.  feed(symbols) as they come, end() -> (bits uint8[n], redone)."""

    def __init__(self, decoders, expected_bits, delay, warm_bits=3060):
        self.L = v224_lib()
        hs = (C.c_void_p * len(decoders))(*[d.h for d in decoders])
        self.h = self.L.v224hip_progressive_begin(hs, len(decoders), int(expected_bits), int(delay), int(warm_bits))
        if not self.h:
            raise RuntimeError("v224hip_progressive_begin: " + self.L.v224hip_last_error().decode())
        self.fed = 0

    def feed(self, syms):
        syms = np.ascontiguousarray(syms, dtype=np.uint8)
        n = len(syms) // 2
        if self.L.v224hip_progressive_feed(self.h, syms.ctypes.data_as(u8p), n) != 0:
            raise RuntimeError("v224hip_progressive_feed: " + self.L.v224hip_last_error().decode())
        self.fed += n

    def end(self):
        out = np.empty(max(self.fed, 1), dtype=np.uint8)
        n, redone = C.c_longlong(0), C.c_int(0)
        h, self.h = self.h, None
        if self.L.v224hip_progressive_end(h, out.ctypes.data_as(u8p), len(out), C.byref(n), C.byref(redone)) != 0:
            raise RuntimeError("v224hip_progressive_end: " + self.L.v224hip_last_error().decode())
        return out[:n.value], redone.value

    def abort(self):
        if self.h:
            self.L.v224hip_progressive_abort(self.h)
            self.h = None

    def __del__(self):
        self.abort()


def decode_frames(decoders, syms, nframes, framebits, startstate=0, endstate=0):
    """v224hip_decode_frames: `nframes` independent frames (init / update(framebits) / chainback each, as
    vtest224.c:116-118 and decode.c:220-222 do), frame f on decoders[f % len(decoders)].  Returns
    uint8[nframes, ceil(framebits/8)]."""
    L = v224_lib()
    syms = np.ascontiguousarray(syms, dtype=np.uint8)
    assert syms.size >= 2 * framebits * nframes
    out = np.zeros((nframes, (framebits + 7) // 8), dtype=np.uint8)
    hs = (C.c_void_p * len(decoders))(*[d.h for d in decoders])
    rc = L.v224hip_decode_frames(hs, len(decoders), syms.ctypes.data_as(u8p), int(nframes), int(framebits),
                                 int(startstate), int(endstate) & 0xFFFFFFFF, out.ctypes.data_as(u8p))
    if rc != 0:
        raise RuntimeError("v224hip_decode_frames: " + L.v224hip_last_error().decode())
    return out

class PmdPeak(C.Structure):
    _fields_ = [("peak", C.c_int), ("maxenergy", C.c_double), ("peak_re", C.c_double), ("peak_im", C.c_double),
                ("next_re", C.c_double), ("next_im", C.c_double), ("prev_re", C.c_double), ("prev_im", C.c_double)]


class PmdMix(C.Structure):
    _fields_ = [("dc_re", C.c_double), ("dc_im", C.c_double), ("amplitude", C.c_double), ("diffsumsq", C.c_double)]


_dsp = None


def dsp_lib():
    global _dsp
    if _dsp is not None:
        return _dsp
    path = lib_path("libisee3dsp_hip.so")
    if not os.path.exists(path):
        raise NativeLibraryMissing("%s not built (no CPU fallback)" % path)
    L = C.CDLL(path)
    L.isee3dsp_last_error.restype = C.c_char_p
    L.isee3dsp_set_device.argtypes = [C.c_int]
    L.symd_create.restype = C.c_void_p
    L.symd_create.argtypes = [C.c_int]
    L.symd_destroy.argtypes = [C.c_void_p]
    L.symd_destroy.restype = None
    L.symd_load.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.symd_store_reset.argtypes = [C.c_void_p]
    L.symd_store_slide.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.symd_store_put.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
    L.symd_store_scan.argtypes = [C.c_void_p, C.c_int]
    L.isee3dsp_dev_alloc.restype = C.c_void_p
    L.isee3dsp_dev_alloc.argtypes = [C.c_size_t]
    L.isee3dsp_dev_free.argtypes = [C.c_void_p]
    L.isee3dsp_dev_free.restype = None
    L.isee3dsp_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.isee3dsp_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.symd_timesearch.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int,
                                  C.POINTER(C.c_double)]
    L.symd_demod.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_int,
                             C.POINTER(C.c_double)]
    L.pmd_create.restype = C.c_void_p
    L.pmd_create.argtypes = [C.c_int]
    L.pmd_destroy.argtypes = [C.c_void_p]
    L.pmd_destroy.restype = None
    L.pmd_set_dechirp.argtypes = [C.c_void_p, C.c_void_p]
    L.pmd_load.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.pmd_fft_peak.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(PmdPeak)]
    L.pmd_mix_quantise.argtypes = [C.c_void_p, C.c_double, C.POINTER(PmdMix), C.c_void_p, C.c_void_p, C.c_int]
    L.pmd_get_spectrum.argtypes = [C.c_void_p, C.c_void_p]
    L.pmd_last_peak_path.argtypes = [C.c_void_p]
    L.isync_create.restype = C.c_void_p
    L.isync_create.argtypes = [C.c_int]
    L.isync_destroy.argtypes = [C.c_void_p]
    L.isync_destroy.restype = None
    L.isync_set_vector.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.isync_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int),
                               C.POINTER(C.c_double), C.c_void_p]
    _dsp = L
    return L


def dsp_error():
    return (dsp_lib().isee3dsp_last_error() or b"").decode()


class SymDemodEngine:
    """symd_* primitives (symdemod.c loop nests) on the GPU."""

    def __init__(self, max_samples):
        self.L = dsp_lib()
        self.h = self.L.symd_create(int(max_samples))
        if not self.h:
            raise RuntimeError("symd_create failed: " + dsp_error())

    def load(self, samples):
        s = np.ascontiguousarray(samples, dtype=np.int16)
        if self.L.symd_load(self.h, s.ctypes.data, len(s), 0) != 0:
            raise RuntimeError("symd_load: " + dsp_error())

    # the window buffer kept in HBM (symdemod.c:96-125): memmove / append / prefix sums, see include/isee3_dsp_hip.h
    def store_slide(self, slide, nsamples):
        if self.L.symd_store_slide(self.h, int(slide), int(nsamples)) != 0:
            raise RuntimeError("symd_store_slide: " + dsp_error())

    def store_put(self, at, samples):
        if isinstance(samples, DeviceBuffer):
            rc = self.L.symd_store_put(self.h, int(at), samples.ptr, samples.nbytes // 2, 1)
        else:
            s = np.ascontiguousarray(samples, dtype=np.int16)
            rc = self.L.symd_store_put(self.h, int(at), s.ctypes.data, len(s), 0)
        if rc != 0:
            raise RuntimeError("symd_store_put: " + dsp_error())

    def store_scan(self, n):
        if self.L.symd_store_scan(self.h, int(n)) != 0:
            raise RuntimeError("symd_store_scan: " + dsp_error())

    def timesearch(self, lo, sw, symbolclocks, nsymbols, noff):
        sw = np.ascontiguousarray(sw, dtype=np.int32)
        en = np.zeros(noff, dtype=np.float64)
        if self.L.symd_timesearch(self.h, int(lo), sw.ctypes.data_as(C.POINTER(C.c_int)), symbolclocks, nsymbols,
                                  noff, en.ctypes.data_as(C.POINTER(C.c_double))) != 0:
            raise RuntimeError("symd_timesearch: " + dsp_error())
        return en

    def demod(self, edges, symbolclocks, nsymbols, gain):
        edges = np.ascontiguousarray(edges, dtype=np.int32)
        out = np.zeros(nsymbols, dtype=np.uint8)
        e = C.c_double(0)
        if self.L.symd_demod(self.h, edges.ctypes.data_as(C.POINTER(C.c_int)), symbolclocks, nsymbols, gain,
                             out.ctypes.data, 0, C.byref(e)) != 0:
            raise RuntimeError("symd_demod: " + dsp_error())
        return out, e.value

    def close(self):
        if getattr(self, "h", None):
            self.L.symd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PmDemodEngine:
    """pmd_* primitives (pmdemod.c per-sample loops) on the GPU."""

    def __init__(self, fftsize):
        self.L = dsp_lib()
        self.N = int(fftsize)
        self.h = self.L.pmd_create(self.N)
        if not self.h:
            raise RuntimeError("pmd_create failed: " + dsp_error())

    def set_dechirp(self, table):
        t = None if table is None else np.ascontiguousarray(table, dtype=np.complex128)
        if self.L.pmd_set_dechirp(self.h, None if t is None else t.ctypes.data) != 0:
            raise RuntimeError("pmd_set_dechirp: " + dsp_error())

    def load(self, iq_block, flip=False):
        iq = np.ascontiguousarray(iq_block, dtype=np.int16)
        assert len(iq) == 2 * self.N
        if self.L.pmd_load(self.h, iq.ctypes.data, 0, int(flip)) != 0:
            raise RuntimeError("pmd_load: " + dsp_error())

    def fft_peak(self, firstbin=0, lastbin=None):
        pk = PmdPeak()
        if self.L.pmd_fft_peak(self.h, firstbin, self.N if lastbin is None else lastbin, C.byref(pk)) != 0:
            raise RuntimeError("pmd_fft_peak: " + dsp_error())
        return pk

    def last_peak_path(self):
        """0 double transform, 1 single-precision search + exact bins, 2 search, then fallen back to the double transform."""
        return int(self.L.pmd_last_peak_path(self.h))

    def spectrum(self):
        out = np.zeros(self.N, dtype=np.complex128)
        if self.L.pmd_get_spectrum(self.h, out.ctypes.data) != 0:
            raise RuntimeError("pmd_get_spectrum: " + dsp_error())
        return out

    def mix_quantise(self, cstep):
        mx = PmdMix()
        out16 = np.zeros(self.N, dtype=np.int16)
        pre = np.zeros(self.N, dtype=np.float64)
        if self.L.pmd_mix_quantise(self.h, cstep, C.byref(mx), out16.ctypes.data, pre.ctypes.data, 0) != 0:
            raise RuntimeError("pmd_mix_quantise: " + dsp_error())
        return mx, out16, pre

    def close(self):
        if getattr(self, "h", None):
            self.L.pmd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def cli_path(name):
    p = os.path.join(BIN_DIR, name)
    if not os.path.exists(p):
        raise NativeLibraryMissing("%s not built" % p)
    return p


# ------------------------------------------------------------------------------------------------
# libisee3chain.so : pmdemod | symdemod | vdecode on memory buffers (include/isee3_chain.h)
# ------------------------------------------------------------------------------------------------
CHAIN_SYMBOLS = ["isee3_chain_default_opts", "isee3_chain_run_mem", "isee3_chain_run_dev", "isee3_chain_run_fd",
                 "isee3_chain_last_error", "isee3_chain_last_stage_ms", "isee3_chain_release"]


class ChainOpts(C.Structure):
    _fields_ = [("samprate", C.c_double), ("binsize", C.c_double), ("search_freq", C.c_double),
                ("search_width", C.c_double), ("flip", C.c_int), ("symrate", C.c_char_p),
                ("decode_delay", C.c_int), ("verbose", C.c_int)]


_chain = None


def chain_lib():
    global _chain
    if _chain is not None:
        return _chain
    path = lib_path("libisee3chain.so")
    if not os.path.exists(path):
        raise NativeLibraryMissing("%s not built (no CPU fallback)" % path)
    dsp_lib()
    v224_lib()
    L = C.CDLL(path)
    L.isee3_chain_default_opts.argtypes = [C.POINTER(ChainOpts)]
    L.isee3_chain_run_mem.argtypes = [C.POINTER(ChainOpts), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                      C.POINTER(C.c_size_t)]
    L.isee3_chain_run_dev.argtypes = L.isee3_chain_run_mem.argtypes
    L.isee3_chain_last_stage_ms.argtypes = [C.POINTER(C.c_double * 3)]
    L.isee3_chain_last_stage_ms.restype = None
    L.isee3_chain_run_fd.argtypes = [C.POINTER(ChainOpts), C.c_int, C.c_int]
    L.isee3_chain_last_error.restype = C.c_char_p
    L.icesync_sync_vector.argtypes = [C.c_double, C.c_void_p, C.c_int]
    L.icesync_corr_create.restype = C.c_void_p
    L.icesync_corr_create.argtypes = [C.c_double, C.c_double, C.c_int]
    L.icesync_corr_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.icesync_corr_framesamples.restype = C.c_double
    L.icesync_corr_framesamples.argtypes = [C.c_void_p]
    L.icesync_corr_synclen.argtypes = [C.c_void_p]
    L.icesync_corr_destroy.argtypes = [C.c_void_p]
    L.icesync_corr_destroy.restype = None
    _chain = L
    return L


def run_chain(iq, samprate=250000.0, binsize=4.0, symrate="1024", decode_delay=200, flip=False,
              search_freq=0.0, search_width=0.0, stage_ms=None, out_cap=None):
    """int16 interleaved IQ -> decoded bits as bytes of '0'/'1' (whole chain on the GPU).  `iq` is a numpy array in host
    memory (isee3_chain_run_mem) or a DeviceBuffer holding the capture in HBM (isee3_chain_run_dev).  stage_ms: optional
    list that receives the ms spent inside the engine calls of [pmdemod, symdemod, vdecode].  out_cap: size of the output
    buffer handed to the library (default: sized from the capture's duration; tests pass a too small one)."""
    L = chain_lib()
    o = ChainOpts()
    L.isee3_chain_default_opts(C.byref(o))
    o.samprate, o.binsize, o.decode_delay, o.flip = samprate, binsize, decode_delay, int(flip)
    o.search_freq, o.search_width = search_freq, search_width
    o.symrate = None if symrate is None else str(symrate).encode()
    on_dev = isinstance(iq, DeviceBuffer)
    if on_dev:
        nvals, ptr = iq.nbytes // 2, iq.ptr
    else:
        iq = np.ascontiguousarray(iq, dtype=np.int16)
        nvals, ptr = len(iq), iq.ctypes.data
    # decoded bits <= symbols / 2; symbols = duration x the rate `symdemod -c` resolves to (symdemod.c:67-77).  10 % and
    # 4 KiB of head room; the library reports a short write as an error instead of truncating.
    if symrate is None:
        rate = 1024.545058
    else:
        rate = float(symrate) if "." in str(symrate) else float(symrate) * 1024.545058 / 1024.0
    cap = min(nvals // 4 + 4096, int(nvals / 2 / samprate * rate / 2 * 1.1) + 4096)
    if out_cap is not None:
        cap = int(out_cap)
    out = C.create_string_buffer(cap)
    n = C.c_size_t(0)
    fn = L.isee3_chain_run_dev if on_dev else L.isee3_chain_run_mem
    rc = fn(C.byref(o), ptr, nvals // 2, out, cap, C.byref(n))
    if rc != 0:
        raise RuntimeError("isee3_chain_run: " + (L.isee3_chain_last_error() or b"").decode())
    if stage_ms is not None:
        ms = (C.c_double * 3)()
        L.isee3_chain_last_stage_ms(C.byref(ms))
        stage_ms[:] = [ms[0], ms[1], ms[2]]
    return out.raw[:n.value]


ICESYNC_SYMBOLS = ["icesync_sync_vector", "icesync_corr_create", "icesync_corr_search", "icesync_corr_framesamples",
                   "icesync_corr_synclen", "icesync_corr_destroy"]
ICESYNC_FAIL = -1234567890


def icesync_sync_vector(symbolsamples):
    """icesync.c:55-97: the Manchester-coded last 34 symbols of the encoded tail + sync word (host code, no GPU)."""
    cap = int(34 * symbolsamples + 2)
    v = np.zeros(cap, np.float64)
    n = chain_lib().icesync_sync_vector(float(symbolsamples), v.ctypes.data, cap)
    if n < 0:
        raise RuntimeError("icesync_sync_vector failed")
    return v[:n]


class IcesyncCorrelator:
    """generate_sync + fft_sync_search of the reference's icesync.c (:55-208) on the GPU FFT (isync_* in libisee3dsp_hip.so)."""

    def __init__(self, samprate=250000.0, symrate=1024.475, corr_size_log2=0):
        self.L = chain_lib()
        self.h = self.L.icesync_corr_create(float(samprate), float(symrate), int(corr_size_log2))
        if not self.h:
            raise RuntimeError("icesync_corr_create failed: " + dsp_error())
        self.framesamples = self.L.icesync_corr_framesamples(self.h)
        self.synclen = self.L.icesync_corr_synclen(self.h)

    def search(self, samples, low, high):
        s = np.ascontiguousarray(samples, dtype=np.int16)
        assert len(s) >= self.framesamples
        mp = C.c_double(0)
        return self.L.icesync_corr_search(self.h, s.ctypes.data, int(low), int(high), C.byref(mp)), mp.value

    def close(self):
        if getattr(self, "h", None):
            self.L.icesync_corr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def release_chain_objects():
    """isee3_chain_release(): free the decoder / pmdemod / symdemod objects the chain library keeps between calls."""
    chain_lib().isee3_chain_release()
