#!/usr/bin/env python3
"""Mint the golden fixtures under tests/golden/ from the REAL reference.

Run in the build container only (needs /root/reference compiled by oracle/Makefile into
oracle/_ref/):   python tests/golden/make_golden.py [--only NAME]

The reference ships no golden vectors, so the pin is the behaviour of the
reference's own code run here:
  * viterbi_framed.npz   viterbi224_port.c through its C API (ctypes on libv224_port_ref.so):
                         chainback bytes, an FNV-1a hash of EVERY 1 MiB decision row and of the
                         final path metrics (minus their minimum), read out of the reference's
                         own struct v224 (port.c:19-26).  SSE2 outputs ride along to document F1.
  * viterbi_stream.npz   port update(1 bit) + decodebit(200,0) per bit, as vdecode.c:145-152 drives it.
  * vdecode_cli.npz      oracle/_ref/vdecode_port_ref (vdecode.c unmodified) stdout, incl. a forced
                         phase flip, the -p start phase and -F.
  * symdemod_cli.npz     oracle/_ref/symdemod_ref (symdemod.c unmodified) stdout.
  * decode_cli2.npz      the same driver through a noise burst (lock lost) and a 3-symbol slip, also with -n -r 512
  * decode_cli.npz       oracle/_ref/decode_port_ref (decode.c -V unmodified, port decoder) stdout: frame-sync
                         correlator + init(0x819fbe)/update(1024)/chainback per frame.
  * decodeword_sse2.npz  viterbi224_sse2.c's decodeword / min_metric / max_metric (the port has none) on CLEAN
                         coded streams, where the SSE2 and the port decoder take the same survivor path (F1).
  * vdecode_stderr.npz   oracle/_ref/vdecode_port_ref stderr (status lines with the re-encode symbol-error tally,
                         vdecode.c:159-184) for the vdecode_cli.npz inputs, -i 256.
  * pmdemod_oracle.npz   NOT from the reference (FFTW3 absent => pmdemod.c cannot be built): outputs
                         of this repo's restatement, kept only as a regression anchor.  UNPINNED.
Fixtures hold data only: inputs (or the seed + sha256 of a regenerable input) and expected outputs.
"""
import argparse
import ctypes as C
import hashlib
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc  # noqa: E402

NST = 1 << 23


def _fnv(buf):
    a = np.ascontiguousarray(buf)
    return int(orc.lib().orc_fnv1a(a.ctypes.data_as(C.c_void_p), a.nbytes))


def _port_struct_views(handle, length):
    """Views into the reference's struct v224 (viterbi224_port.c:19-26) for hashing its state."""
    base = handle
    off_m1 = 4                                   # int len; then union of uint32 (align 4)
    off_m2 = off_m1 + 4 * NST
    off_dp = (off_m2 + 4 * NST + 7) & ~7         # first pointer member, 8-byte aligned
    ptrs = (C.c_uint64 * 4).from_address(base + off_dp)   # dp, old_metrics, new_metrics, decisions
    dp, old, new, dec = [int(x) for x in ptrs]
    assert old in (base + off_m1, base + off_m2), "struct layout assumption broken"
    metrics = np.ctypeslib.as_array((C.c_uint32 * NST).from_address(old))
    rows = np.ctypeslib.as_array((C.c_uint8 * (length * (NST // 8))).from_address(dec)).reshape(length, NST // 8)
    return metrics, rows, (dp - dec) // (NST // 8)


def framed_case(spec):
    name, syms, nbits, length, start, end = spec
    r = orc.RefV224(length, "port")
    r.init(start)
    r.update(syms, nbits)
    data = r.chainback(nbits, end)
    metrics, rows, dp = _port_struct_views(r.h, length)
    rowhash = np.array([_fnv(rows[i]) for i in range(min(nbits, length))], dtype=np.uint64)
    mn = int(metrics.min())
    rel = (metrics - np.uint32(mn)).astype(np.uint32)
    out = dict(syms=syms, nbits=nbits, length=length, start=start, end=end, port_data=data,
               port_rowhash=rowhash, port_metric_hash=np.uint64(_fnv(rel)),
               port_spread=np.uint32(int(metrics.max()) - mn), port_dp=np.int32(dp))
    # best-state traceback (decodebit with endstate < 0, port.c:113-122)
    out["port_decodebit_best"] = np.array([r.decodebit(d, -1) for d in (1, 24, min(nbits, length))], dtype=np.int8)
    r.close()
    s = orc.RefV224(length, "sse2")
    s.init(start)
    s.update(syms, nbits)
    out["sse2_data"] = s.chainback(nbits, end)
    s.close()
    return name, out


def make_framed():
    L = orc.lib()
    specs = []
    specs.append(("uniform256", orc.gen_uniform(101, 512), 256, 256, 0, 0))
    specs.append(("erasure128", np.full(256, 128, np.uint8), 128, 128, 0, 0))
    s, _ = orc.gen_coded_frame(103, 512, 2.0)
    specs.append(("coded2dB_512", s, 512, 512, 0, 0))
    s, _ = orc.gen_coded_frame(104, 1024, 4.0)
    specs.append(("coded4dB_1024", s, 1024, 1024, 0, 0))
    # non-zero start and end state: encode from a non-zero encoder state, no tail
    data = orc.gen_uniform(105, 32)
    st0 = 0x5A5A5A & 0x7FFFFF
    sy, st1 = orc.encode(data, st0)
    rng = np.random.default_rng(105)
    noisy = np.clip(128 + 24 * (2 * sy.astype(np.int32) - 1) + np.rint(rng.normal(0, 20, sy.shape)), 0, 255).astype(np.uint8)
    specs.append(("startend256", noisy, 256, 256, st0, st1 & 0x7FFFFF))
    # saturated symbols (0/255 only) -> many exact ties between the two c values
    hard = (orc.gen_uniform(106, 384) & 1) * 255
    specs.append(("hard192", hard.astype(np.uint8), 192, 192, 0, 0))
    # decision ring shorter than the block: rows wrap (port.c:187-188) and chainback uses n % len
    specs.append(("wrap_len50", orc.gen_uniform(107, 240), 120, 50, 0, 0))
    with mp.Pool(min(len(specs), 7)) as pool:
        res = pool.map(framed_case, specs)
    flat = {}
    for name, d in res:
        for k, v in d.items():
            flat["%s/%s" % (name, k)] = v
    flat["names"] = np.array([n for n, _ in res])
    np.savez_compressed(os.path.join(HERE, "viterbi_framed.npz"), **flat)
    print("viterbi_framed.npz:", [n for n, _ in res])


def make_stream():
    nsym = 20000
    syms, bits = orc.gen_coded_stream(201, nsym // 2, 3.0, 24.0, 5)
    delay, length = 200, 201
    r = orc.RefV224(length, "port")
    r.init(0)
    out = []
    for u in range(nsym // 2):
        r.update(syms[2 * u:2 * u + 2], 1)
        if u >= delay:
            out.append(r.decodebit(delay, 0))
        if u % 1000 == 0:
            print("stream", u, flush=True)
    best = np.array([r.decodebit(d, -1) for d in (1, 100, 200)], dtype=np.int8)
    r.close()
    np.savez_compressed(os.path.join(HERE, "viterbi_stream.npz"), syms=syms, sent=bits, delay=delay,
                        length=length, port_bits=np.packbits(np.array(out, np.uint8)), nout=len(out),
                        port_decodebit_best_final=best)
    print("viterbi_stream.npz:", len(out), "bits")


def _tlm_symbols(seed, nbits, ebn0):
    """Telemetry-like symbol stream (frames with sync word), via the baseband source's encoder."""
    # reuse orc_gen_baseband's symbol source at 2 samples/symbol, noise-free, to obtain hard symbols
    bb, sent = orc.gen_baseband(seed, 2.0, nbits * 2.0, 1.0, 1000.0, 0.0)
    hard = (bb[1::2] > 0).astype(np.int32)                       # second half-sample carries the sign
    rng = np.random.default_rng(seed)
    sigma = 24.0 * np.sqrt(0.5) / 10 ** (0.05 * (ebn0 + 10 * np.log10(0.5)))
    sy = np.clip(np.rint(128 + 24 * (2 * hard - 1) + rng.normal(0, sigma, hard.shape)), 0, 255).astype(np.uint8)
    return sy, sent


def vdecode_case(spec):
    name, args, syms = spec
    out = orc.ref_cli("vdecode_port_ref", ["-q"] + args, syms.tobytes())
    return name, args, syms, out


def make_vdecode():
    specs = []
    sy, _ = _tlm_symbols(301, 3 * 1024 + 300, 5.0)
    # drop one symbol after ~1.2 frames: decoder is then out of phase and must flip (vdecode.c:126-139)
    flipped = np.concatenate([sy[:2501], sy[2502:]])
    specs.append(("flip", [], flipped))
    specs.append(("startphase_p", ["-p"], sy[1:2 * 700 + 1]))
    specs.append(("forced_F", ["-F"], sy[:2 * 600]))
    specs.append(("delay64", ["-d", "64", "-F"], sy[:2 * 500]))
    specs.append(("delay_too_small", ["-d", "10", "-F"], sy[:2 * 300]))
    with mp.Pool(len(specs)) as pool:
        res = pool.map(vdecode_case, specs)
    flat = {"names": np.array([r[0] for r in res])}
    for name, args, syms, out in res:
        flat[name + "/args"] = np.array(args if args else [""], dtype="U8")
        flat[name + "/syms"] = syms
        flat[name + "/stdout"] = np.frombuffer(out, dtype=np.uint8)
        print("vdecode", name, len(out), "bits")
    np.savez_compressed(os.path.join(HERE, "vdecode_cli.npz"), **flat)


def make_decode():
    """the reference's framed driver decode.c (Fano off: -V) on the port decoder."""
    sy, _ = _tlm_symbols(601, 4 * 1024 + 200, 5.0)
    out = orc.ref_cli("decode_port_ref", ["-V"], sy.tobytes())
    np.savez_compressed(os.path.join(HERE, "decode_cli.npz"), syms=sy, stdout=np.frombuffer(out, dtype=np.uint8),
                        args=np.array(["-V"]))
    print("decode", len(sy), "symbols ->", len(out), "bytes of frame dump")
    print(out.decode(errors="replace")[:600])


def _decode_case(spec):
    name, args, syms = spec
    return name, args, syms, orc.ref_cli("decode_port_ref", args, syms.tobytes())


def make_decode2():
    """More of decode.c -V: a noise burst (bad frame, lock lost, the correlator searches again) and a symbol
    slip (the next sync is found at a new offset), with and without -n / -r."""
    sy, _ = _tlm_symbols(611, 9 * 1024 + 300, 5.0)
    rng = np.random.default_rng(612)
    hurt = sy.copy()
    hurt[3 * 2048 + 700:4 * 2048 + 300] = rng.integers(0, 256, 2048 - 400, dtype=np.uint8)   # burst inside frames 4/5
    hurt = np.concatenate([hurt[:6 * 2048 + 911], hurt[6 * 2048 + 914:]])                     # three symbols lost
    specs = [("slip_noise", ["-V"], hurt), ("slip_noise_nobad_r512", ["-V", "-n", "-r", "512"], hurt)]
    with mp.Pool(len(specs)) as pool:
        res = pool.map(_decode_case, specs)
    flat = {"names": np.array([r[0] for r in res])}
    for name, args, syms, out in res:
        flat[name + "/args"] = np.array(args, dtype="U8")
        flat[name + "/syms"] = syms
        flat[name + "/stdout"] = np.frombuffer(out, dtype=np.uint8)
        print("decode2", name, len(syms), "symbols ->", len(out), "bytes")
        print(out.decode(errors="replace")[:300])
    np.savez_compressed(os.path.join(HERE, "decode_cli2.npz"), **flat)


def make_symdemod():
    cases = [
        # name, args, generator kwargs
        ("r25k_w05", ["-q", "-r", "25000", "-c", "1024", "-w", "0.5"],
         dict(seed=401, samprate=25000.0, seconds=3.3, amp=1500.0, noise_sigma=3000.0)),
        ("r250k_default", ["-q"],
         dict(seed=402, samprate=250000.0, seconds=3.2, amp=800.0, noise_sigma=6000.0)),
        ("r25k_track", ["-q", "-r", "25000", "-c", "1024", "-w", "0.5", "-t"],
         dict(seed=403, samprate=25000.0, seconds=2.7, amp=1500.0, noise_sigma=2500.0)),
        ("r16k_64bps_C8", ["-q", "-r", "16000", "-c", "128", "-w", "2"],
         dict(seed=404, samprate=16000.0, seconds=9.0, symrate=128.0 * 1024.545058 / 1024.0, amp=1200.0,
              noise_sigma=3000.0)),
        ("r25k_exact_clock", ["-q", "-r", "25000", "-c", "1000.5", "-w", "0.5"],
         dict(seed=405, samprate=25000.0, seconds=2.6, symrate=1000.5, amp=2000.0, noise_sigma=1000.0)),
        ("r25k_clip", ["-q", "-r", "25000", "-c", "1024", "-w", "0.5"],
         dict(seed=406, samprate=25000.0, seconds=2.2, amp=30000.0, noise_sigma=20000.0)),
    ]
    flat = {"names": np.array([c[0] for c in cases])}
    for name, args, kw in cases:
        seed = kw.pop("seed")
        bb, _ = orc.gen_baseband(seed, **kw)
        out = orc.ref_cli("symdemod_ref", args, bb.tobytes())
        flat[name + "/args"] = np.array(args, dtype="U16")
        flat[name + "/gen"] = np.array([seed, kw["samprate"], kw["seconds"], kw.get("symrate", 1024.545058),
                                        kw["amp"], kw["noise_sigma"]], dtype=np.float64)
        flat[name + "/in_sha256"] = np.array(hashlib.sha256(bb.tobytes()).hexdigest())
        if bb.nbytes <= 200_000:
            flat[name + "/in"] = bb
        flat[name + "/stdout"] = np.frombuffer(out, dtype=np.uint8)
        print("symdemod", name, len(bb), "samples ->", len(out), "symbols")
    np.savez_compressed(os.path.join(HERE, "symdemod_cli.npz"), **flat)


def make_pmdemod():
    flat = {}
    cases = [("b4_r16384", dict(samprate=16384.0, binsize=4.0), dict(seed=501, seconds=1.3, fc_hz=1234.567)),
             ("b4_r16384_neg_flip", dict(samprate=16384.0, binsize=4.0, flip=True), dict(seed=502, seconds=0.8, fc_hz=-3000.25)),
             ("b4_r16384_doppler", dict(samprate=16384.0, binsize=4.0, doppler_rate=35.0), dict(seed=503, seconds=0.8, fc_hz=777.7)),
             ("b4_r16384_W", dict(samprate=16384.0, binsize=4.0, search_width=200.0, search_freq=1000.0), dict(seed=504, seconds=1.1, fc_hz=1100.0))]
    for name, cfg, gen in cases:
        iq, _ = orc.gen_iq(gen["seed"], cfg["samprate"], gen["seconds"], fc_hz=gen["fc_hz"], amp=3000.0,
                           cn0_dbhz=50.0, symrate=1024.545058)
        out, pre, rep, N = orc.pmdemod(iq, **cfg)
        flat[name + "/iq"] = iq
        flat[name + "/cfg"] = np.array([cfg.get("samprate"), cfg.get("binsize"), cfg.get("search_freq", 0.0),
                                        cfg.get("search_width", 0.0), cfg.get("doppler_rate", 0.0), 21.0,
                                        float(cfg.get("flip", False))])
        flat[name + "/out"] = out
        flat[name + "/pre"] = pre
        flat[name + "/peak"] = np.array([r["peak"] for r in rep], np.int32)
        flat[name + "/carrier_freq"] = np.array([r["carrier_freq"] for r in rep])
        flat[name + "/cn0"] = np.array([r["cn0"] for r in rep])
        print("pmdemod", name, "N", N, "blocks", len(rep), [round(r["carrier_freq"], 3) for r in rep])
    flat["names"] = np.array([c[0] for c in cases])
    np.savez_compressed(os.path.join(HERE, "pmdemod_oracle.npz"), **flat)


def make_decodeword():
    """a10: decodeword exists only in viterbi224_sse2.c.  Clean coded streams (Eb/N0 7 dB: no channel error reaches the
    survivor), several delays and end states incl. the best-state search (end < 0)."""
    flat = {"names": np.array(["clean_len300", "clean_wrap_len200"])}
    for name, seed, nbits, length in (("clean_len300", 701, 300, 300), ("clean_wrap_len200", 702, 460, 200)):
        syms, sent = orc.gen_coded_stream(seed, nbits, 7.0, 24.0, 0)
        r = orc.RefV224(length, "sse2")
        r.init(0)
        r.update(syms, nbits)
        q = [(64, 0), (64, -1), (17, -1), (1, 0), (100, -1), (min(length, nbits), -1), (64, 0x2aaaaa)]
        words = np.array([r.decodeword(d, e) for d, e in q], dtype=np.uint64)
        r.close()
        flat[name + "/syms"], flat[name + "/sent"] = syms, sent
        flat[name + "/nbits"], flat[name + "/length"] = nbits, length
        flat[name + "/queries"] = np.array(q, dtype=np.int64)
        flat[name + "/sse2_words"] = words
        print("decodeword", name, [hex(int(w)) for w in words])
    np.savez_compressed(os.path.join(HERE, "decodeword_sse2.npz"), **flat)


def _vdecode_err_case(spec):
    name, args, syms = spec
    exe = os.path.join(orc.REF_DIR, "vdecode_port_ref")
    import subprocess
    p = subprocess.run([exe] + args, input=syms.tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, LANG="C", LC_ALL="C"), check=True, timeout=3600)
    return name, args, p.stdout, p.stderr


def make_vdecode_stderr():
    z = np.load(os.path.join(HERE, "vdecode_cli.npz"))
    specs = []
    for name in [str(n) for n in z["names"]]:
        args = [a for a in z[name + "/args"] if a] + ["-i", "256"]
        specs.append((name, args, z[name + "/syms"]))
    with mp.Pool(len(specs)) as pool:
        res = pool.map(_vdecode_err_case, specs)
    flat = {"names": np.array([r[0] for r in res])}
    for name, args, out, err in res:
        assert out == z[name + "/stdout"].tobytes(), "stdout moved?"
        flat[name + "/args"] = np.array(args, dtype="U8")
        flat[name + "/stderr"] = np.frombuffer(err, dtype=np.uint8)
        print("vdecode stderr", name)
        print(err.decode())
    np.savez_compressed(os.path.join(HERE, "vdecode_stderr.npz"), **flat)


ALL = dict(decodeword=make_decodeword, vdecode_stderr=make_vdecode_stderr, framed=make_framed, stream=make_stream, vdecode=make_vdecode, symdemod=make_symdemod,
           pmdemod=make_pmdemod, decode=make_decode, decode2=make_decode2)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", choices=sorted(ALL), action="append")
    a = ap.parse_args()
    assert orc.have_ref(), "oracle/_ref missing: run `make -C oracle ref` in the build container"
    for k in (a.only or list(ALL)):
        ALL[k]()
