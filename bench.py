#!/usr/bin/env python3
"""bench.py -- headline benchmark of the ISEE-3 receive-chain hot path on MI355X.

BASELINE.json's metric has two halves; the ONE JSON line rank 0 prints carries both, and the stress configuration:

  top level  "Viterbi K=24 Msymbols/s" on BASELINE configs[1]: one 1e7-symbol synthetic soft-symbol stream per GPU,
             decoded with the semantics of `vdecode -d 200` (one trellis step + decodebit(200,0) per bit).  A "step" =
             one full pass of that stream (init + 5e6 trellis steps + 5e6 tracebacks) on two decoders (verified split).
             `value`: symbols resident in HBM when the timed region starts, bits left there; config.host_buffers: one
             more step from host symbols to host bits (SURVEY 8(d) config 2's region, PCIe inclusive).
  "frames"   BASELINE configs[0]'s shape: 50 frames x 1 000 bits as one v224hip_decode_frames batch.
  "chain"    "end-to-end IQ Msamples/s" on BASELINE configs[2] / [3]: 60 s x 250 kS/s synthetic int16 IQ per GPU through
             pmdemod | symdemod | vdecode in one process (libisee3chain.so), capture resident in HBM
             (isee3_chain_run_dev); the rate with the capture in host memory (PCIe included) rides along.
  "stress"   BASELINE configs[4]: 10 MS/s, 1 Hz bins (N = 2^23), ONE capture of --stress-blocks (128) blocks cut into 64
             overlapped segments dealt to the ranks (--segment-concurrency = 2 chains at a time per GPU), stitched on rank 0
             with every seam verified (strong scaling).

N > 1: `python bench.py --gpus N` starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child process
(before anything here touches a device); one rank per GPU, RCCL only for the barrier / MAX / SUM / the gather of decoded
bits: every rank works on its own independent stream and capture -- segments shard one per GPU, no collective in the
data path -> weak scaling (the stress record: strong).

  roofline      the dominant kernel k_acs_lds15 over the timed region (two decoders side by side) with the lone launch
                under single_decoder.  It is VALU-issue bound (15 trellis steps share ONE pass over the path metrics):
                `frac` = VALU issue time / time per launch <= 1; `hbm` = PMC bytes per launch / time per launch against
                8 TB/s; the SURVEY 8(d) algorithmic figure (34 603 008 B per trellis step) is `algorithmic_x_peak`.
                Time per launch: HIP events on decoder 0's own stream, live in this run.  chain.roofline / stress.roofline:
                physical HBM bytes of all the chain's kernels (committed PMC passes) / step time against 8 TB/s.
  cpu_baseline  the reference's SSE2 decoder (oracle/_ref, built from /root/reference in the build container) on ONE
                pinned host core, bounded sample, with the host's CPU model / nproc / build flags; chain.cpu_baseline:
                oracle pmdemod -> reference symdemod -> reference vdecode (SSE2) on the first seconds of the same capture.
"""
import argparse
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
ALG_BYTES_PER_STEP = (1 << 23) * 2 * 2 + (1 << 23) // 8          # read + write u16 metrics + decisions
HBM_PEAK_GBS = 8000.0
CLOCK_GHZ = 2.4                                                 # MI355X max shader clock (MI355X_MICROARCH.md)
N_SIMD = 256 * 4
WAVES_PER_LAUNCH = 256 * 16                                     # k_acs_lds15: 256 workgroups x 1024 threads


def load_pkg():
    name = "isee3_decoder_amd"
    if name in sys.modules:
        return sys.modules[name]
    pkgdir = os.path.join(ROOT, "isee3-decoder_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(pkgdir, "__init__.py"),
                                                  submodule_search_locations=[pkgdir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _orc():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    return orc


def host_info():
    """Where the CPU baselines ran: CPU model, logical CPUs of the box / of this process, and how the reference objects
    under oracle/_ref were compiled (oracle/Makefile writes BUILD_FLAGS.txt beside them)."""
    model, flags = "unknown", set()
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("flags") and not flags:
                flags = set(line.split(":", 1)[1].split())
    except OSError:
        pass
    info = {"cpu_model": model, "nproc": os.cpu_count(), "nproc_allowed": len(os.sched_getaffinity(0)),
            "avx2": "avx2" in flags}
    bf = os.path.join(ROOT, "oracle", "_ref", "BUILD_FLAGS.txt")
    if os.path.exists(bf):
        info["ref_build"] = dict(l.strip().split(": ", 1) for l in open(bf) if ": " in l)
    return info


class pinned_to_one_core:
    """The CPU legs run on ONE core: the process is pinned (sched_setaffinity, no taskset hop) to the first core it is
    allowed on for the duration of the leg."""

    def __enter__(self):
        self.before = os.sched_getaffinity(0)
        self.core = min(self.before)
        try:
            os.sched_setaffinity(0, {self.core})
            self.ok = True
        except OSError:
            self.ok = False
        return self

    def __exit__(self, *exc):
        try:
            os.sched_setaffinity(0, self.before)
        except OSError:
            pass


def _time_ref_update(d, syms, nbits_sample):
    d.init(0)
    d.update(syms[:512], 256)                     # touch memory
    d.init(0)
    t0 = time.perf_counter()
    done = 0
    while done < nbits_sample:
        n = min(256, nbits_sample - done)
        d.update(syms[2 * done:2 * (done + n)], n)
        done += n
    return time.perf_counter() - t0


def cpu_baseline(nbits_sample):
    """Reference SSE2 decoder on one host core (falls back to the oracle port restatement).  Two builds of the same
    reference source are timed where the host can run both -- plain x86-64 (-msse2) and x86-64-v3, the portable
    stand-in for the reference's -march=native -- and the FASTER one is the baseline."""
    orc = _orc()
    rng = np.random.default_rng(1)
    syms = rng.integers(0, 256, 2 * nbits_sample, dtype=np.uint8)
    host = host_info()
    with pinned_to_one_core() as pin:
        builds = {}
        if orc.have_ref():
            variants = ["sse2"]
            if host["avx2"] and os.path.exists(os.path.join(orc.REF_DIR, "libv224_sse2_v3_ref.so")):
                variants.append("sse2_v3")
            for var in variants:
                d = orc.RefV224(256, var)
                n = nbits_sample if var == "sse2" else max(512, nbits_sample // 2)
                dt = _time_ref_update(d, syms, n)
                d.close()
                builds[var] = {"Msymbols_per_s": round(2 * n / dt / 1e6, 6), "steps": n, "seconds": round(dt, 2)}
            best = max(builds, key=lambda v: builds[v]["Msymbols_per_s"])
            kind, what = "reference", "viterbi224_sse2.c (oracle/_ref, build %s), update_viterbi224_blk" % best
            value, nb, dt = builds[best]["Msymbols_per_s"], builds[best]["steps"], builds[best]["seconds"]
        else:
            os.environ["OMP_NUM_THREADS"] = "1"
            d = orc.OracleV224(256, orc.FAST)
            dt = _time_ref_update(d, syms, nbits_sample)
            d.close()
            kind, what = "port", "oracle FAST engine (port semantics, SSE2), 1 thread"
            value, nb = round(2 * nbits_sample / dt / 1e6, 6), nbits_sample
        res = {"value": value, "unit": "Msymbols/s", "cores": 1, "kind": kind,
               "sample": "%d trellis steps of uniform-random symbols, %s, %.1f s" % (nb, what, dt),
               "pinned": {"sched_setaffinity": pin.ok, "core": pin.core}, "host": host}
        if builds:
            res["builds"] = builds
        if orc.have_ref():          # the parity target itself, for the record (not the x100 denominator)
            p = orc.RefV224(64, "port")
            p.init(0)
            n = 320
            t0 = time.perf_counter()
            for i in range(0, n, 64):
                p.update(syms[2 * i:2 * (i + 64)], 64)
            res["port_value"] = round(2 * n / (time.perf_counter() - t0) / 1e6, 6)
            res["port_sample"] = "%d trellis steps, viterbi224_port.c (oracle/_ref), 1 core" % n
            p.close()
    return res


def chain_cpu_baseline(iq, fs, binsize, seconds):
    """The reference chain on ONE host core over the first `seconds` of the capture: pmdemod = the oracle restatement
    (pmdemod.c needs FFTW3, absent: not buildable), symdemod and vdecode (SSE2 decoder) = the reference's own binaries
    where oracle/_ref travelled along, else their oracle restatements."""
    orc = _orc()
    os.environ["OMP_NUM_THREADS"] = "1"
    N = 1 << int(np.rint(np.log2(fs / binsize)))
    nsamp = int(seconds * fs) // N * N
    part = np.ascontiguousarray(iq[:2 * nsamp])
    with pinned_to_one_core() as pin:                 # children (the reference's pipe stages) inherit the one-core mask
        rec = _chain_cpu_legs(orc, part, nsamp, fs, binsize)
    rec["pinned"] = {"sched_setaffinity": pin.ok, "core": pin.core}
    rec["host"] = host_info()
    return rec


def _chain_cpu_legs(orc, part, nsamp, fs, binsize):
    t0 = time.perf_counter()
    bb, _, _, _ = orc.pmdemod(part, samprate=fs, binsize=binsize, want_pre=False)
    t1 = time.perf_counter()
    if orc.have_ref():
        sy = orc.ref_cli("symdemod_ref", ["-q", "-r", str(int(fs)), "-c", "1024"], bb.tobytes())
        t2 = time.perf_counter()
        bits = orc.ref_cli("vdecode_sse2_ref", ["-q"], sy)
        kind = "reference"
        what = "oracle pmdemod (pmdemod.c unbuildable: FFTW3) | reference symdemod.c | reference vdecode.c + viterbi224_sse2.c"
    else:
        sy, _, _ = orc.symdemod(bb, samprate=int(fs), c_opt="1024")
        sy = sy.tobytes()
        t2 = time.perf_counter()
        bits, _ = orc.vdecode(np.frombuffer(sy, np.uint8))
        kind, what = "port", "oracle pmdemod | oracle symdemod | oracle vdecode (port semantics, 1 thread)"
    t3 = time.perf_counter()
    return {"value": round(nsamp / (t3 - t0) / 1e6, 6), "unit": "Msamples/s", "cores": 1, "kind": kind,
            "sample": "first %.1f s of the capture (%d samples), %s: pmdemod %.2f s, symdemod %.2f s, vdecode %.2f s, %d bits"
                      % (nsamp / fs, nsamp, what, t1 - t0, t2 - t1, t3 - t2, len(bits))}


def pmc_constants(kern):
    """PMC passes cannot share a run with the timing: the per-launch HBM bytes and VALU instruction count of the kernel
    are committed constants (profiles/), refreshed whenever the kernel changes."""
    for tname in ("r03_pmc_lds15.json", "r02_pmc_lds15.json", "r01c_pmc_traffic.json", "r01_pmc_traffic.json"):
        tpath = os.path.join(ROOT, "profiles", tname)
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("kernel") == kern:
                return tj, "profiles/" + tname
    return {}, None


def chain_check(bits, sent):
    got = np.frombuffer(bits, np.uint8) - ord("0")
    s = "".join(map(str, sent))
    # the tail: vdecode may need one 2048-symbol frame to settle its symbol-pair phase (vdecode.c:126-139)
    return len(got), bool(len(got) > 2500 and "".join(map(str, got[-1100:-100])) in s)


def chain_roofline(fs, binsize, nsamp_step, dt_step, stage_ms=None):
    """HBM roofline of the whole chain: physical bytes per IQ sample and stage from the committed PMC passes
    (profiles/r03_pmc_chain.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, FETCH x 2 on gfx950 as the
    guide prescribes, summed by kernel and divided by the samples those runs processed) x the samples of this step /
    this step's wall time, against the 8 TB/s peak; SURVEY 8(d)'s algorithmic 8 + (symrate / fs) x 17 301 504 B per sample
    beside it.  None when no PMC record matches this sample rate / bin size."""
    path = os.path.join(ROOT, "profiles", "r03_pmc_chain.json")
    if not os.path.exists(path):
        return None
    for e in json.load(open(path)).get("configs", []):
        if abs(e["samprate"] - fs) < 0.5 and abs(e["binsize"] - binsize) < 1e-9:
            break
    else:
        return None
    bps = e["hbm_bytes_per_sample"]
    total = float(sum(bps.values()))
    traffic = total * nsamp_step
    alg = 8.0 + 1024.545058 / fs * 17301504.0
    r = {"bound": "hbm", "achieved": round(traffic / dt_step / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(traffic / dt_step / 1e9 / HBM_PEAK_GBS, 4), "traffic": int(traffic),
         "traffic_unit": "HBM bytes per step, all kernels of the chain: PMC bytes per sample (%s) x %d samples" % (e["source"], nsamp_step),
         "hbm_bytes_per_sample": {k: round(v, 1) for k, v in bps.items()},
         "algorithmic_bytes_per_sample": round(alg, 1),
         "algorithmic_GBps": round(alg * nsamp_step / dt_step / 1e9, 1),
         "physical_over_algorithmic": round(total / alg, 3)}
    if stage_ms:
        r["stages"] = {st: {"hbm_bytes": int(bps.get(st, 0.0) * nsamp_step), "engine_ms": round(ms, 3),
                            "GBps_over_engine_time": round(bps.get(st, 0.0) * nsamp_step / (ms * 1e-3) / 1e9, 1) if ms > 0 else None}
                       for st, ms in zip(("pmdemod", "symdemod", "viterbi"), stage_ms)}
        r["stages"]["what"] = ("engine_ms = time the stage's host thread spent inside engine calls in the last timed step (waits "
                               "included; the three stages run at the same time on the one GPU, so their rates do not add up to "
                               "`achieved`)")
    return r


def chain_record(a, ctx, seconds, rate, binsize, steps, warmup, with_cpu):
    """BASELINE configs[2] / [3]: the whole chain on one independent capture per rank.  Timed with the capture resident
    in HBM (the contract's `value`) and again with the capture in host memory (PCIe inclusive)."""
    pkg, harness, synth, dist, torch = ctx["pkg"], ctx["harness"], ctx["synth"], ctx["dist"], ctx["torch"]
    rank, world, redev = ctx["rank"], ctx["world"], ctx["redev"]
    fs = float(rate)
    iq, sent = synth.iq_capture(3 + rank, fs, seconds, amp=None)
    d_iq = pkg.DeviceBuffer.from_numpy(iq)
    out, ms = {}, []

    def step_dev():
        out["bits"] = pkg.run_chain(d_iq, samprate=fs, binsize=binsize, symrate="1024", decode_delay=a.delay, stage_ms=ms)

    def step_host():
        out["hbits"] = pkg.run_chain(iq, samprate=fs, binsize=binsize, symrate="1024", decode_delay=a.delay)

    fence = harness.make_fence(dist if world > 1 else None, torch.cuda.synchronize)
    dt_local = harness.timed_steps(step_dev, steps, warmup, fence)
    dt = harness.max_over_ranks(dist if world > 1 else None, torch, dt_local, redev)
    hsteps = max(1, min(steps, 3))
    dth = harness.max_over_ranks(dist if world > 1 else None, torch, harness.timed_steps(step_host, hsteps, 1, fence), redev)
    per_rank = harness.gather_per_rank(dist if world > 1 else None, round(dt_local / steps * 1e3, 3))
    nbits, ok = chain_check(out["bits"], sent)
    d_iq.free()
    if rank != 0:
        return None
    nsamp = len(iq) // 2
    rec = {"metric": "end-to-end IQ Msamples/s", "value": round(nsamp * world * steps / dt / 1e6, 3), "unit": "Msamples/s",
           "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3), "ms_per_step_per_rank": per_rank,
           "workload": "pmdemod|symdemod|vdecode in one process on %g s of %g kS/s int16 IQ, %g Hz bins (N = 2^%d), 1024 sym/s "
                       "Manchester, vdecode -d %d, one independent capture per GPU; capture resident in HBM, sample streams "
                       "between the stages stay in HBM" % (seconds, fs / 1e3, binsize, int(np.rint(np.log2(fs / binsize))), a.delay),
           "decoded_bits": nbits,
           "host_capture": {"value": round(nsamp * world * hsteps / dth / 1e6, 3), "ms_per_step": round(dth / hsteps * 1e3, 3),
                            "what": "the same with the capture in pageable host memory (isee3_chain_run_mem): PCIe H2D of 4 B "
                                    "per sample inside the time", "identical_output": bool(out["hbits"] == out["bits"])},
           "stage_engine_ms": {"pmdemod": round(ms[0], 3), "symdemod": round(ms[1], 3), "vdecode": round(ms[2], 3),
                               "what": "ms the last timed step spent inside the engine calls of each stage (they include "
                                       "waiting for the GPU; the three stages run concurrently)"},
           "algorithmic_bytes_per_sample": {"pmdemod": 6, "symdemod": 2,
                                            "viterbi": round(1024.545058 / fs * 17301504, 1)},
           "check": {"decoded_run_found_in_sent_stream": ok}}
    rec["roofline"] = chain_roofline(fs, binsize, nsamp * world, dt / steps, ms)
    if with_cpu:
        rec["cpu_baseline"] = chain_cpu_baseline(iq, fs, binsize, a.chain_cpu_seconds)
        rec["speedup_vs_cpu_1core"] = round(rec["value"] / rec["cpu_baseline"]["value"], 1)
    return rec


def frames_record(pkg, synth, nframes=50, framebits=1000):
    """configs[0]'s shape -- vtest224's init / update(framebits) / chainback per frame (vtest224.c:116-118, 170-176), 50 frames
    of 1 000 bits -- as one v224hip_decode_frames batch on two and on three decoder objects (what `decode -V` uses)."""
    frames, sent = [], []
    for f in range(nframes):
        syms, bits, _ = synth.coded_stream(5000 + f, framebits, 3.0, 24.0, 0.0)
        frames.append(syms[:2 * framebits]); sent.append(bits[:framebits])
    syms = np.concatenate(frames)
    rows = 2 * ((framebits + 14) // 15 * 15)
    decs = [pkg.Viterbi224(rows) for _ in range(3)]
    rec = {"what": "%d independent frames of %d bits (BASELINE configs[0] shape), v224hip_decode_frames, host buffers in and "
                   "out (H2D of the symbols and D2H of the bytes inside the time)" % (nframes, framebits)}
    out = None
    for nd in (2, 3):
        pkg.decode_frames(decs[:nd], syms, nframes, framebits)          # warm-up at full size: the staging buffers are grown once
        t0 = time.perf_counter()
        got = pkg.decode_frames(decs[:nd], syms, nframes, framebits)
        dt = time.perf_counter() - t0
        rec["decoders_%d" % nd] = {"ms_per_frame": round(dt * 1e3 / nframes, 4), "Msymbols_per_s": round(2 * framebits * nframes / dt / 1e6, 4)}
        rec["identical_output"] = bool(out is None or np.array_equal(out, got)) and rec.get("identical_output", True)
        out = got
    # decoded data = sent data away from the unterminated frame end (the oracle comparison lives in tests/)
    dec, ref = np.unpackbits(out, axis=1)[:, :framebits], np.stack(sent)
    n = framebits - 150                                  # (the frames are not terminated: the last bits hang on the end state)
    rec["ber_away_from_frame_end"] = min(float(np.mean(dec[:, sh:sh + n] != ref[:, :n])) for sh in range(0, 25))
    for d in decs:
        d.close()
    return rec


def chain_workload(a, ctx):
    """`--workload chain`: the chain line alone (profiling), or -- with --chain-segments S > 1 -- BASELINE configs[4]:
    ONE capture cut into S overlapped block-aligned segments, segment g -> rank g mod world, two chains at a time per
    GPU, parts stitched on rank 0 (strong scaling)."""
    pkg, harness, synth, dist, torch = ctx["pkg"], ctx["harness"], ctx["synth"], ctx["dist"], ctx["torch"]
    rank, world, redev = ctx["rank"], ctx["world"], ctx["redev"]
    if a.chain_segments <= 1:
        rec = chain_record(a, ctx, a.chain_seconds, a.chain_rate, a.chain_bin, a.steps, a.warmup, not a.no_cpu and world == 1)
        if rank == 0:
            rec.update({"n_gpus": world, "ranks_seen": ctx["ranks_seen"], "higher_is_better": True, "scaling": "weak",
                        "vs_baseline": None, "dtype": "f64+u16", "data": "synthetic", "config": {"workload": rec.pop("workload")}})
            print(json.dumps(rec), flush=True)
        return
    fs = float(a.chain_rate)
    iq, sent = synth.iq_capture(3, fs, a.chain_seconds, amp=None)        # the same capture on every rank
    rec = segmented_record(a, ctx, iq, sent, fs, a.chain_bin, a.chain_segments, a.chain_warm_blocks, a.steps, a.warmup)
    if rank == 0:
        rec.update({"n_gpus": world, "ranks_seen": ctx["ranks_seen"], "higher_is_better": True, "vs_baseline": None,
                    "dtype": "f64+u16", "data": "synthetic"})
        print(json.dumps(rec), flush=True)


def segmented_record(a, ctx, iq, sent, fs, binsize, nseg, warm_blocks, steps, warmup):
    """BASELINE configs[4]: ONE capture (the same array on every rank, resident in HBM) cut into nseg overlapped
    block-aligned segments, segment g -> rank g mod world, two chains at a time per GPU, the decoded parts all-gathered
    (a few kB: the only exchange) and stitched on rank 0 with every seam verified.  Strong scaling: the capture is fixed."""
    pkg, harness, dist, torch = ctx["pkg"], ctx["harness"], ctx["dist"], ctx["torch"]
    rank, world, redev = ctx["rank"], ctx["world"], ctx["redev"]
    from importlib import import_module
    segmod = import_module("isee3_decoder_amd.segment")
    N = 1 << int(np.rint(np.log2(fs / binsize)))
    nblocks = (len(iq) // 2) // N
    plan = segmod.plan_segments(nblocks, nseg, warm_blocks)
    mine = harness.shard_segments(len(plan), world, rank)
    d_iq = pkg.DeviceBuffer.from_numpy(iq)
    out = {}

    def run_one(b0, b1):
        view = pkg.DeviceBuffer.__new__(pkg.DeviceBuffer)            # a window into the resident capture
        view.ptr, view.nbytes = d_iq.ptr + 4 * b0 * N, 4 * (b1 - b0) * N
        try:
            return pkg.run_chain(view, samprate=fs, binsize=binsize, symrate="1024", decode_delay=a.delay)
        finally:
            view.ptr = None

    def step():
        out["parts"] = harness.run_segments(plan, mine, run_one, concurrency=a.segment_concurrency)

    dd = dist if world > 1 else None
    fence = harness.make_fence(dd, torch.cuda.synchronize)
    dt_local = harness.timed_steps(step, steps, warmup, fence)
    dt = harness.max_over_ranks(dd, torch, dt_local, redev)
    per_rank = harness.gather_per_rank(dd, round(dt_local / steps * 1e3, 3))
    bits, seams = harness.gather_and_stitch(dd, plan, out["parts"], N / fs * 1024.545058 / 2, segmod.stitch)
    d_iq.free()
    if rank != 0:
        return None
    nbits, ok = chain_check(bits, sent)
    nsamp = nblocks * N
    processed = int(sum((b1 - b0) * N for b0, _, b1 in plan))
    return {"metric": "end-to-end IQ Msamples/s", "value": round(nsamp * steps / dt / 1e6, 3), "unit": "Msamples/s",
            "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3), "ms_per_step_per_rank": per_rank,
            "scaling": "strong",
            "config": {"workload": "ONE capture of %d blocks of 2^%d samples (%.1f s of %g kS/s int16 IQ, %g Hz bins, full-band "
                                   "carrier search) cut into %d overlapped block-aligned segments (%d after merging those that "
                                   "start at block 0), warm-up %d blocks, segment g -> rank g mod %d, %d chains at a time per "
                                   "GPU, capture resident in HBM, stitched on rank 0"
                                   % (nblocks, int(np.log2(N)), nsamp / fs, fs / 1e3, binsize, nseg, len(plan), warm_blocks, world, a.segment_concurrency),
                       "decoded_bits": nbits, "segments": len(plan), "seams": seams,
                       "samples_processed_incl_overlap": processed,
                       "processed_rate_Msamples_per_s": round(processed * steps / dt / 1e6, 3)},
            "roofline": chain_roofline(fs, binsize, processed, dt / steps),
            "check": {"decoded_run_found_in_sent_stream": ok, "all_seams_verified": bool(seams and seams["matched"] == seams["total"])}}


def stress_capture(a, world):
    """The 64-block 10 MS/s capture of the stress record (2.1 GB), generated once per run by forked worker processes.
    Called BEFORE this process imports torch.cuda or loads a HIP library: the workers must not inherit an initialised GPU
    runtime (no fork of a process that holds the device)."""
    from importlib import import_module
    load_pkg()
    synth = import_module("isee3_decoder_amd.synth")
    workers = max(1, min(16, len(os.sched_getaffinity(0)) // max(1, world)))
    t0 = time.perf_counter()
    iq, sent = synth.iq_capture_parallel(3, 1.0e7, a.stress_blocks * (1 << 23), workers=workers)
    return iq, sent, time.perf_counter() - t0


def stress_record(a, ctx, capture):
    """BASELINE configs[4] in the default line: 10 MS/s, 1 Hz bins (N = 2^23), 64 overlapped segments of one block each
    (+ 7 warm-up blocks), full-band search, on the capture stress_capture() made at the start of the run."""
    iq, sent, gen_s = capture
    rec = segmented_record(a, ctx, iq, sent, 1.0e7, 1.0, a.stress_segments, 7, max(1, min(a.steps, 2)), 1)
    if rec is not None:
        rec["capture_generated_in_s"] = round(gen_s, 1)
    return rec


def spawn_ranks(n):
    """One process per GPU: `python -m torch.distributed.run --nnodes=1 --nproc-per-node n bench.py <same arguments>` as a
    child of this (GPU-free) process; returns its exit code.  Rendezvous on 127.0.0.1 and a free port."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def dry_ranks(a, rank, world):
    """--dry-ranks: the N-rank path without a device.  Same launcher, same harness calls as the real run (segments dealt
    to ranks, barrier on both sides of the timed region, MAX over ranks, SUM of 1 = ranks_seen, per-rank times), gloo
    instead of RCCL, a sleep instead of the decode.  Rank 0 prints the one line."""
    import torch
    import torch.distributed as dist
    load_pkg()
    from importlib import import_module
    harness = import_module("isee3_decoder_amd.harness")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    dd = dist if world > 1 else None
    nseg = world * a.segments_per_gpu
    mine = harness.shard_segments(nseg, world, rank)
    units = 1000                                      # stand-in "symbols" per segment and step

    def step():
        for _ in mine:
            time.sleep(0.01 * (1 + rank))             # later ranks are slower: the MAX has something to pick

    fence = harness.make_fence(dd, lambda: None)
    dt_local = harness.timed_steps(step, a.steps, a.warmup, fence)
    dt = harness.max_over_ranks(dd, torch, dt_local, "cpu")
    seen = harness.ranks_seen(dd, torch, "cpu")
    per_rank = harness.gather_per_rank(dd, round(dt_local / a.steps * 1e3, 3))
    segs = harness.gather_per_rank(dd, mine)
    if rank == 0:
        print(json.dumps({"metric": "dry run of the rank launcher (no device work)", "value": round(units * nseg * a.steps / dt, 1),
                          "unit": "stand-in units/s", "n_gpus": world, "ranks_seen": seen, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": round(dt / a.steps * 1e3, 3), "ms_per_step_per_rank": per_rank, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "none", "dry_ranks": True,
                          "config": {"workload": "sleep stand-in, %d segment(s) per rank" % a.segments_per_gpu,
                                     "segments_by_rank": segs}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--symbols", type=int, default=10_000_000)
    ap.add_argument("--delay", type=int, default=200)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--segments-per-gpu", type=int, default=1,
                    help="independent streams decoded concurrently on each GPU (own HIP streams)")
    ap.add_argument("--split", type=int, default=2,
                    help="> 1: decode each stream with this many decoders working on consecutive parts at the same time, "
                         "verified at the seams (v224hip_stream_decode_split): same output as one decoder.  1: one "
                         "decoder per stream.  With > 1 a single-decoder pass is timed too (1 step) and reported "
                         "beside the headline, together with the check that both outputs are identical")
    ap.add_argument("--split-warm", type=int, default=14280, help="warm-up bits before each part (14 chunks)")
    ap.add_argument("--cpu-bits", type=int, default=6000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-frames", action="store_true", help="skip the framed-batch record")
    ap.add_argument("--no-chain", action="store_true", help="skip the chain half of the metric (kernel profiling runs)")
    ap.add_argument("--workload", choices=["viterbi", "chain"], default="viterbi")
    ap.add_argument("--chain-seconds", type=float, default=60.0)
    ap.add_argument("--chain-rate", type=float, default=250000.0)
    ap.add_argument("--chain-bin", type=float, default=1.0)
    ap.add_argument("--chain-steps", type=int, default=0, help="timed chain steps in the default run (0: as --steps)")
    ap.add_argument("--chain-cpu-seconds", type=float, default=21.0, help="seconds of capture given to the CPU chain baseline")
    ap.add_argument("--chain-segments", type=int, default=1,
                    help="> 1: cut ONE capture into this many overlapped segments over all ranks (configs[4])")
    ap.add_argument("--chain-warm-blocks", type=int, default=7)
    ap.add_argument("--no-stress", action="store_true", help="skip the configs[4] record (10 MS/s, 64 overlapped segments)")
    ap.add_argument("--stress-blocks", type=int, default=128,
                    help="blocks of 2^23 samples in the stress capture (64 B: every one of the 64 segments owns B blocks and carries 7 "
                         "warm-up blocks; B = 2: 107 s of signal, 4.3 GB)")
    ap.add_argument("--stress-segments", type=int, default=64)
    ap.add_argument("--segment-concurrency", type=int, default=2,
                    help="chains that run at the same time on one GPU in the segmented (stress / --chain-segments) form")
    ap.add_argument("--dry-ranks", action="store_true",
                    help="no device work: every rank runs a stand-in step through the same launcher / sharding / fence / MAX "
                         "code on gloo and rank 0 prints the line (CPU test of the N-rank path)")
    a = ap.parse_args()

    # `python bench.py --gpus N`, N > 1, outside torchrun: start the N ranks ourselves -- as a CHILD process, before this
    # process has imported torch.cuda or loaded a HIP library (never an exec of a process that touched the GPU)
    if a.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(spawn_ranks(a.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d -- launch as `python bench.py --gpus N` or as "
                         "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`" % (a.gpus, world))
    if a.dry_ranks:
        return dry_ranks(a, rank, world)
    # the stress record's capture: made now, by forked workers, while this process has not touched a device yet
    capture = stress_capture(a, world) if a.workload == "viterbi" and not a.no_chain and not a.no_stress else None

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback in the product path)")
    # rehearsal mode for a one-GPU box: ISEE3_BENCH_ONE_DEVICE=1 puts every rank on device 0 and
    # uses gloo for the barrier / MAX (RCCL refuses two ranks on one device)
    one_dev = os.environ.get("ISEE3_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local = 0
    torch.cuda.set_device(local)
    redev = "cpu" if one_dev else "cuda"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    pkg = load_pkg()
    from importlib import import_module
    synth = import_module("isee3_decoder_amd.synth")
    harness = import_module("isee3_decoder_amd.harness")
    pkg.v224_lib().v224hip_set_device(local)
    pkg.dsp_lib().isee3dsp_set_device(local)
    ctx = dict(pkg=pkg, harness=harness, synth=synth, dist=dist, torch=torch, rank=rank, world=world, redev=redev)
    ctx["ranks_seen"] = harness.ranks_seen(dist if world > 1 else None, torch, redev)
    if a.workload == "chain":
        chain_workload(a, ctx)
        if world > 1:
            dist.destroy_process_group()
        return

    nbits = a.symbols // 2
    nseg = world * a.segments_per_gpu
    mine = harness.shard_segments(nseg, world, rank)      # segment g -> rank g mod world
    chunk = a.chunk or 2040                               # 136 passes of 15 steps; tracebacks in the decoder's own stream
    segs = []
    for g in mine:
        syms, bits, noise_mask = synth.coded_stream(1000 + g, nbits, 3.0, 24.0, 1.0)
        decs = []
        for _ in range(max(1, a.split)):
            dec = pkg.Viterbi224(a.delay + chunk, 3)           # the 15-step LDS engine (the remainder engines are test subjects)
            dec.set_option("chunk", chunk)
            decs.append(dec)
        segs.append(dict(dec=decs[0], decs=decs, d_syms=pkg.DeviceBuffer.from_numpy(syms), d_out=pkg.DeviceBuffer(nbits),
                         syms=np.ascontiguousarray(syms), bits=bits, noise_mask=noise_mask))
    dec = segs[0]["dec"]

    slab = 8 * chunk                                     # bits handed over per call and segment
    redone = [0]                                         # parts the split decode had to redo (seam check failed)

    def step_single():
        # every segment has its own decoder and HIP streams.  With several segments per GPU the
        # enqueues are interleaved slab by slab, so that launches of different segments sit next to
        # each other in the device queues and overlap (one segment's loads/stores under another's
        # arithmetic: 9.3 vs 15.4 us per launch with two, scratch/acs_bench.hip).
        for sg in segs:
            sg["dec"].init(0)
        for pos in range(0, nbits, slab):
            n = min(slab, nbits - pos)
            for sg in segs:
                sg["dec"].stream_decode_dev(sg["d_syms"], n, a.delay, sg["d_out"], sym_offset=2 * pos, out_offset=pos)

    def step_split():
        for sg in segs:
            redone[0] += pkg.stream_decode_split(sg["decs"], sg["d_syms"], nbits, a.delay, sg["d_out"], a.split_warm)

    dd = dist if world > 1 else None
    fence = harness.make_fence(dd, torch.cuda.synchronize)
    single = None
    if a.split > 1:
        # reference pass: ONE decoder per stream (1 warm-up + 1 timed step); its kernel timing feeds `roofline`, its
        # output is what the split decode has to reproduce byte for byte
        harness.timed_steps(step_single, 0, 1, fence)
        dec.set_option("profile", 4)
        dec.acs_stats(reset=True)
        dt1 = harness.timed_steps(step_single, 1, 0, fence)
        launches, ms, steps_timed = dec.acs_stats()
        dec.set_option("profile", 0)
        dt1 = harness.max_over_ranks(dd, torch, dt1, redev)
        single_out = segs[0]["d_out"].to_numpy(np.uint8).copy()
        harness.timed_steps(step_split, 0, a.warmup, fence)
        redone[0] = 0
        dec.set_option("profile", 4)
        dec.acs_stats(reset=True)
        dt_local = harness.timed_steps(step_split, a.steps, 0, fence)
        l2, ms2, st2 = dec.acs_stats()
        dec.set_option("profile", 0)
        single = {"value": round(2 * nbits * nseg / dt1 / 1e6, 4), "ms_per_step": round(dt1 * 1e3, 3),
                  "identical_output": bool(np.array_equal(single_out, segs[0]["d_out"].to_numpy(np.uint8))),
                  "split_avg_launch_ms": round(ms2 / l2, 6) if l2 else None}
    else:
        harness.timed_steps(step_single, 0, a.warmup, fence)
        dec.set_option("profile", 4)
        dec.acs_stats(reset=True)
        dt_local = harness.timed_steps(step_single, a.steps, 0, fence)
        launches, ms, steps_timed = dec.acs_stats()
        dec.set_option("profile", 0)
    dt = harness.max_over_ranks(dd, torch, dt_local, redev)
    per_rank = harness.gather_per_rank(dd, round(dt_local / a.steps * 1e3, 3))

    # SURVEY 8(d) config 2's region -- "first kernel enqueue -> last output bit in host memory, H2D of the input
    # included": ONE more step that starts from host symbols and ends with the bits in a host array.  Reported beside
    # `value` (which, by the bench contract, is the rate with the inputs resident in HBM).
    resident_out = segs[0]["d_out"].to_numpy(np.uint8).copy()
    host_out = [np.empty(nbits, np.uint8) for _ in segs]
    L = pkg.v224_lib()

    def step_host_buffers():
        for sg, ho in zip(segs, host_out):
            if L.v224hip_h2d(sg["d_syms"].ptr, sg["syms"].ctypes.data, sg["syms"].nbytes) != 0:
                raise RuntimeError("h2d failed")
        (step_split if a.split > 1 else step_single)()
        for sg, ho in zip(segs, host_out):
            if L.v224hip_d2h(ho.ctypes.data, sg["d_out"].ptr, ho.nbytes) != 0:
                raise RuntimeError("d2h failed")

    dt_host = harness.max_over_ranks(dd, torch, harness.timed_steps(step_host_buffers, 1, 0, fence), redev)
    host_same = bool(np.array_equal(host_out[0], resident_out))
    d_out, bits, noise_mask = segs[0]["d_out"], segs[0]["bits"], segs[0]["noise_mask"]

    # sanity: decoded bits equal sent bits away from the noise blocks (does not replace tests/)
    out = d_out.to_numpy(np.uint8)
    dec_bits = out[a.delay + 23:]
    ref_bits = bits[1:1 + len(dec_bits)]
    clean = np.ones(len(dec_bits), bool)
    nz = np.flatnonzero(noise_mask[::2])
    for i in nz[:: 256]:
        lo = max(0, i - 1200); clean[lo:i + 1200] = False
    ber = float(np.mean(dec_bits[clean] != ref_bits[clean])) if clean.any() else -1.0
    for sg in segs:                                     # free the decoders (2.2 GiB rings) before the chain runs
        for d in sg["decs"]:
            d.close()
        sg["d_syms"].free(); sg["d_out"].free()

    frames = frames_record(pkg, synth) if not a.no_frames else None
    chain = None
    if not a.no_chain:
        chain = chain_record(a, ctx, a.chain_seconds, a.chain_rate, a.chain_bin, a.chain_steps or a.steps, max(1, a.warmup),
                             world == 1 and not a.no_cpu)
        pkg.release_chain_objects()
    stress = None
    if capture is not None:
        stress = stress_record(a, ctx, capture)
        capture = None
        pkg.release_chain_objects()

    if rank == 0:
        total_syms = 2 * nbits * nseg * a.steps
        avg_ms = ms / launches if launches else float("nan")
        steps_per_launch = steps_timed / launches if launches else 0
        kern = "k_acs_lds15"
        pmc, psrc = pmc_constants(kern)
        traffic = pmc.get("hbm_bytes_per_launch")
        valu = pmc.get("valu_insts_per_wave")
        cyc = pmc.get("valu_cycles_per_inst", 4.2)
        alg = ALG_BYTES_PER_STEP * steps_per_launch
        peak = N_SIMD * CLOCK_GHZ / cyc

        def fractions(t):
            """the kernel's roofs at t seconds per launch: VALU issue (the binding one), physical HBM, SURVEY 8(d)'s algorithmic bytes"""
            f = {"avg_launch_ms": round(t * 1e3, 6), "algorithmic_GBps": round(alg / t / 1e9, 1),
                 "algorithmic_x_peak": round(alg / t / 1e9 / HBM_PEAK_GBS, 4)}
            if traffic:
                f["hbm"] = {"bound": "hbm", "achieved": round(traffic / t / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(traffic / t / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic}
                f["hbm_physical_GBps"], f["hbm_physical_frac"] = f["hbm"]["achieved"], f["hbm"]["frac"]
            if valu:
                ach = valu * WAVES_PER_LAUNCH / t / 1e9
                f.update({"achieved": round(ach, 1), "peak": round(peak, 1), "unit": "G wave-instructions/s (VALU)",
                          "frac": round(ach / peak, 4)})
            elif traffic:
                f.update({"achieved": f["hbm"]["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": f["hbm"]["frac"]})
            return f

        roof = {"kernel": kern, "bound": "valu-issue" if valu else "hbm", "trellis_steps_per_launch": steps_per_launch,
                "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, %s)" % psrc,
                "algorithmic_bytes_per_step": ALG_BYTES_PER_STEP, "algorithmic_bytes_per_launch": int(alg),
                "valu_insts_per_wave": valu, "valu_cycles_per_inst": cyc,
                "note": "15 trellis steps share one pass over the path metrics, so a launch is bound by VALU issue, not by HBM: "
                        "frac = VALU instructions per wave x 4 waves per SIMD x %.1f cycles per instruction (measured for this VOP3P / "
                        "VOP2 mix, profiles/r01_valu_rate.txt) / (time per launch x %.1f GHz); `hbm` = the same launch against the "
                        "8 TB/s HBM peak with the bytes it physically moves; the SURVEY 8(d) algorithmic bytes exceed what is "
                        "moved (algorithmic_x_peak: fusion depth, not efficiency)" % (cyc, CLOCK_GHZ)}
        lone = fractions(avg_ms * 1e-3) if launches else None
        if lone:
            lone.update({"launches_timed": launches,
                         "measured_on": ("the single-decoder reference pass of this run" if a.split > 1 else "the timed steps") +
                                        ": one %s at a time on the GPU; HIP events on the decoder's own stream around runs of "
                                        "back-to-back launches" % kern})
        if single and single.get("split_avg_launch_ms"):
            # the timed region: two decoders' launches side by side; decoder 0's launch period covers one launch of EACH
            # decoder, i.e. two launches complete per period
            tp = single["split_avg_launch_ms"] * 1e-3
            roof.update(fractions(tp / 2))
            roof.update({"launch_pair_ms": single["split_avg_launch_ms"],
                         "measured_on": "the timed steps: this stream's two decoders run their launch chains side by side (HIP events on "
                                        "decoder 0's stream; one launch of each decoder per period, avg_launch_ms = period / 2); the lone "
                                        "launch is under single_decoder",
                         "single_decoder": lone})
        elif lone:
            roof.update(lone)
        res = {
            "metric": "Viterbi K=24 Msymbols/s",
            "value": round(total_syms / dt / 1e6, 4), "unit": "Msymbols/s",
            "n_gpus": world, "ranks_seen": ctx["ranks_seen"], "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "ms_per_step_per_rank": per_rank,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u16", "data": "synthetic",
            "config": {"workload": "viterbi224 ACS+chainback streaming, 2^23 states, 8-bit soft syms, "
                                   "decode delay %d, %d symbols per GPU per step" % (a.delay, 2 * nbits),
                       "residency": "value: symbols resident in HBM when the timed region starts, decoded bits left in HBM "
                                    "(bench contract); host_buffers: the same step from host symbols to host bits",
                       "host_buffers": {"value": round(2 * nbits * nseg / dt_host / 1e6, 4), "ms_per_step": round(dt_host * 1e3, 3),
                                        "steps": 1, "identical_output": host_same,
                                        "what": "SURVEY 8(d) config 2's region: %.1f MB of symbols H2D (pageable numpy array, "
                                                "blocking copy), the whole decode, %.1f MB of bits D2H into a host array -- "
                                                "PCIe inclusive, never the headline" % (2 * nbits / 1e6, nbits / 1e6)},
                       "engine": "lds15", "steps_per_launch": 15,
                       "chunk_bits": chunk,
                       "segments_per_gpu": a.segments_per_gpu, "parallelism": "segments x%d" % nseg,
                       "decoders_per_stream": a.split,
                       "split": None if a.split <= 1 else {
                           "what": "each stream is cut into %d consecutive parts decoded at the same time by %d decoders; "
                                   "part j > 0 warms up %d bits early and its path metrics are compared with the previous "
                                   "decoder's at the seam (all 2^23 states equal => identical continuation, else the part is "
                                   "decoded again): v224hip_stream_decode_split" % (a.split, a.split, a.split_warm),
                           "parts_redone": redone[0], "single_decoder": single}},
            "roofline": roof,
            "check": {"ber_clean": ber, "bits": int(clean.sum())},
        }
        if world == 1 and not a.no_cpu:
            res["cpu_baseline"] = cpu_baseline(a.cpu_bits)
            res["speedup_vs_cpu_1core"] = round(res["value"] / res["cpu_baseline"]["value"], 1)
        if frames is not None:
            res["frames"] = frames
        if chain is not None:
            res["chain"] = chain
        if stress is not None:
            res["stress"] = stress
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
