#!/usr/bin/env python3
"""bench.py -- headline benchmark: K=24 r=1/2 Viterbi streaming decode on MI355X.

Workload (BASELINE.json configs[1]): one 1e7-symbol synthetic soft-symbol stream per GPU, decoded
with the semantics of `vdecode -d 200` (one trellis step + decodebit(200,0) per bit), inputs
resident in HBM before the timed region, decoded bits left in HBM.  A "step" = one full pass of
that stream (init + 5e6 trellis steps + 5e6 tracebacks).  N>1: every rank decodes its own
independent stream (segments shard one-per-GPU, no collective in the data path) -> weak scaling.

Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline      : HBM roofline of the ACS kernel (algorithmic 34 603 008 B per trellis step),
                  average launch time from HIP events recorded on the decoder's own stream
  cpu_baseline  : the reference's SSE2 decoder (oracle/_ref, built from /root/reference in the
                  build container) on ONE host core, bounded sample
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


def cpu_baseline(nbits_sample):
    """Reference SSE2 decoder on one host core (falls back to the oracle port restatement)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    syms = np.full(2 * nbits_sample, 128, np.uint8)
    rng = np.random.default_rng(1)
    syms = rng.integers(0, 256, 2 * nbits_sample, dtype=np.uint8)
    if orc.have_ref():
        d = orc.RefV224(256, "sse2")
        kind, what = "reference", "viterbi224_sse2.c (oracle/_ref), update_viterbi224_blk"
    else:
        os.environ["OMP_NUM_THREADS"] = "1"
        d = orc.OracleV224(256, orc.FAST)
        kind, what = "port", "oracle FAST engine (port semantics, SSE2), 1 thread"
    d.init(0)
    d.update(syms[:512], 256)                     # touch memory
    d.init(0)
    t0 = time.perf_counter()
    done = 0
    while done < nbits_sample:
        n = min(256, nbits_sample - done)
        d.update(syms[2 * done:2 * (done + n)], n)
        done += n
    dt = time.perf_counter() - t0
    d.close()
    res = {"value": round(2 * nbits_sample / dt / 1e6, 6), "unit": "Msymbols/s", "cores": 1, "kind": kind,
           "sample": "%d trellis steps of uniform-random symbols, %s, %.1f s" % (nbits_sample, what, dt)}
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


def chain_workload(a, rank, world, local, dist, torch, pkg, redev="cuda"):
    """BASELINE.json configs[2]/[3]: full pmdemod | symdemod | vdecode chain on synthetic int16 IQ,
    one independent capture per GPU (libisee3chain.so = the three C pipe stages as threads of the
    calling process; the capture is fed from host memory through a pipe, so PCIe and pipe copies are
    inside the time)."""
    from importlib import import_module
    synth = import_module("isee3_decoder_amd.synth")
    harness = import_module("isee3_decoder_amd.harness")
    fs = float(a.chain_rate)
    # segmented mode: ONE capture, the same on every rank; otherwise one capture per rank
    iq, sent = synth.iq_capture(3 if a.chain_segments > 1 else 3 + rank, fs, a.chain_seconds, amp=None)
    pkg.v224_lib().v224hip_set_device(local)
    pkg.dsp_lib().isee3dsp_set_device(local)
    out = {}
    segmod = import_module("isee3_decoder_amd.segment")
    N = 1 << int(np.rint(np.log2(fs / a.chain_bin)))
    nblocks = (len(iq) // 2) // N
    plan = segmod.plan_segments(nblocks, a.chain_segments, a.chain_warm_blocks) if a.chain_segments > 1 else None
    # configs[4]: the capture is cut into overlapped block-aligned segments, segment g -> rank g mod world,
    # two chains at a time per GPU (their kernels overlap), parts stitched on rank 0
    mine = harness.shard_segments(len(plan), world, rank) if plan is not None else []

    def step():
        # libisee3chain.so: the three C pipe stages as threads of THIS process (HIP context stays warm)
        if plan is None:
            out["bits"] = pkg.run_chain(iq, samprate=fs, binsize=a.chain_bin, symrate="1024", decode_delay=a.delay)
            return
        import threading
        todo, parts, lock = list(mine), {}, threading.Lock()

        def worker():
            while True:
                with lock:
                    if not todo:
                        return
                    g = todo.pop(0)
                b0, _, b1 = plan[g]
                parts[g] = pkg.run_chain(iq[2 * b0 * N:2 * b1 * N], samprate=fs, binsize=a.chain_bin, symrate="1024",
                                         decode_delay=a.delay)
        ts = [threading.Thread(target=worker) for _ in range(2)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        out["parts"] = parts

    fence = harness.make_fence(dist if world > 1 else None, torch.cuda.synchronize)
    dt = harness.timed_steps(step, a.steps, a.warmup, fence)
    dt = harness.max_over_ranks(dist if world > 1 else None, torch, dt, redev)
    seams = None
    if plan is not None:
        allparts = [out["parts"]]
        if world > 1:
            allparts = [None] * world
            dist.all_gather_object(allparts, out["parts"])
        if rank == 0:
            merged = {}
            for d in allparts:
                merged.update(d)
            bps = 1024.545058 / 2
            ovl = [int((plan[i + 1][1] - plan[i + 1][0]) * N / fs * bps) for i in range(len(plan) - 1)]
            bits_all, ok, tot = segmod.stitch([merged[g] for g in range(len(plan))], ovl,
                                              from_start=[plan[i + 1][0] == 0 for i in range(len(plan) - 1)])
            out["bits"] = bits_all
            seams = {"matched": ok, "total": tot}
        else:
            out["bits"] = b""
    got = np.frombuffer(out["bits"], np.uint8) - ord("0")
    s = "".join(map(str, sent))
    # the tail: vdecode may need one 2048-symbol frame to settle its symbol-pair phase (vdecode.c:126-139)
    ok = len(got) > 2500 and "".join(map(str, got[-1100:-100])) in s
    if rank == 0:
        nsamp = len(iq) // 2
        units = nsamp * a.steps if plan is not None else nsamp * world * a.steps   # segmented: ONE capture in total
        print(json.dumps({
            "metric": "end-to-end IQ Msamples/s", "value": round(units / dt / 1e6, 3),
            "unit": "Msamples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong" if plan is not None else "weak",
            "vs_baseline": None, "dtype": "f64+u16", "data": "synthetic",
            "config": {"workload": "pmdemod|symdemod|vdecode on %g s of %g kS/s int16 IQ, %g Hz bins, 1024 sym/s "
                                   "Manchester, one capture per GPU, capture in host memory (H2D / D2H of every stage included)"
                                   % (a.chain_seconds, fs / 1e3, a.chain_bin), "decoded_bits": int(len(got)),
                       "segments": a.chain_segments, "warm_blocks": a.chain_warm_blocks if plan is not None else 0,
                       "seams": seams},
            "roofline": None, "check": {"decoded_run_found_in_sent_stream": bool(ok)}}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--symbols", type=int, default=10_000_000)
    ap.add_argument("--delay", type=int, default=200)
    ap.add_argument("--engine", type=int, default=-1)
    ap.add_argument("--k", type=int, default=0)
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
    ap.add_argument("--workload", choices=["viterbi", "chain"], default="viterbi")
    ap.add_argument("--chain-seconds", type=float, default=60.0)
    ap.add_argument("--chain-rate", type=float, default=250000.0)
    ap.add_argument("--chain-bin", type=float, default=1.0)
    ap.add_argument("--chain-segments", type=int, default=1,
                    help="> 1: cut ONE capture into this many overlapped segments over all ranks (configs[4])")
    ap.add_argument("--chain-warm-blocks", type=int, default=7)
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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
    if a.workload == "chain":
        chain_workload(a, rank, world, local, dist, torch, pkg, redev)
        if world > 1:
            dist.destroy_process_group()
        return
    pkg.v224_lib().v224hip_set_device(local)
    from importlib import import_module
    synth = import_module("isee3_decoder_amd.synth")

    harness = import_module("isee3_decoder_amd.harness")
    nbits = a.symbols // 2
    nseg = world * a.segments_per_gpu
    mine = harness.shard_segments(nseg, world, rank)      # segment g -> rank g mod world
    eng = a.engine if a.engine >= 0 else int(os.environ.get("V224HIP_ENGINE", "3"))
    chunk = a.chunk or (1024 if eng == 2 else 1020)           # 1020 = 68 x 15 = 204 x 5
    segs = []
    for g in mine:
        syms, bits, noise_mask = synth.coded_stream(1000 + g, nbits, 3.0, 24.0, 1.0)
        decs = []
        for _ in range(max(1, a.split)):
            dec = pkg.Viterbi224(a.delay + 2 * chunk, a.engine, a.k)
            dec.set_option("chunk", chunk)
            decs.append(dec)
        segs.append(dict(dec=decs[0], decs=decs, d_syms=pkg.DeviceBuffer.from_numpy(syms), d_out=pkg.DeviceBuffer(nbits),
                         bits=bits, noise_mask=noise_mask))
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

    fence = harness.make_fence(dist if world > 1 else None, torch.cuda.synchronize)
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
        dt1 = harness.max_over_ranks(dist if world > 1 else None, torch, dt1, redev)
        single_out = segs[0]["d_out"].to_numpy(np.uint8).copy()
        harness.timed_steps(step_split, 0, a.warmup, fence)
        redone[0] = 0
        dec.set_option("profile", 4)
        dec.acs_stats(reset=True)
        dt = harness.timed_steps(step_split, a.steps, 0, fence)
        l2, ms2, st2 = dec.acs_stats()
        dec.set_option("profile", 0)
        single = {"value": round(2 * nbits * nseg / dt1 / 1e6, 4), "ms_per_step": round(dt1 * 1e3, 3),
                  "identical_output": bool(np.array_equal(single_out, segs[0]["d_out"].to_numpy(np.uint8))),
                  "split_avg_launch_ms": round(ms2 / l2, 6) if l2 else None}
    else:
        harness.timed_steps(step_single, 0, a.warmup, fence)
        dec.set_option("profile", 4)
        dec.acs_stats(reset=True)
        dt = harness.timed_steps(step_single, a.steps, 0, fence)
        launches, ms, steps_timed = dec.acs_stats()
        dec.set_option("profile", 0)
    dt = harness.max_over_ranks(dist if world > 1 else None, torch, dt, redev)
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

    if rank == 0:
        total_syms = 2 * nbits * nseg * a.steps
        avg_ms = ms / launches if launches else float("nan")
        steps_per_launch = steps_timed / launches if launches else 0
        ach = ALG_BYTES_PER_STEP * steps_per_launch / (avg_ms * 1e-3) / 1e9 if launches else None
        traffic = None
        kern = {0: "k_acs_simple", 1: "k_acs_fused", 2: "k_acs_lds8", 3: "k_acs_lds15"}[eng]
        tsrc = None
        for tname in ("r01c_pmc_traffic.json", "r01_pmc_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", tname)
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                if tj.get("kernel") == kern:   # PMC passes cannot share a run with the timing: committed constant
                    traffic, tsrc = tj["hbm_bytes_per_launch"], "profiles/" + tname
                    break
        res = {
            "metric": "Viterbi K=24 Msymbols/s",
            "value": round(total_syms / dt / 1e6, 4), "unit": "Msymbols/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u16", "data": "synthetic",
            "config": {"workload": "viterbi224 ACS+chainback streaming, 2^23 states, 8-bit soft syms, "
                                   "decode delay %d, %d symbols per GPU per step" % (a.delay, 2 * nbits),
                       "engine": {0: "simple", 1: "fused", 2: "lds8", 3: "lds15"}[eng],
                       "steps_per_launch": {0: 1, 1: a.k or int(os.environ.get("V224HIP_K", "5")), 2: 8, 3: 15}[eng],
                       "chunk_bits": chunk,
                       "segments_per_gpu": a.segments_per_gpu, "parallelism": "segments x%d" % nseg,
                       "decoders_per_stream": a.split,
                       "split": None if a.split <= 1 else {
                           "what": "each stream is cut into %d consecutive parts decoded at the same time by %d decoders; "
                                   "part j > 0 warms up %d bits early and its path metrics are compared with the previous "
                                   "decoder's at the seam (all 2^23 states equal => identical continuation, else the part is "
                                   "decoded again): v224hip_stream_decode_split" % (a.split, a.split, a.split_warm),
                           "parts_redone": redone[0], "single_decoder": single}},
            "roofline": {"bound": "hbm", "achieved": round(ach, 1) if ach else None, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4) if ach else None,
                         "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, %s)" % tsrc,
                         "physical_GBps": round(traffic / (avg_ms * 1e-3) / 1e9, 1) if (traffic and launches) else None,
                         "note": "algorithmic bytes (SURVEY 8d: 34 603 008 B per trellis step) exceed the physical traffic "
                                 "because one launch carries the metrics through several steps; the launch itself is "
                                 "VALU-issue bound, see DESIGN.md section 3",
                         "algorithmic_bytes_per_launch": int(ALG_BYTES_PER_STEP * steps_per_launch),
                         "kernel": kern,
                         "measured_on": "the single-decoder reference pass of this run (one k_acs_lds15 at a time on the GPU); in "
                                        "the split pass two decoders' launches overlap" if a.split > 1 else "the timed steps",
                         "avg_launch_ms": round(avg_ms, 6), "trellis_steps_per_launch": steps_per_launch,
                         "algorithmic_bytes_per_step": ALG_BYTES_PER_STEP, "launches_timed": launches},
            "check": {"ber_clean": ber, "bits": int(clean.sum())},
        }
        if world == 1 and not a.no_cpu:
            res["cpu_baseline"] = cpu_baseline(a.cpu_bits)
            res["speedup_vs_cpu_1core"] = round(res["value"] / res["cpu_baseline"]["value"], 1)
        print(json.dumps(res), flush=True)
    for sg in segs:
        for d in sg["decs"]:
            d.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
