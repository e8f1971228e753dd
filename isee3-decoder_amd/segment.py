"""Overlapped-segment decoding of one long capture (BASELINE.json configs[4]).

The reference has no such notion: its stages are single sequential filters.  Cutting a capture is what
lets one capture use several GPUs (or several concurrent chains on one GPU, whose kernels overlap).
Each segment is a block-aligned slice of the IQ capture, extended to the left by `warm_blocks` FFT
blocks so that symdemod's timing search, vdecode's symbol-pair phase (decided once per 4096 symbols) and
the Viterbi path history have settled before the segment's own samples begin: the warm-up must span
more than ~2600 decoded bits (5.2 s at 512 bit/s).  Segments are decoded
independently (no collective), then stitched on the host by matching decoded bits inside the overlap.

This is NOT bit-exact by construction -- a decoder restarted inside the stream starts from different
path metrics -- so the stitcher reports how many seams matched exactly; on a clean signal all do.
"""
import threading

import numpy as np


def plan_segments(nblocks, nseg, warm_blocks):
    """[(first block incl. warm-up, first own block, end block)] for nseg near-equal block-aligned segments.  Segments
    whose warm-up would reach back to block 0 all decode the capture from its first sample; they are merged into ONE
    first segment (decoding the same start several times buys nothing), so fewer than nseg entries may come back."""
    edges = [round(i * nblocks / nseg) for i in range(nseg + 1)]
    segs = [(max(0, edges[i] - warm_blocks), edges[i], edges[i + 1]) for i in range(nseg) if edges[i + 1] > edges[i]]
    first = [s for s in segs if s[0] == 0]
    rest = [s for s in segs if s[0] > 0]
    return ([(0, 0, first[-1][2])] if first else []) + rest


def stitch(parts, overlap_bits, probe_len=160, tail_bits=None, verify_back=400, settled=2300, end_guard=64, delay=200,
           window_bits=512):
    """parts: decoded bit strings (bytes of '0'/'1') of consecutive overlapping segments;
    overlap_bits[i]: how many decoded bits lie between the start of part i+1 and the cut (its warm-up region).
    A probe is taken from the settled end of that region and located in the previous part -- but only where it
    can be: part i ends `tail_bits` = (min, max) bits short of the cut, so bit w of part i+1 sits at
    len(part i) + tail - (overlap - w).  The tail is what the stages hold back at the end of their input: vdecode its
    `delay` bits (vdecode.c:151-158), symdemod the last partial window (symdemod.c:124-125, up to `window_bits`), so by
    default tail_bits = (delay - 50, delay + window_bits + 78): the window of possible probe positions is then 768 bits
    wide, NARROWER than one 1 024-bit minor frame, and exactly repeating frames (idle / fill telemetry) cannot match
    twice in it.  Every hit inside the window is tried, nearest to the window's middle first, and must hold up over
    EVERYTHING the two parts share from up to `verify_back` bits before the probe (not before the part's `settled`
    bit) to the end of part i; a seam where no hit -- or more than one -- verifies counts as unmatched.
    end_guard: the last bits of a part are not compared -- its final symdemod window ran into the end of the input
    (symdemod.c reads whatever its buffer holds there), so they may differ from a decode that had the samples.
    Returns (joined bits, seams matched, seams total)."""
    if tail_bits is None:
        tail_bits = (max(0, delay - 50), delay + window_bits + 78)
    out = parts[0]
    ok = 0
    for i, (nxt, ovl) in enumerate(zip(parts[1:], overlap_bits)):
        placed = False
        # the first ~2300 bits of a restarted decode are unreliable: start-up delay, and vdecode decides its
        # symbol-pair phase only once per 2048 ODD symbols = two frames (vdecode.c:122-139).  Probe between
        # there and the end of the overlap.
        for w in range(max(settled, ovl - 300), settled - 100, -100):
            probe = nxt[w:w + probe_len]
            if len(probe) < probe_len:
                continue
            expect = len(out) - (ovl - w)
            lo = max(0, expect + tail_bits[0] - 64)
            hi = min(len(out), expect + tail_bits[1] + 64 + probe_len)
            hits, p = [], out.find(probe, lo, hi)
            while p >= 0:
                hits.append(p)
                p = out.find(probe, p + 1, hi)
            mid = expect + (tail_bits[0] + tail_bits[1]) // 2
            good = []
            for p in sorted(hits, key=lambda q: abs(q - mid)):
                back = max(0, min(verify_back, w - settled, p))
                share = max(probe_len + back, len(out) - end_guard - (p - back))
                if out[p - back:p - back + share] == nxt[w - back:w - back + share]:
                    good.append(p)           # the alignment holds over the rest of the overlap
            if len(good) != 1:
                continue                     # absent, or the overlap is periodic inside the window: not decidable here
            out = out[:good[0]] + nxt[w:]
            placed = True
            break
        if placed:
            ok += 1
        else:
            out = out + nxt          # no verified match inside the overlap: keep everything, caller sees the count
    return out, ok, len(parts) - 1


def decode_segmented(iq, samprate, binsize, nseg, run_chain, warm_blocks=7, concurrency=2, symrate="1024",
                     decode_delay=200):
    """Cut `iq` (int16 interleaved) into nseg overlapped segments, run the whole chain on each
    (`concurrency` chains at a time in this process, so their kernels overlap on the device) and
    stitch.  Returns (bits, seams_ok, seams, samples_processed_including_overlap)."""
    iq = np.ascontiguousarray(iq, dtype=np.int16)
    N = 1 << int(np.rint(np.log2(samprate / binsize)))
    nblocks = (len(iq) // 2) // N
    plan = plan_segments(nblocks, nseg, warm_blocks)
    parts = [None] * len(plan)
    lock = threading.Lock()
    todo = list(range(len(plan)))
    errors = []

    def worker():
        while True:
            with lock:
                if not todo:
                    return
                i = todo.pop(0)
            b0, _, b1 = plan[i]
            try:
                parts[i] = run_chain(iq[2 * b0 * N:2 * b1 * N], samprate=samprate, binsize=binsize, symrate=symrate,
                                     decode_delay=decode_delay)
            except Exception as e:                      # noqa: BLE001
                errors.append(e)
                return

    threads = [threading.Thread(target=worker) for _ in range(max(1, concurrency))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    bps = (1024.545058 / 2 if symrate in (None, "1024") else float(symrate) / 2)
    overlaps = [int((plan[i + 1][1] - plan[i + 1][0]) * N / samprate * bps) for i in range(len(plan) - 1)]
    bits, ok, seams = stitch(parts, overlaps)
    processed = sum((b1 - b0) * N for b0, _, b1 in plan)
    return bits, ok, seams, processed
